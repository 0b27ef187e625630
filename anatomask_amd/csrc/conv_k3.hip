// Dense 3x3x3 stride-1 convolution (forward and data gradient) for gfx950, bf16, channels-last: the decoder's convolutions
// (P/decoder3D.py:20-22, conv3x3x3 of UNetBlock; P/AnatoMask.py:63-65 densify projections) at the sizes that dominate a step.
//
// Structure (round 4): ONE persistent 8-wave workgroup per CU, operands into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), counted
// `vmcnt`, raw `s_barrier`, and the two waves of every SIMD in ANTIPHASE: while waves 0-3 ("X") run the MFMA cluster of a tap run,
// waves 4-7 ("Y") fetch the fragments of theirs from LDS, and vice versa -- the matrix pipe of a SIMD always has one wave feeding it and
// the LDS reads / address work / DMA issue of the other wave run in its shadow.  (conv_igemm.hip, two independent 4-wave workgroups per
// CU with register staging, leaves the pipe idle half of the cycles: profiles/r02_clock_and_pipe.md.)
//
//   * a workgroup owns an 8x4x16 brick of output voxels (wave w of each half: d-plane w / w + 4, four h-rows of 16 voxels) and 64 output
//     channels: 16 accumulator tiles (16 couts x 16 voxels) per wave, kept in registers over all channel slabs and taps;
//   * per 32-channel slab the haloed 10x6x18 source brick (69 KB, 64-byte rows) is DMA'd once and read by all 27 taps; it is double
//     buffered, the next slab (or the next brick's first slab: the workgroup is persistent) lands during the current one.  Rows that lie
//     outside the volume are out-of-range buffer offsets: the DMA writes zeros.  The 16-byte chunks of a row are XOR-swizzled with bit 2
//     of the row's x coordinate ON THE SOURCE SIDE (the DMA writes LDS linearly, lane i -> base + 16 i): a B-fragment read -- 16
//     consecutive voxels x 4 chunks -- is then bank-conflict free for every tap shift, and its address is (lane constant of the tap's
//     w-shift) + (compile-time immediate of its d- / h-shift): no vector or scalar instruction per read;
//   * weights arrive per "run" (the three h-taps of one (d, w) shift: they share 6 fragment rows instead of reading 12) in two 12 KB
//     slots, the run after next is in flight while the current one is contracted;
//   * one time slot = [X: fragment reads + DMA issue | Y: 48 MFMAs], barrier, [X: 48 MFMAs | Y: fragment reads], barrier.
// Epilogue as conv_igemm.hip (bias, optional eval-mode BatchNorm scale/shift + skip + activation, 16-byte stores); the per-channel
// (sum, sum of squares) of the STORED values are kept per lane over all bricks of the workgroup and reduced once at the end (8 rows per
// workgroup), deterministic.
#include <stdlib.h>
#include "common.h"
#include "../../include/anatomask_hip.h"
#include "conv_plan.h"

using namespace amconv;

namespace {

constexpr int KBD = 8, KBH = 4, KBW = 16;
constexpr int KED = KBD + 2, KEH = KBH + 2, KEW = KBW + 2;
constexpr int KPLANE = KEH * KEW;                 // 108 rows per d-plane of the staged brick
constexpr int KNROW = KED * KPLANE;               // 1080
constexpr int KBRICKB = KNROW * 64;               // 69 120 bytes per slab buffer
constexpr int KNPIECE = 68;                       // 16-row DMA pieces (1 KB each); the last one is anchored at row 1064 (rewrites 8 rows)
constexpr int KWSLOT = 3 * 64 * 64;               // one run of weights: 3 taps x 64 couts x 64 B
constexpr int KLDS_W = 2 * KBRICKB;
constexpr int KLDS = KLDS_W + 2 * KWSLOT;         // 162 816 of 163 840
constexpr int K3_MIN_UNITS = 512;                 // below two units per CU the brick kernel of conv_igemm.hip fills the chip better

struct K3Args {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* partials;
  const float* ep_scale; const float* ep_shift; const bf16_t* ep_res; int ep_act;
  int B, D, H, W, Cin, Cout, Cinp, Coutp;
  int nbd, nbh, nbw, ny, nunit, nslab;   // bricks per dimension, output-channel tiles, units = B * bricks * ny, 32-channel slabs
  int flip;                              // data gradient: the weight tap index is mirrored (26 - t)
  int nt_store;
  unsigned w_bytes;
#ifdef AM_ABLATE
  int dbg;                               // tools build: 1 no stores, 2 no brick DMA, 4 no weight DMA, 8 no MFMAs
#endif
};

#define K3_LDSP(off) ((__attribute__((address_space(3))) void*)(lds + (off)))

// (unit, slab) + the unit's brick / channel tile, packed (all wave-uniform: these live in SGPRs across the whole walk)
struct Item { int u, k, c0, c1; };                  // c0 = bw | bh << 8 | bd << 16 (brick indices), c1 = b | ytile << 8
#define IT_Q0W(it) (((it).c0 & 255) * KBW)
#define IT_Q0H(it) ((((it).c0 >> 8) & 255) * KBH)
#define IT_Q0D(it) ((((it).c0 >> 16) & 255) * KBD)
#define IT_B(it) ((it).c1 & 255)
#define IT_CO0(it) (((it).c1 >> 8) * 64)

template <int NS>
__global__ __launch_bounds__(512, 2) void conv_k3_kernel(K3Args a) {
  static_assert(NS == 4, "64-channel output tiles");
  constexpr int VS = 4, NT = 16 * NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isX = wave < 4;
  const int wq = wave & 3;
  const int g = lane >> 4, r16 = lane & 15;
  const int G = gridDim.x;

  // ---- unit walk: the 32 workgroups of an XCD (ids equal mod 8 under round-robin dispatch: speed only) take 32 consecutive units
  const int gperm = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  auto decode = [&](Item& it) {
    int br = it.u / a.ny;
    const int yt = it.u - br * a.ny;
    const int bw_ = br % a.nbw; br /= a.nbw;
    const int bh_ = br % a.nbh; br /= a.nbh;
    const int bd_ = br % a.nbd;
    it.c0 = bw_ | (bh_ << 8) | (bd_ << 16); it.c1 = (br / a.nbd) | (yt << 8);
  };
  auto advance = [&](Item& it) {
    if (it.k + 1 < a.nslab) { ++it.k; return; }
    it.k = 0; it.u += G;
    if (it.u < a.nunit) decode(it);
  };
  Item cur; cur.u = gperm; cur.k = 0;
  if (cur.u >= a.nunit) return;
  decode(cur);
  Item nxt = cur; advance(nxt);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int wtapB = a.Coutp * a.Cinp * 2;

  // ---- lane constants
  // weights, DMA source: piece = 16 MFMA rows x 64 B; lane -> (row = lane >> 2, LDS chunk position = lane & 3); the tile's rows are the channels
  // crow(16 tile + row) = (tile >> 1) * 32 + (tile & 1) * 4 + ((row >> 2) & 3) * 8 + (row & 3), the stored chunk = position ^ swizzle(row)
  const int wrow = lane >> 2;
  const unsigned wlane = (unsigned)(((((wrow >> 2) & 3) * 8 + (wrow & 3)) * a.Cinp + (((lane & 3) ^ ((wrow >> 1) & 2)) * 8)) * 2);
  // weights, fragment read: row 16 i + r16, chunk g (conv_plan.h swz); slot parity and tap / tile offsets are added per run / as immediates
  const int aoff = swz(r16, g);
  // source brick, fragment read: voxel (d-plane wave, h-row j, w = r16) under shift (zd, yy, xw): row (wave + zd) * 108 + (j + yy) * 18 + r16 + xw
  int bx[3];
#pragma unroll
  for (int xw = 0; xw < 3; ++xw) bx[xw] = (wave * KPLANE + r16 + xw) * 64 + ((g ^ (((r16 + xw) >> 2) & 1) * 2) << 4);

  // ---- DMA issue (X waves)
  auto issue_weights = [&](const Item& it, const int run, const int slot) __attribute__((always_inline)) {
    // run = zd * 3 + xw; its taps th = 0..2 are the h-shifts; X wave wq brings cout tile wq of each tap
    const int zd = run / 3, xw = run - zd * 3;
#pragma unroll
    for (int th = 0; th < 3; ++th) {
      const int t = zd * 9 + th * 3 + xw, widx = a.flip ? 26 - t : t;
      const unsigned so = (unsigned)(widx * wtapB + ((IT_CO0(it) + (wq >> 1) * 32 + (wq & 1) * 4) * a.Cinp + it.k * 32) * 2);
#ifdef AM_ABLATE
      if (a.dbg & 4) continue;
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, K3_LDSP(KLDS_W + slot * KWSLOT + th * (NT * 64) + wq * 1024), 16, wlane, so, 0, 0);
    }
  };
  // brick pieces of item `it` into slab buffer `sbuf`: X wave wq owns pieces wq, wq + 4, ... (17 of them); [k0, k1) of those
  auto issue_pieces = [&](const Item& it, const int sbuf, const int k0, const int k1) __attribute__((always_inline)) {
    const int q0d = IT_Q0D(it), q0h = IT_Q0H(it), q0w = IT_Q0W(it);
    int dbase = q0d - 1; dbase = dbase < 0 ? 0 : dbase;
    // (the lane's (z, y, x) of a piece are lane constants; recomputed per piece -- ~20 vector instructions in a phase that waits for the
    // other half's MFMAs anyway -- instead of living in 17+ registers for the whole kernel: `lz` is a zero the compiler cannot see through)
    int lz; asm volatile("v_mov_b32 %0, 0" : "=v"(lz));
    const size_t plane_elems = (size_t)a.H * a.W * a.Cin;
    const size_t left = (size_t)(a.D - dbase) * plane_elems * 2;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + ((size_t)IT_B(it) * a.D + dbase) * plane_elems), 0, (int)(left < 0x7fffff00ull ? left : 0x7fffff00ull), 0x00020000);
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int p = wq + 4 * k;
      const int row0 = p * 16 < KNROW - 16 ? p * 16 : KNROW - 16;
      const int rho = row0 + (lane >> 2) + lz;
      const int z = rho / KPLANE, rem = rho - z * KPLANE, yy = rem / KEW, xx = rem - yy * KEW;
      const int d = q0d - 1 + z, h = q0h - 1 + yy, w_ = q0w - 1 + xx;
      const bool ok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w_ < (unsigned)a.W;
      unsigned vo = ok ? (unsigned)(((((d - dbase) * a.H + h) * a.W + w_) * a.Cin + (((lane & 3) ^ (((xx >> 2) & 1) * 2)) * 8)) * 2) : OOB;
#ifdef AM_ABLATE
      if (a.dbg & 2) vo = OOB;
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, K3_LDSP(sbuf * KBRICKB + row0 * 64), 16, vo, (unsigned)(it.k * 64), 0, 0);
    }
  };

  // ---- epilogue state
  f32x4 acc[NS][VS];
  float st1[NS][4], st2[NS][4];                          // per-lane running (sum, sum of squares) of the stored values of channel (tile i, row r)
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[i][r] = 0.f; st2[i][r] = 0.f; }
  const bool want_stats = a.partials != nullptr;
  bf16_t* __restrict__ yg = a.y;
  const float act_slope = a.ep_act == AM_ACT_LRELU ? 0.01f : (a.ep_act == AM_ACT_RELU6 ? 0.f : 1.f);
  const float act_hi = a.ep_act == AM_ACT_RELU6 ? 6.f : __builtin_inff();
  const bool fused = a.ep_scale != nullptr || a.ep_res != nullptr || a.ep_act != AM_ACT_NONE;
  typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  typedef __attribute__((ext_vector_type(8))) float f32x8;

  // per-channel-tile statistics rows: flushed when the workgroup moves to another channel tile and at the end
  auto flush_stats = [&](const int co0) __attribute__((always_inline)) {
    float* part = a.partials + ((size_t)blockIdx.x * 8 + wave) * a.Cout * 2;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(st1[i][r]), s2 = row16_sum(st2[i][r]);
        const int c = co0 + (i >> 1) * 32 + g * 8 + (i & 1) * 4 + r;
        // (own row, one adder per address: deterministic; the atomic executes at the memory side, behind the prologue's zero fill)
        if (r16 == 0 && c < a.Cout) { atomicAdd(part + c * 2, s1); atomicAdd(part + c * 2 + 1, s2); }
        st1[i][r] = 0.f; st2[i][r] = 0.f;
      }
  };

  auto epilogue = [&](const int pc0, const int pc1) __attribute__((always_inline)) {
    Item it; it.c0 = pc0; it.c1 = pc1;
    const int co0 = IT_CO0(it), q0d = IT_Q0D(it), q0h = IT_Q0H(it), q0w = IT_Q0W(it), ib = IT_B(it);
    f32x4 bia[NS], esc[NS], esh[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int co = co0 + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
      bia[i] = (a.bias && co < a.Cout) ? *(const f32x4*)(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
      const bool okc = a.ep_scale && co < a.Cout;
      esc[i] = okc ? *(const f32x4*)(a.ep_scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
      esh[i] = okc ? *(const f32x4*)(a.ep_shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int od = q0d + wave;
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      const int oh = q0h + j, ow = q0w + r16;
      const bool inr = od < a.D && oh < a.H && ow < a.W;
      const size_t ovox = ((size_t)(ib * a.D + od) * a.H + oh) * a.W + ow;
      bf16_t* dstv = yg + ovox * a.Cout + co0 + g * 8;
#pragma unroll
      for (int h = 0; h < NS / 2; ++h) {
        f32x4 o0 = acc[2 * h][j] + bia[2 * h], o1 = acc[2 * h + 1][j] + bia[2 * h + 1];
        bf16_t* dst = dstv + h * 32;
        const bool wr = inr && co0 + h * 32 + g * 8 < a.Cout;
        if (fused) {
          f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
          if (a.ep_res && wr) {
            const f32x8 f = __builtin_convertvector(*(const bfx8*)(a.ep_res + (dst - yg)), f32x8);
            r0 = f32x4{f[0], f[1], f[2], f[3]}; r1 = f32x4{f[4], f[5], f[6], f[7]};
          }
          if (a.ep_scale) { o0 = o0 * esc[2 * h] + esh[2 * h]; o1 = o1 * esc[2 * h + 1] + esh[2 * h + 1]; }
          o0 += r0; o1 += r1;
          if (a.ep_act != AM_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t0 = fmaxf(o0[r], o0[r] * act_slope); o0[r] = t0 > act_hi ? act_hi : t0;
              const float t1 = fmaxf(o1[r], o1[r] * act_slope); o1[r] = t1 > act_hi ? act_hi : t1;
            }
          }
        }
        const bfx4 p0 = __builtin_convertvector(o0, bfx4), p1 = __builtin_convertvector(o1, bfx4);
        const bfx8 pk = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
#ifdef AM_ABLATE
        if (!(a.dbg & 1))
#endif
        if (wr) { if (a.nt_store) __builtin_nontemporal_store(pk, (bfx8*)dst); else *(bfx8*)dst = pk; }
        if (want_stats) {
          const f32x8 s = __builtin_convertvector(pk, f32x8);        // the STORED values
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v0 = wr ? s[r] : 0.f, v1 = wr ? s[4 + r] : 0.f;
            st1[2 * h][r] += v0; st2[2 * h][r] += v0 * v0;
            st1[2 * h + 1][r] += v1; st2[2 * h + 1][r] += v1 * v1;
          }
        }
      }
    }
  };

  // ---- prologue: this wave's statistics row starts at zero; first weights run and first slab
  if (want_stats) {
    float* part = a.partials + ((size_t)blockIdx.x * 8 + wave) * a.Cout * 2;
    for (int c = lane; c < a.Cout * 2; c += 64) part[c] = 0.f;
  }
  if (isX) {
    issue_weights(cur, 0, 0);
    issue_pieces(cur, 0, 0, 17);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!isX) __builtin_amdgcn_s_barrier();               // Y runs one time slot behind X from here on

  int par = 0, sb = 0;                                   // weight slot of the current run, slab buffer of the current item
  int pc0 = 0, pc1 = 0; bool have_prev = false;         // the finished unit whose epilogue is due
  while (true) {
    if (cur.k == 0) {
      if (have_prev) {
        epilogue(pc0, pc1);
        if (want_stats && (pc1 >> 8) != (cur.c1 >> 8)) flush_stats((pc1 >> 8) * 64);
      }
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool has_next = nxt.u < a.nunit;
    int bxc[3];
#pragma unroll
    for (int xw = 0; xw < 3; ++xw) bxc[xw] = bx[xw] + sb * KBRICKB;
#pragma unroll
    for (int run = 0; run < 9; ++run) {
      const int zd = run / 3, xw = run % 3;
      // ---------------- L: fragments of this run from LDS, DMA issue for later runs
      u32x4 brow[VS + 2], af[3][NS];
      const int wbase = KLDS_W + par * KWSLOT + aoff;
#pragma unroll
      for (int r = 0; r < VS + 2; ++r) brow[r] = *(const u32x4*)(lds + bxc[xw] + (zd * KPLANE + r * KEW) * 64);
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int i = 0; i < NS; ++i) af[th][i] = *(const u32x4*)(lds + wbase + th * (NT * 64) + i * 1024);
      int npend = 0;                                     // brick pieces issued after the weights in this L phase (compile time)
      if (isX) {
        if (run < 8) issue_weights(cur, run + 1, par ^ 1);
        else if (has_next) issue_weights(nxt, 0, par ^ 1);
        if (has_next) {
          // 17 pieces per X wave over runs 1..7 (none in run 0: the previous brick's epilogue shares that phase; none in run 8:
          // the slab must have landed when its last M phase ends)
          if (run == 1) issue_pieces(nxt, sb ^ 1, 0, 3);
          else if (run == 2) issue_pieces(nxt, sb ^ 1, 3, 6);
          else if (run == 3) issue_pieces(nxt, sb ^ 1, 6, 9);
          else if (run == 4) issue_pieces(nxt, sb ^ 1, 9, 11);
          else if (run == 5) issue_pieces(nxt, sb ^ 1, 11, 13);
          else if (run == 6) issue_pieces(nxt, sb ^ 1, 13, 15);
          else if (run == 7) issue_pieces(nxt, sb ^ 1, 15, 17);
        }
      }
      npend = run == 0 || run == 8 ? 0 : (run <= 3 ? 3 : 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---------------- M: 48 MFMAs
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#ifdef AM_ABLATE
      if (!(a.dbg & 8))
#endif
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int j = 0; j < VS; ++j)
#pragma unroll
          for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<bf16_t>(af[th][i], brow[j + th], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (isX) {
        // the weights of the next run (issued in this run's L phase, BEFORE its brick pieces) must have landed: all but the npend youngest
        // (no next item: no pieces were issued, the weights are the youngest)
        if (has_next && npend == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (has_next && npend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      par ^= 1;
    }
    sb ^= 1;
    if (cur.k + 1 == a.nslab) { pc0 = cur.c0; pc1 = cur.c1; have_prev = true; }
    if (!has_next) break;
    cur = nxt; advance(nxt);
  }
  epilogue(pc0, pc1);
  if (want_stats) flush_stats((pc1 >> 8) * 64);
  if (isX) __builtin_amdgcn_s_barrier();                // (Y's extra barrier of the prologue)
}

}  // namespace

namespace amconv {

// rows of partial sums a launch of this kernel writes (8 per workgroup), or 0 when the shape does not qualify
static int k3_grid(int B, int D, int H, int W, int Cout, int* units) {
  const int nb = B * (D / KBD) * (H / KBH) * (W / KBW), ny = Cout / 64;
  *units = nb * ny;
  int cus = 256;
  static int cached = 0;
  if (!cached) { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cached = p.multiProcessorCount; else cached = 256; (void)hipGetLastError(); }
  cus = cached;
  return *units < cus ? *units : cus;
}

static bool k3_qualifies(int mode, int dtype, int ksize, int stride, int D, int H, int W, int Cin, int Cout, bool masks) {
  if (dtype != AM_DT_BF16 || ksize != 3 || stride != 1 || (mode != AM_CONV_FWD && mode != AM_CONV_DGRAD) || masks) return false;
  if (Cin % 32 || Cout % 64 || D % KBD || H % KBH || W % KBW) return false;
  return true;
}

int conv_k3_rows(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cin, int Cout, int masks) {
  if (!k3_qualifies(mode, dtype, ksize, stride, D, H, W, Cin, Cout, masks != 0)) return 0;
  int units;
  const int G = k3_grid(B, D, H, W, Cout, &units);
  if (units < K3_MIN_UNITS) return 0;
  return G * 8;
}

int conv_k3_launch(int mode, int dtype, int ksize, int stride, ConvArgs& c, void* stream) {
  if (!k3_qualifies(mode, dtype, ksize, stride, c.Di, c.Hi, c.Wi, c.Cin, c.Cout, c.in_mask.m || c.out_mask.m)) return 0;
  if (c.accumulate || c.nb_x) return 0;
  if (c.Di != c.Do || c.Hi != c.Ho || c.Wi != c.Wo) return 0;
  if ((size_t)(KED + 1) * c.Hi * c.Wi * c.Cin * 2 >= 0x7fffff00ull) return 0;     // the brick's planes must stay below 2 GB (32-bit offsets)
  int units;
  const int G = k3_grid(c.B, c.Di, c.Hi, c.Wi, c.Cout, &units);
  if (units < K3_MIN_UNITS) return 0;
#ifdef AM_ABLATE
  if (getenv("AM_CV_NOK3")) return 0;
#endif
  K3Args a;
  a.x = (const bf16_t*)c.x; a.w = (const bf16_t*)c.w; a.bias = c.bias; a.y = (bf16_t*)c.y; a.partials = c.partials;
  a.ep_scale = c.ep_scale; a.ep_shift = c.ep_shift; a.ep_res = (const bf16_t*)c.ep_res; a.ep_act = c.ep_act;
  a.B = c.B; a.D = c.Di; a.H = c.Hi; a.W = c.Wi; a.Cin = c.Cin; a.Cout = c.Cout; a.Cinp = c.Cinp; a.Coutp = c.Coutp;
  a.nbd = c.Di / KBD; a.nbh = c.Hi / KBH; a.nbw = c.Wi / KBW; a.ny = c.Cout / 64; a.nunit = units; a.nslab = c.Cinp / 32;
  a.flip = mode == AM_CONV_DGRAD;
  a.nt_store = (size_t)c.B * c.Do * c.Ho * c.Wo * c.Cout * 2 >= ((size_t)384 << 20);
  a.w_bytes = (unsigned)c.w_bytes;
#ifdef AM_ABLATE
  { const char* e = getenv("AM_K3_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  auto kern = conv_k3_kernel<4>;
  static PerDeviceOnce lds_cap;
  lds_cap.run([&](int) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); });
  AM_LAUNCH(kern, dim3(G), dim3(512), KLDS, (hipStream_t)stream, a);
  AM_CHECK_LAUNCH();
  return 1;
}

}  // namespace amconv
