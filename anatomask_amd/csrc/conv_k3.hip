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
//   * between two barriers (one per run and wave) X runs [fragment reads + DMA issue, 48 MFMAs] and Y [48 MFMAs, fragment reads + DMA issue].
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
constexpr int KLDS_TAB = KLDS_W + 2 * KWSLOT;     // 162 816: per-channel epilogue constants of the workgroup's channel tile (bias | scale | shift, 64 floats each)
constexpr int KLDS = KLDS_TAB + 768;              // 163 584 of 163 840
#ifndef AM_K3_MIN_UNITS
#define AM_K3_MIN_UNITS 256
#endif
constexpr int K3_MIN_UNITS = AM_K3_MIN_UNITS;     // below one unit per CU the brick kernel of conv_igemm.hip fills the chip better (A/B at B=4, tools/mkvariant.sh ... -DAM_K3_MIN_UNITS=512 vs 256: 512->512 @16^3 1 283-1 309 -> 1 479-1 483 TFLOP/s)

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

// wave-uniform values the compiler cannot PROVE uniform (results of the vector ALU's integer division, 64-bit offset chains) must be made
// provably so before they enter a buffer descriptor: otherwise every buffer operation is wrapped in a "waterfall" loop (v_readfirstlane x 4,
// compare, s_and_saveexec, op, loop: cdna_hip_programming.md T20 -- the deferred stores took ~800 cycles each that way)
__device__ __forceinline__ int k3_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P> __device__ __forceinline__ P* k3_uni_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}

// (unit, slab) + the unit's brick / channel tile, packed (all wave-uniform: these live in SGPRs across the whole walk)
struct Item { int u, k, c0, c1, p; };               // c0 = bw | bh << 8 | bd << 16 (brick indices), c1 = b | ytile << 8; p (transposed conv): output parity class pd << 2 | ph << 1 | pw
#define IT_Q0W(it) (((it).c0 & 255) * KBW)
#define IT_Q0H(it) ((((it).c0 >> 8) & 255) * KBH)
#define IT_Q0D(it) ((((it).c0 >> 16) & 255) * KBD)
#define IT_B(it) ((it).c1 & 255)
#define IT_CO0(it) (((it).c1 >> 8) * NT)

// EP: the fused store epilogue (eval-mode BatchNorm scale / shift, skip tensor, activation: the teacher's decoder) is its own instantiation --
// its scale / shift registers would push the plain kernel over 256.  ST: the statistics rows (16 more registers), likewise
// CT: ConvTranspose3d k4 s2 p1 (P/decoder3D.py:17 `up_sample`) on the same skeleton.  Output voxel 2 q + p of parity class p = (pd, ph, pw)
// is a 2 x 2 x 2-tap convolution over the coarse grid -- per axis the shifts {p - 1, p} of the k3 window, kernel index 3 - 2 z + p for window
// position z in {p, p + 1} -- so the SAME haloed 10 x 6 x 18 source brick serves all eight classes: a workgroup walks (brick, channel tile)
// units and, inside one, the classes x slabs; a "run" is one (d, w) shift with its TWO h-taps (5 shared fragment rows, 32 MFMAs), four
// runs per (class, slab).  With two slabs (Cin = 64) both stay resident in the two slab buffers for all eight classes: the brick is
// fetched ONCE for 64 taps (conv_igemm.hip gives every class its own workgroup and its own copy: that launch ran at the CU's fill rate).
// A class's 8 x 4 x 16 outputs leave as 16-byte stores two voxels apart (the other classes' voxels lie between them).
// S1 (with ST): the caller reads only the SUM column of the statistics rows (the data gradient whose per-channel sum is the bias gradient of
// the transposed conv in front, P/decoder3D.py:17): the sum-of-squares half of the epilogue (a multiply, a DPP exchange and an add per stored
// value, 8 registers) is not compiled in; that column is written as zero.
template <int NS, bool EP, bool ST, bool CT = false, bool S1 = false>
__global__ __launch_bounds__(512, 2) void conv_k3_kernel(K3Args a) {
  static_assert(NS == 4 || NS == 2, "64- or 32-channel output tiles (NS = 2: the decoder's last conv, C -> C / 2 = 32 at STUNet-B; a wave then owns 8 accumulator tiles)");
  constexpr int VS = 4, NT = 16 * NS;
  constexpr int NCH = VS * NS / 2;                  // 16-byte store chunks per lane and unit
  constexpr int NRUN = CT ? 4 : 9;                  // runs per (unit, slab) item; taps per run: 2 / 3
  constexpr int NTR = CT ? 2 : 3;
  static_assert(!CT || (NS == 4 && !EP && !ST), "transposed conv: 64-channel tiles, plain epilogue (bias)");
  static_assert(!S1 || (ST && NS == 4 && !CT), "sum-only statistics: a variant of the statistics instantiation");
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
    it.c0 = k3_uni(bw_ | (bh_ << 8) | (bd_ << 16)); it.c1 = k3_uni((br / a.nbd) | (yt << 8));
  };
  auto advance = [&](Item& it) {
    if (it.k + 1 < a.nslab) { ++it.k; return; }
    it.k = 0;
    if (CT && it.p < 7) { ++it.p; return; }           // (transposed conv: the eight parity classes of the brick, then the next brick)
    it.p = 0; it.u += G;
    if (it.u < a.nunit) decode(it);
  };
  Item cur; cur.u = gperm; cur.k = 0; cur.p = 0;
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

  // source brick, DMA source: piece p = wave + 8 k covers rows 16 p .. 16 p + 15 (the last one, p = 67, rows 1064 .. 1079); waves 0-3 own
  // 9 pieces, waves 4-7 own 8; lane -> (row, chunk position).  Per piece and lane: the byte offset from the brick's first haloed voxel and
  // a 6-bit code of the brick faces the row lies on.
  unsigned pl[9], pbits[2] = {0u, 0u};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int p = wave + 8 * k;
    const int row0 = p * 16 < KNROW - 16 ? p * 16 : KNROW - 16;
    const int rho = row0 + (lane >> 2);
    const int z = rho / KPLANE, rem = rho - z * KPLANE, yy = rem / KEW, xx = rem - yy * KEW;
    pl[k] = (unsigned)((((z * a.H + yy) * a.W + xx) * a.Cin + (((lane & 3) ^ (((xx >> 2) & 1) * 2)) * 8)) * 2);
    const unsigned code = (z == 0 ? 1u : 0u) | (z == KED - 1 ? 2u : 0u) | (yy == 0 ? 4u : 0u) | (yy == KEH - 1 ? 8u : 0u) | (xx == 0 ? 16u : 0u) | (xx == KEW - 1 ? 32u : 0u);
    pbits[k / 5] |= code << (6 * (k % 5));
  }

  // ---- DMA issue (weights: X waves; brick pieces: Y waves)
  auto issue_weights = [&](const Item& it, const int run, const int slot) __attribute__((always_inline)) {
    // run = zd * 3 + xw; its taps th = 0..2 are the h-shifts; X wave wq brings cout tile wq of each tap
    // (transposed conv: run = 2 (zd - pd) + (xw - pw) over window positions z in {p, p + 1} per axis, two h-taps; kernel index 3 - 2 z + p)
    const int pd_ = (it.p >> 2) & 1, ph_ = (it.p >> 1) & 1, pw_ = it.p & 1;
    const int zd = CT ? pd_ + (run >> 1) : run / 3, xw = CT ? pw_ + (run & 1) : run - (run / 3) * 3;
    if (wq >= NS) return;                              // (NS = 2: waves 0, 1 bring the two cout tiles)
#pragma unroll
    for (int th = 0; th < NTR; ++th) {
      const int t = zd * 9 + th * 3 + xw;
      const int widx = CT ? ((3 - 2 * zd + pd_) * 4 + (3 - 2 * (ph_ + th) + ph_)) * 4 + (3 - 2 * xw + pw_) : (a.flip ? 26 - t : t);
      const unsigned so = (unsigned)(widx * wtapB + ((IT_CO0(it) + (wq >> 1) * 32 + (wq & 1) * 4) * a.Cinp + it.k * 32) * 2);
#ifdef AM_ABLATE
      if (a.dbg & 4) continue;
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, K3_LDSP(KLDS_W + slot * KWSLOT + th * (NT * 64) + wq * 1024), 16, wlane, so, 0, 0);
    }
  };
  // brick pieces [k0, k1) of this wave for item `it` into slab buffer `sbuf`.  Everything about a piece that depends on the lane only was
  // computed once (pl[], pbits[]); per piece: 3 vector instructions.
  auto issue_pieces = [&](const Item& it, const int sbuf, const int k0, const int k1) __attribute__((always_inline)) {
    const int q0d = IT_Q0D(it), q0h = IT_Q0H(it), q0w = IT_Q0W(it);
    // which faces of the haloed brick lie outside the volume (D % 8 == H % 4 == W % 16 == 0): bit 0 z == 0, 1 z == 9, 2 y == 0, 3 y == 5, 4 x == 0, 5 x == 17
    const unsigned m = (q0d == 0 ? 1u : 0u) | (q0d + KBD == a.D ? 2u : 0u) | (q0h == 0 ? 4u : 0u) | (q0h + KBH == a.H ? 8u : 0u) |
                       (q0w == 0 ? 16u : 0u) | (q0w + KBW == a.W ? 32u : 0u);
    // descriptor anchored at the brick's first haloed voxel (before the tensor for the first brick: those lanes are out of range anyway)
    const long long org = ((((long long)IT_B(it) * a.D + (q0d - 1)) * a.H + (q0h - 1)) * a.W + (q0w - 1)) * (long long)a.Cin;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)k3_uni_ptr(a.x + org), 0, 0x7fffff00, 0x00020000);
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const unsigned bad = pbits[k / 5] & (m << (6 * (k % 5)));
      unsigned vo = bad ? OOB : pl[k];
#ifdef AM_ABLATE
      if (a.dbg & 2) vo = OOB;
#endif
      const int p = wave + 8 * k;
      const int row0 = p * 16 < KNROW - 16 ? p * 16 : KNROW - 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, K3_LDSP(sbuf * KBRICKB + row0 * 64), 16, vo, (unsigned)(it.k * 64), 0, 0);
    }
  };

  // ---- epilogue state
  f32x4 acc[NS][VS];
  // per-lane running (sum, sum of squares) of the stored values, FOLDED over lane pairs: the even lane of a pair holds channel (tile 2h, row r),
  // the odd lane channel (tile 2h + 1, row r), each summed over both lanes' voxels -- 16 registers instead of 32 (the unfolded form spilled)
  float f1[NS / 2][4], f2[NS / 2][4];
#pragma unroll
  for (int h = 0; h < NS / 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) { f1[h][r] = 0.f; f2[h][r] = 0.f; }
  const unsigned oddm = (lane & 1) ? 0xffffffffu : 0u;
#ifdef AM_ABLATE
  const bool want_stats = ST && a.partials != nullptr && !(a.dbg & 32);      // (32: stamps without the statistics work)
#else
  const bool want_stats = ST && a.partials != nullptr;
#endif
  bf16_t* __restrict__ yg = a.y;
  const float act_slope = a.ep_act == AM_ACT_LRELU ? 0.01f : (a.ep_act == AM_ACT_RELU6 ? 0.f : 1.f);
  const float act_hi = a.ep_act == AM_ACT_RELU6 ? 6.f : __builtin_inff();
  constexpr bool fused = EP;
  typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  typedef __attribute__((ext_vector_type(8))) float f32x8;

  // the wave's statistics row, written once at the end of the walk (the channel tile of a workgroup never changes)
  auto flush_stats = [&](const int co0) __attribute__((always_inline)) {
    float* part = a.partials + ((size_t)blockIdx.x * 8 + wave) * a.Cout * 2;
#define K3_DPP_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xF, 0xF, true))
#pragma unroll
    for (int h = 0; h < NS / 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = f1[h][r], s2 = f2[h][r];
        // sum over the 8 lanes of the 16-lane row that have this lane's parity: xor 2, rotate by 4, rotate by 8
        K3_DPP_ADD(s1, 0x4E); K3_DPP_ADD(s1, 0x124); K3_DPP_ADD(s1, 0x128);
        if constexpr (!S1) { K3_DPP_ADD(s2, 0x4E); K3_DPP_ADD(s2, 0x124); K3_DPP_ADD(s2, 0x128); }
        const int c = co0 + h * 32 + g * 8 + (r16 & 1) * 4 + r;     // (even lane: tile 2h, odd lane: tile 2h + 1)
        if (r16 < 2) { part[c * 2] = s1; part[c * 2 + 1] = s2; }
      }
#undef K3_DPP_ADD
  };

  // D % 8 == H % 4 == W % 16 == Cout % 64 == 0 (k3_qualifies): every voxel / channel of a unit is in range, so the stores are unpredicated
  // buffer stores -- descriptor anchored at the wave's first output voxel (scalar), lane constant offset, per-row scalar offset: no
  // vector address arithmetic at all (the first version spent ~3 000 cycles per unit and wave half on 64-bit address math and store predicates).
  // The stores themselves are DEFERRED: a finished unit is converted to its stored form (8 x 16 bytes per lane, `pk`) in the phase that
  // opens the next unit, and its 8 store instructions (+ the statistics of those values) go out one per L phase over the next unit's first
  // 8 runs -- issued back to back they drain at ~12 B/clk/CU (2 700 cycles per unit and wave half, with the other half's MFMAs waiting at
  // the barrier: 13 % of the kernel).
  // 64-channel tiles: a voxel's 128 bytes are one cache line whose two 64-byte halves (channel groups h = 0, 1) sit in the SAME lanes'
  // registers pk[j][0], pk[j][1] -- stored as they are, every store instruction would write HALF of 16 lines (round 4: 5.05 GB written for a
  // 4.29 GB output).  The lanes of a 16-lane row swap halves through one DPP rotation (row_ror:8) per register, so that store h of a row
  // writes the 8 voxels 8 h .. 8 h + 7 COMPLETELY: lane (r16, g) sends chunk 4 (r16 >> 3) + g of voxel 8 h + (r16 & 7).
  constexpr bool FULL = NS == 4;
  const bool lo8 = r16 < 8;
  const unsigned olane = FULL ? (unsigned)(((r16 & 7) * (CT ? 2 : 1) * a.Cout + (r16 >> 3) * 32 + g * 8) * 2)
                              : (unsigned)((r16 * (CT ? 2 : 1) * a.Cout + g * 8) * 2);      // (transposed conv: a class's voxels are two apart)
  const unsigned ohalf = (unsigned)(8 * (CT ? 2 : 1) * a.Cout * 2);                      // FULL: voxels 8 .. 15 of the row (store h = 1)
  const unsigned olane_own = (unsigned)((r16 * (CT ? 2 : 1) * a.Cout + g * 8) * 2);      // the lane's OWN voxel r16, chunk g (+ 64 h): the skip tensor is read in accumulator layout
  const unsigned orow = (unsigned)((CT ? 4 : 1) * a.W * a.Cout * 2);                 // (... and two rows of the 2 W wide fine grid)
  u32x4 pk[VS][NS / 2];
  long long pk_off = 0;                                  // element offset of the pending unit's first output voxel of this wave (wave-uniform)
  auto finish_unit = [&](const int pc0, const int pc1, const int pcls) __attribute__((always_inline)) {
    Item it; it.c0 = pc0; it.c1 = pc1;
    const int co0 = IT_CO0(it);
    if constexpr (CT)                                      // fine voxel 2 q + p of the 2D x 2H x 2W output
      pk_off = ((((long long)IT_B(it) * (2 * a.D) + (2 * (IT_Q0D(it) + wave) + ((pcls >> 2) & 1))) * (2 * a.H) + (2 * IT_Q0H(it) + ((pcls >> 1) & 1))) * (2 * a.W) +
                (2 * IT_Q0W(it) + (pcls & 1))) * (long long)a.Cout + co0;
    else
    pk_off = ((((long long)IT_B(it) * a.D + (IT_Q0D(it) + wave)) * a.H + IT_Q0H(it)) * a.W + IT_Q0W(it)) * (long long)a.Cout + co0;
    // per-channel constants from the LDS table (global loads here would put a `vmcnt` wait into this phase)
    f32x4 bia[NS], esc[NS], esh[NS];
    if (a.bias) {
#pragma unroll
      for (int i = 0; i < NS; ++i) bia[i] = *(const f32x4*)(lds + KLDS_TAB + ((i >> 1) * 32 + g * 8 + (i & 1) * 4) * 4);
    }
    if (EP && a.ep_scale) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        esc[i] = *(const f32x4*)(lds + KLDS_TAB + 256 + ((i >> 1) * 32 + g * 8 + (i & 1) * 4) * 4);
        esh[i] = *(const f32x4*)(lds + KLDS_TAB + 512 + ((i >> 1) * 32 + g * 8 + (i & 1) * 4) * 4);
      }
    }
    if (EP && a.ep_res) {                                  // the skip tensor's chunks land in pk and are replaced by the results
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)k3_uni_ptr(a.ep_res + pk_off), 0, 0x7fffff00, 0x00020000);
#pragma unroll
      for (int j = 0; j < VS; ++j)
#pragma unroll
        for (int h = 0; h < NS / 2; ++h) pk[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rr, olane_own, j * orow + h * 64, 0);
    }
#pragma unroll
    for (int j = 0; j < VS; ++j) {
#pragma unroll
      for (int h = 0; h < NS / 2; ++h) {
        f32x4 o0 = acc[2 * h][j], o1 = acc[2 * h + 1][j];
        if (a.bias) { o0 += bia[2 * h]; o1 += bia[2 * h + 1]; }
        if (fused) {
          if (a.ep_scale) { o0 = o0 * esc[2 * h] + esh[2 * h]; o1 = o1 * esc[2 * h + 1] + esh[2 * h + 1]; }
          if (a.ep_res) {
            const f32x8 f = __builtin_convertvector(__builtin_bit_cast(bfx8, pk[j][h]), f32x8);
            o0 += f32x4{f[0], f[1], f[2], f[3]}; o1 += f32x4{f[4], f[5], f[6], f[7]};
          }
          if (a.ep_act != AM_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t0 = fmaxf(o0[r], o0[r] * act_slope); o0[r] = t0 > act_hi ? act_hi : t0;
              const float t1 = fmaxf(o1[r], o1[r] * act_slope); o1[r] = t1 > act_hi ? act_hi : t1;
            }
          }
        }
        const bfx4 p0 = __builtin_convertvector(o0, bfx4), p1 = __builtin_convertvector(o1, bfx4);
        pk[j][h] = __builtin_bit_cast(u32x4, __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7));
        acc[2 * h][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * h + 1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  // statistics of half a chunk (words q and 2 + q of the 16 stored bytes): words 0, 1 = channels 0..3 of tile 2h (rows r), words 2, 3 = those of
  // tile 2h + 1.  A lane keeps the tile of its parity and gives the other one to its pair lane (quad_perm [1, 0, 3, 2]).  Bitwise selects on the
  // packed words (v_bfi_b32): a `odd ? v[4 + r] : v[r]` on the unpacked vector becomes a dynamic element index -- an 8-way compare / select chain.
  auto stats_half = [&](const int c, const int q) __attribute__((always_inline)) {
    const int j = c / (NS / 2), h = c % (NS / 2);
    const u32x4 wv = pk[j][h];
    const unsigned kw = (wv[2 + q] & oddm) | (wv[q] & ~oddm), gw = (wv[q] & oddm) | (wv[2 + q] & ~oddm);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 2 * q + e;
      const float keep = __uint_as_float(e ? (kw & 0xffff0000u) : (kw << 16)), give = __uint_as_float(e ? (gw & 0xffff0000u) : (gw << 16));
      f1[h][r] += keep;
      f1[h][r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
      if constexpr (!S1) {
        f2[h][r] += keep * keep;
        const float g2 = give * give;
        f2[h][r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g2), 0xB1, 0xF, 0xF, true));
      }
    }
  };
  // Units of two or more slabs do the statistics of a finished unit as 16 half-chunk pieces, one per L phase over the next unit's first 16
  // runs, instead of one whole chunk behind each of the 8 deferred stores (the L phases that carry a store are the long ones).  Same-box A/B
  // (profiles/r06_experiments.md): 64 -> 64 @128^3 with statistics 5.755 -> 5.713 ms; -DAM_K3_NO_SPREAD_STATS builds the old placement.
#ifdef AM_K3_NO_SPREAD_STATS
  constexpr bool spread_stats = false;
#else
  const bool spread_stats = ST && !CT && NCH == 8 && a.nslab >= 2;
#endif
  // chunk c = (NS / 2) j + h of the pending unit: one 16-byte store per lane (+ the statistics of the stored values)
  auto store_chunk = [&](const int c, const bool tail) __attribute__((always_inline)) {
    const int j = c / (NS / 2), h = c % (NS / 2);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)k3_uni_ptr(yg + pk_off), 0, 0x7fffff00, 0x00020000);
#ifdef AM_ABLATE
    if (!(a.dbg & 1))
#endif
    {
      u32x4 sv = pk[j][h];
      unsigned so_ = (unsigned)(j * orow + h * 64);
      if constexpr (FULL) {
        // what the partner lane (r16 ^ 8) holds of the OTHER half: store 0 takes its h = 1 chunk of voxels 0 .. 7, store 1 its h = 0 chunk of 8 .. 15
        const u32x4 ot = pk[j][h ^ 1];
        u32x4 rx_;
#pragma unroll
        for (int e = 0; e < 4; ++e) rx_[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ot[e], 0x128, 0xF, 0xF, false);
        const bool own = h == 0 ? lo8 : !lo8;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = own ? sv[e] : rx_[e];
        so_ = (unsigned)(j * orow) + (h ? ohalf : 0u);
      }
      if (a.nt_store) __builtin_amdgcn_raw_buffer_store_b128(sv, ry, olane, so_, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(sv, ry, olane, so_, 0);
      // A 16-byte store's data registers must not be rewritten in the next cycles.  hipcc inserts the wait states only for stores whose
      // soffset is an immediate; with the row offset in an SGPR (chunks 2..5) it emits none, and the TAIL flush -- convert, store, convert
      // into the same registers -- stored corrupted values in ~7 % of those chunks (found by tools/k3_check.py: NaNs in the last unit of
      // every workgroup, rows 1 and 2 of each plane, plain instantiation only: the others have statistics code between two stores).
      // (the asm holds the data registers live across the wait states: nothing can be scheduled into them in between)
      // (FULL: the store's data is a temporary -- the select of own / partner words -- whose registers the next chunk's select may take at once;
      // the transposed instantiation issues two stores back to back: the same hazard in every phase, not only in the tail)
      if (tail || FULL) asm volatile("s_nop 4" : "+v"(sv) :: "memory");
    }
    if (want_stats && (tail || !spread_stats)) { stats_half(c, 0); stats_half(c, 1); }
  };

  // ---- prologue: this wave's statistics row starts at zero; first weights run and first slab
  if (want_stats && a.ny > 1) {                          // (channels of the other tiles: this workgroup contributes nothing to them)
    float* part = a.partials + ((size_t)blockIdx.x * 8 + wave) * a.Cout * 2;
    for (int c = lane; c < a.Cout * 2; c += 64) part[c] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid < NT) {                                        // (the channel tile of a workgroup never changes: the grid is a multiple of ny)
    const int c = IT_CO0(cur) + tid;
    ((float*)(lds + KLDS_TAB))[tid] = a.bias ? a.bias[c] : 0.f;
    ((float*)(lds + KLDS_TAB))[64 + tid] = a.ep_scale ? a.ep_scale[c] : 1.f;
    ((float*)(lds + KLDS_TAB))[128 + tid] = a.ep_scale ? a.ep_shift[c] : 0.f;
  }
  if (isX) { issue_weights(cur, 0, 0); issue_pieces(cur, 0, 0, 9); }
  else issue_pieces(cur, 0, 0, 8);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#ifdef AM_ABLATE
  // diagnostic stamps (AM_K3_DBG & 16): cycles this wave spends in its L phases, waiting at the barrier behind them, in its M phases and
  // at the barrier behind those, summed over the walk; written as raw integers into the wave's statistics row
  unsigned tL = 0, tWL = 0, tM = 0, tWM = 0, tE = 0, t_prev = 0;
  unsigned long long rt0 = 0;
  const bool stamps = (a.dbg & 16) != 0;
#define K3_STAMP(ACC) if (stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ACC += (unsigned)t_ - t_prev; t_prev = (unsigned)t_; }
  if (stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); t_prev = (unsigned)t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory"); }
#else
#define K3_STAMP(ACC)
#endif
  int par = 0, sb = 0;                                   // weight slot of the current run, slab buffer of the current item
  int pc0 = 0, pc1 = 0, pcls = 0; bool have_prev = false;   // the finished unit (transposed conv: and class) whose epilogue is due
  bool spend = false;                                    // its 8 stores are pending: one per L phase of this slab's runs 0..7
  bool pend_stats = false;                               // (AM_K3_SPREAD_STATS) its statistics are pending: half a chunk per L phase of the first two slabs' runs
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  while (true) {
    spend = false;
    if (cur.k == 0 && have_prev) {
      finish_unit(pc0, pc1, pcls);                       // (also zeroes the accumulators)
      spend = true; pend_stats = true;
      K3_STAMP(tE);
    } else if (cur.k >= 2) pend_stats = false;
    const bool has_next = nxt.u < a.nunit;
    int bxc[3];
#pragma unroll
    for (int xw = 0; xw < 3; ++xw) bxc[xw] = bx[xw] + sb * KBRICKB;
    // (transposed conv) does the next item's slab have to be fetched?  With two slabs both stay in the two buffers for the brick's eight
    // classes (slab k in buffer k: sixteen items per brick keep the parity); only a new brick brings new bytes
    const bool fetch_next = has_next && (!CT || a.nslab != 2 || nxt.p == 0);
    const int cpd = (cur.p >> 2) & 1, cph = (cur.p >> 1) & 1, cpw = cur.p & 1;
#pragma unroll
    for (int run = 0; run < NRUN; ++run) {
      const int zd = run / 3, xw = run % 3;              // (k3; the transposed conv's window position is runtime: class parity + run bits)
      // ---------------- L: fragments of this run from LDS, DMA issue for later runs
      u32x4 brow[VS + 2], af[3][NS];
      const int wbase = KLDS_W + par * KWSLOT + aoff;
      if constexpr (CT) {
        const int zq = cpd + (run >> 1), xq = cpw + (run & 1);
        const int bsel = (xq == 0 ? bxc[0] : (xq == 1 ? bxc[1] : bxc[2])) + (zq * KPLANE + cph * KEW) * 64;
#pragma unroll
        for (int r = 0; r < VS + 1; ++r) brow[r] = *(const u32x4*)(lds + bsel + r * KEW * 64);
      } else {
#pragma unroll
      for (int r = 0; r < VS + 2; ++r) brow[r] = *(const u32x4*)(lds + bxc[xw] + (zd * KPLANE + r * KEW) * 64);
      }
#pragma unroll
      for (int th = 0; th < NTR; ++th)
#pragma unroll
        for (int i = 0; i < NS; ++i) af[th][i] = *(const u32x4*)(lds + wbase + th * (NT * 64) + i * 1024);
      // DMA issue.  X: W(run + 1) into the slot Y finished reading one slot ago (it has this phase and the M phase that follows to land),
      // then its share of the next slab (9 pieces over runs 0..7), then a deferred store.  Y: a deferred store, then its share of the next
      // slab (8 pieces over runs 0..6; retired at the end of M(7): X reads that slab two time slots later).
      // (transposed conv, 4 runs per item: X 3 + 3 + 3 pieces over runs 0..2, Y 4 + 4 over runs 0, 1; the 8 deferred stores two per run)
      const int pk0 = CT ? (isX ? 3 * run : 4 * run) : (run == 0 ? 0 : (run == 1 ? 1 : run + 1));          // first piece of this phase: 0 | 1 2 | 3 | 4 | ...
      const int npx = CT ? (run < 3 ? 3 : 0) : (run == 1 ? 2 : (run < 8 ? 1 : 0));                 // pieces X issues in this phase (9 in all, the last in run 7)
      const int npy = CT ? (run < 2 ? 4 : 0) : (run == 1 ? 2 : (run < 7 ? 1 : 0));                 // pieces Y issues (8 in all, the last in run 6)
      constexpr int SPR = CT ? 2 : 1;                                   // deferred stores per L phase
      if (isX) {
        if (run < NRUN - 1) issue_weights(cur, run + 1, par ^ 1);
        else if (has_next) issue_weights(nxt, 0, par ^ 1);
        if (fetch_next && npx > 0) issue_pieces(nxt, sb ^ 1, CT ? 3 * run : pk0, (CT ? 3 * run : pk0) + npx);
        if (spend && run * SPR < NCH) {                  // (behind the weights: the counted wait below leaves them in flight)
#pragma unroll
          for (int q = 0; q < SPR; ++q) store_chunk(run * SPR + q, false);
        }
      } else {
        if (spend && run * SPR < NCH) {
#pragma unroll
          for (int q = 0; q < SPR; ++q) store_chunk(run * SPR + q, false);
        }
        if (fetch_next && npy > 0) issue_pieces(nxt, sb ^ 1, CT ? 4 * run : pk0, (CT ? 4 * run : pk0) + npy);
      }
      if (spread_stats && want_stats && pend_stats) {
        // (k3: 9 runs per slab; half-chunks 0..8 in the first slab, 9..15 in the second.  `run` is a constant of the unrolled loop: each branch
        // indexes the register arrays with compile-time values)
        if (cur.k == 0) stats_half(run >> 1, run & 1);
        else if (cur.k == 1 && 9 + run < 16) stats_half((9 + run) >> 1, (9 + run) & 1);
      }
      // ONE barrier per run and wave.  Between two barriers X runs [L(n) M(n)] and Y runs [M(n - 1) L(n)]: X fetches while Y multiplies, then
      // the other way round.  What the barrier orders: every X wave's weights of run n + 1 have landed (its counted `vmcnt` wait sits in front
      // of it) before anyone reads them; Y's reads of the weight slot of run n -- and, at a slab's end, of the slab buffer -- are complete
      // (its `lgkmcnt(0)` sits in front of it) before X's DMA refills them in the next interval.  X's own reads are consumed by its MFMAs.
      K3_STAMP(tL);
      if (!isX) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        K3_STAMP(tWL);
      }
      // ---------------- M: 48 MFMAs
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#ifdef AM_ABLATE
      if (!(a.dbg & 8))
#endif
#pragma unroll
      for (int th = 0; th < NTR; ++th)
#pragma unroll
        for (int j = 0; j < VS; ++j)
#pragma unroll
          for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<bf16_t>(af[th][i], brow[j + th], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // X: the weights of the next run (issued in this run's L phase) must have landed; Y: the next slab, before the barrier that ends M(7)
      // X: the weights of the next run must have landed -- everything but what it issued behind them in this run's L phase (pieces, a
      // deferred store); at the end of M(8) the whole next slab.  Y: at the end of M(7) its share of the next slab (a store issued in L(7) is
      // the youngest operation and stays in flight).
      if (isX) {
        const int nbehind = (fetch_next ? npx : 0) + ((spend && run * SPR < NCH) ? SPR : 0);
        if (nbehind == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nbehind == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (nbehind == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (nbehind == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (nbehind == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else if (run == NRUN - 2) {                      // (Y's last pieces went out a run or more ago; the stores of this run's L phase are the youngest operations)
        if (spend && run * SPR < NCH) { if (SPR == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      K3_STAMP(tM);
      if (isX) {
        __builtin_amdgcn_s_barrier();
        K3_STAMP(tWM);
      }
      par ^= 1;
    }
    sb ^= 1;
    if (cur.k + 1 == a.nslab) { pc0 = cur.c0; pc1 = cur.c1; pcls = cur.p; have_prev = true; }
    if (!has_next) break;
    cur = nxt; advance(nxt);
  }
  finish_unit(pc0, pc1, pcls);
#pragma unroll
  for (int c = 0; c < NCH; ++c) store_chunk(c, true);
  if (want_stats) flush_stats((pc1 >> 8) * NT);
#ifdef AM_ABLATE
  if (stamps && a.partials) {
    K3_STAMP(tE);
    unsigned long long rt1; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
    unsigned* row = (unsigned*)(a.partials + ((size_t)blockIdx.x * 8 + wave) * a.Cout * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) { row[0] = tL; row[1] = tWL; row[2] = tM; row[3] = tWM; row[4] = tE; row[5] = (unsigned)(rt1 - rt0); }
  }
#endif
}

}  // namespace

namespace amconv {

// rows of partial sums a launch of this kernel writes (8 per workgroup), or 0 when the shape does not qualify
static int k3_grid(int B, int D, int H, int W, int Cout, int* units) {
  const int nb = B * (D / KBD) * (H / KBH) * (W / KBW), ny = Cout % 64 ? Cout / 32 : Cout / 64;      // (32-channel tiles for Cout = 32, 96, 160)
  *units = nb * ny;
  int cus = 256;
  static int cached = 0;
  if (!cached) { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cached = p.multiProcessorCount; else cached = 256; (void)hipGetLastError(); }
  cus = cached;
  // a multiple of the channel-tile count: a workgroup's units (u, u + G, ...) then all have the same channel tile
  const int G = *units < cus ? *units : cus;
  return G - G % ny > 0 ? G - G % ny : ny;
}

static bool k3_qualifies(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cin, int Cout, bool masks) {
  if (dtype != AM_DT_BF16 || ksize != 3 || stride != 1 || (mode != AM_CONV_FWD && mode != AM_CONV_DGRAD) || masks) return false;
  if (Cin % 32 || Cout % 32 || D % KBD || H % KBH || W % KBW) return false;
  // `Item` packs the sample index and the three brick indices into 8-bit fields (IT_B / IT_Q0*): larger launches go to conv_igemm.hip
  if (B > 255 || D / KBD > 255 || H / KBH > 255 || W / KBW > 255) return false;
  return true;
}

int conv_k3_rows(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cin, int Cout, int masks) {
  if (!k3_qualifies(mode, dtype, ksize, stride, B, D, H, W, Cin, Cout, masks != 0)) return 0;
  int units;
  const int G = k3_grid(B, D, H, W, Cout, &units);
  if (units < K3_MIN_UNITS) return 0;
  return G * 8;
}

int conv_k3_launch(int mode, int dtype, int ksize, int stride, ConvArgs& c, void* stream) {
  if (!k3_qualifies(mode, dtype, ksize, stride, c.B, c.Di, c.Hi, c.Wi, c.Cin, c.Cout, c.in_mask.m || c.out_mask.m)) return 0;
  if (c.accumulate || c.nb_x) return 0;
  if (c.Di != c.Do || c.Hi != c.Ho || c.Wi != c.Wo) return 0;
  if ((size_t)(KED + 1) * c.Hi * c.Wi * c.Cin * 2 >= 0x7fffff00ull) return 0;     // the brick's planes must stay below 2 GB (32-bit offsets)
  int units;
  const int G = k3_grid(c.B, c.Di, c.Hi, c.Wi, c.Cout, &units);
  if (units < K3_MIN_UNITS) return 0;
#ifdef AM_ABLATE
  if (getenv("AM_CV_NOK3")) return 0;
  { const char* e_ = getenv("AM_K3_NO32"); if (e_ && atoi(e_) && c.Cout % 64) return 0; }
#endif
  K3Args a;
  a.x = (const bf16_t*)c.x; a.w = (const bf16_t*)c.w; a.bias = c.bias; a.y = (bf16_t*)c.y; a.partials = c.partials;
  a.ep_scale = c.ep_scale; a.ep_shift = c.ep_shift; a.ep_res = (const bf16_t*)c.ep_res; a.ep_act = c.ep_act;
  a.B = c.B; a.D = c.Di; a.H = c.Hi; a.W = c.Wi; a.Cin = c.Cin; a.Cout = c.Cout; a.Cinp = c.Cinp; a.Coutp = c.Coutp;
  a.nbd = c.Di / KBD; a.nbh = c.Hi / KBH; a.nbw = c.Wi / KBW; a.ny = c.Cout % 64 ? c.Cout / 32 : c.Cout / 64; a.nunit = units; a.nslab = c.Cinp / 32;
  a.flip = mode == AM_CONV_DGRAD;
  a.nt_store = (size_t)c.B * c.Do * c.Ho * c.Wo * c.Cout * 2 >= ((size_t)384 << 20);
  a.w_bytes = (unsigned)c.w_bytes;
#ifdef AM_ABLATE
  { const char* e = getenv("AM_K3_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  const bool ep = a.ep_scale || a.ep_res || a.ep_act != AM_ACT_NONE;
  if (ep && a.partials) return 0;                  // (no caller fuses a store epilogue AND asks for statistics: conv_igemm.hip serves it)
  auto kern = ep ? conv_k3_kernel<4, true, false> : (a.partials ? conv_k3_kernel<4, false, true> : conv_k3_kernel<4, false, false>);
  if (c.Cout % 64) kern = ep ? conv_k3_kernel<2, true, false> : (a.partials ? conv_k3_kernel<2, false, true> : conv_k3_kernel<2, false, false>);
  else if (a.partials && !ep && c.stats_sum_only) kern = conv_k3_kernel<4, false, true, false, true>;      // (the rows' sum-of-squares column stays zero)
  {
    static PerDeviceOnce lds_cap; static int optin_err = 0;
    lds_cap.run([&](int) {
      const void* ks[7] = {(const void*)conv_k3_kernel<4, true, false>, (const void*)conv_k3_kernel<4, false, true>, (const void*)conv_k3_kernel<4, false, false>,
                           (const void*)conv_k3_kernel<2, true, false>, (const void*)conv_k3_kernel<2, false, true>, (const void*)conv_k3_kernel<2, false, false>,
                           (const void*)conv_k3_kernel<4, false, true, false, true>};
      for (const void* kp : ks) { hipError_t e_ = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e_ != hipSuccess) optin_err = (int)e_; }
      (void)hipGetLastError();
    });
    if (optin_err) return AM_STAGE_ERR(optin_err);
  }
  AM_LAUNCH(kern, dim3(G), dim3(512), KLDS, (hipStream_t)stream, a);
  AM_CHECK_LAUNCH_STAGE();
  return 1;
}

// ConvTranspose3d k4 s2 p1 (bf16, dense): 1 = served by the persistent kernel's transposed instantiation, 0 = does not qualify
int conv_k3t_launch(int mode, int dtype, int ksize, int stride, ConvArgs& c, void* stream) {
  if (mode != AM_CONVT_FWD || dtype != AM_DT_BF16 || ksize != 4 || stride != 2 || c.in_mask.m || c.out_mask.m) return 0;
  if (c.accumulate || c.nb_x || c.partials || c.ep_scale || c.ep_res || c.ep_act != AM_ACT_NONE) return 0;
  if (c.Cin % 32 || c.Cin < 64 || c.Cout % 64 || c.Di % KBD || c.Hi % KBH || c.Wi % KBW) return 0;      // (>= 2 slabs: the deferred stores of a class ride in the next one's first item)
  if (c.Do != 2 * c.Di || c.Ho != 2 * c.Hi || c.Wo != 2 * c.Wi) return 0;
  if (c.B > 255 || c.Di / KBD > 255 || c.Hi / KBH > 255 || c.Wi / KBW > 255) return 0;
  if ((size_t)(KED + 1) * c.Hi * c.Wi * c.Cin * 2 >= 0x7fffff00ull || (size_t)16 * c.Wo * c.Cout * 2 >= 0x7fffff00ull) return 0;
  int units;
  const int G = k3_grid(c.B, c.Di, c.Hi, c.Wi, c.Cout, &units);          // units = bricks x channel tiles (the eight classes live inside a unit)
  if (units < K3_MIN_UNITS) return 0;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_CV_NOK3T"); if (e_ && atoi(e_)) return 0; }
#endif
  K3Args a;
  a.x = (const bf16_t*)c.x; a.w = (const bf16_t*)c.w; a.bias = c.bias; a.y = (bf16_t*)c.y; a.partials = nullptr;
  a.ep_scale = nullptr; a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_act = AM_ACT_NONE;
  a.B = c.B; a.D = c.Di; a.H = c.Hi; a.W = c.Wi; a.Cin = c.Cin; a.Cout = c.Cout; a.Cinp = c.Cinp; a.Coutp = c.Coutp;
  a.nbd = c.Di / KBD; a.nbh = c.Hi / KBH; a.nbw = c.Wi / KBW; a.ny = c.Cout / 64; a.nunit = units; a.nslab = c.Cinp / 32;
  a.flip = 0;
  a.nt_store = (size_t)c.B * c.Do * c.Ho * c.Wo * c.Cout * 2 >= ((size_t)384 << 20);
  a.w_bytes = (unsigned)c.w_bytes;
#ifdef AM_ABLATE
  { const char* e = getenv("AM_K3_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  auto kern = conv_k3_kernel<4, false, false, true>;
  AM_LDS_OPTIN_STAGE(kern);
  AM_LAUNCH(kern, dim3(G), dim3(512), KLDS, (hipStream_t)stream, a);
  AM_CHECK_LAUNCH_STAGE();
  return 1;
}

}  // namespace amconv
