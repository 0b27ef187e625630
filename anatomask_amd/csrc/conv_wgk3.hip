// Weight gradient of the dense 3x3x3 stride-1 convolutions (the decoder's conv3x3x3 pairs and densify projections, P/decoder3D.py:20-22,
// P/AnatoMask.py:63-65; autograd's dW of loss.backward(), P/pretrain_AntoMask.py:435) for gfx950, bf16, channels-last, 64 x 64 channel tiles:
//     dW[td][th][tw][cy][cx] = sum_q dY[q][cy] * X[q + (td, th, tw) - 1][cx]
// a GEMM with M = cy, N = cx, K = voxels.  Round 5: the structure that carried conv_k3.hip, applied to the contraction over voxels.
// conv_wgrad.hip runs two independent 4-wave workgroups per CU with register staging (global -> VGPR -> ds_write -> barrier -> 4 k-steps);
// its matrix pipe is busy half of the cycles with clock to spare.  Here:
//   * ONE 8-wave workgroup per CU owns a (64 cy x 64 cx) tile of ONE d-tap (9 taps (th, tw)) and walks one-plane 1 x 8 x 16 bricks of dY,
//     d fastest, the columns of an XCD's slots interleaved (the walk of conv_wgrad.hip: x and dY come from HBM ~1.26 times); the three
//     d-taps and the channel tiles of a brick-walk slot are sibling workgroups on one XCD that walk in step (one L2);
//   * both operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KB = 8 rows of 128 bytes per instruction, 39 pieces per
//     brick: 16 of dY, 23 of the haloed 10 x 18 X plane) into a ring of THREE buffers; bricks n + 1 and n + 2 are in flight while brick n is contracted.  Rows are
//     unpadded; the 32-byte channel groups of a row are XOR-swizzled with ((w >> 1) & 3) ON THE SOURCE SIDE (lane l fetches the chunk that
//     belongs at its linear LDS position), which spreads the 8 voxel rows of a half-wave's transposing read over all 64 banks for every
//     tap shift.  Halo rows outside the volume are out-of-range offsets (zeros), decided per piece by a 5-bit lane code AND a scalar mask;
//   * fragments by `ds_read_b64_tr_b16` (the contraction index is the voxel, not the contiguous channel); per k-step of 32 voxels (two
//     h-rows) a wave reads 8 dY fragments (64 cy) and 12 X fragments (16 cx: 4 rows x 3 w-shifts shared by the three h-taps) for 36 MFMAs;
//   * waves 0-3 ("X") contract h-rows 0-3 of every brick, waves 4-7 ("Y") h-rows 4-7, in ANTIPHASE with one raw barrier per k-step:
//     between two barriers X runs [20 reads + DMA issue, 36 MFMAs] and Y [36 MFMAs, 20 reads + DMA issue] -- a SIMD's matrix pipe always
//     has one of its two waves feeding it.  144 accumulator registers per wave; the two halves' partial sums meet in the fp32 atomics of
//     the flush (as the brick-walk slots' do).
// CYT = 2: 32-wide cy tiles (Cy % 64 != 0: the 64 -> 32 conv in front of the projection, STUNet-H's 192 -> 96): dY rows of 64 bytes (8 pieces
// per brick, 31 in all: four per wave), 2 dY fragments and 18 MFMAs per k-step -- the X brick is fetched for half the MFMAs, so this form
// needs 27 B/clk/CU of DMA where the 64-wide one needs 17.
// Serves: bf16, dense operands, H % 8 == W % 16 == 0, Cx % 64 == Cy % 32 == 0, atomic (non-deterministic) accumulation; everything else,
// and the deterministic mode, stays on conv_wgrad.hip (am_conv3d_wgrad decides).
#include <stdlib.h>
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

constexpr int WMV = 128;                               // brick: 1 x 8 x 16 voxels of dY
constexpr int WEW = 18;                                // haloed X brick: 10 x 18 voxels of ONE plane
constexpr int WXROWS = 184;                            // 180 rows rounded up to whole 8-row pieces
constexpr int WDXB = WXROWS * 128;                     // 23 552 bytes of X per buffer
// per cy-tile count CYT (16 channels each): dY bytes, DMA pieces of dY (1 KB each) and in all, pieces per wave, buffer and ring bytes
template <int CYT> struct WkGeo {
  static constexpr int DYB = WMV * 32 * CYT;            // 16 384 (CYT = 4) / 8 192
  static constexpr int NPY = 4 * CYT, NP = NPY + 23;    // 39 / 31 pieces per brick
  static constexpr int NPW = (NP + 7) / 8;              // 5 / 4 per wave (the last wave re-issues the last piece)
  static constexpr int BUF = DYB + WDXB;                // 39 936 / 31 744
  static constexpr int LDS = 3 * BUF;                   // 119 808 / 95 232: a ring of three bricks (one contracted, two in flight)
};

struct Wk3Args {
  const bf16_t* x; const bf16_t* dy; float* dw;
  int B, D, H, W, Cx, Cy;
  int nbh, nbw;                                        // bricks per plane
  int split, ntile, ncxt;                              // brick-walk slots (a multiple of 8), (cy, cx) tiles, cx tiles
  int seg_len, nseg;                                   // d-segments of a column
#ifdef AM_ABLATE
  int dbg;                                             // tools build: 1 no flush, 2 no MFMAs, 4 DMA of zeros, 16 no fragment reads, 32 no DMA instructions, 64 no barriers
#endif
};

__device__ __forceinline__ int w_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P> __device__ __forceinline__ P* w_uni_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}
#define WK_LDSP(off) ((__attribute__((address_space(3))) void*)(lds + (off)))

// The transposing reads are issued as inline assembly: behind an LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every
// `ds_read_b64_tr_b16` it emits itself (the intrinsic carries no memory operand, so its scoreboard assumes the read may alias the DMA's
// LDS destination) -- the next brick's DMA would be waited for at the top of every k-step instead of flying under the MFMAs.  The price:
// the compiler no longer counts these reads either, so the `lgkmcnt` waits in front of the MFMAs are explicit (wk_wait ties them to the
// fragment registers so that no MFMA can be scheduled above its wait).
#define WK_TRR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
// fragments of k-step KS of the brick whose buffer the lane addresses point into: dY h-rows 2 KS, 2 KS + 1 (8 reads), then the X rows
// 2 KS .. 2 KS + 3 under the three w-shifts, rows 0 and 1 first (the h-tap 0 MFMAs start on them)
template <int KS, int CYT>
__device__ __forceinline__ void wk_read(const unsigned (&ya)[CYT], const unsigned (&xb)[3], s16x4 (&alo)[CYT], s16x4 (&ahi)[CYT], s16x4 (&xr)[3][4]) {
  constexpr int HR = 512 * CYT;                          // bytes of one h-row of dY (16 voxels x 32 CYT bytes)
  WK_TRR(alo[0], ya[0], KS * 2 * HR); WK_TRR(ahi[0], ya[0], KS * 2 * HR + HR);
  WK_TRR(alo[1], ya[1], KS * 2 * HR); WK_TRR(ahi[1], ya[1], KS * 2 * HR + HR);
  if constexpr (CYT == 4) {
    WK_TRR(alo[2], ya[2], KS * 2 * HR); WK_TRR(ahi[2], ya[2], KS * 2 * HR + HR);
    WK_TRR(alo[3], ya[3], KS * 2 * HR); WK_TRR(ahi[3], ya[3], KS * 2 * HR + HR);
  }
  WK_TRR(xr[0][0], xb[0], (KS * 2 + 0) * 2304); WK_TRR(xr[1][0], xb[1], (KS * 2 + 0) * 2304); WK_TRR(xr[2][0], xb[2], (KS * 2 + 0) * 2304);
  WK_TRR(xr[0][1], xb[0], (KS * 2 + 1) * 2304); WK_TRR(xr[1][1], xb[1], (KS * 2 + 1) * 2304); WK_TRR(xr[2][1], xb[2], (KS * 2 + 1) * 2304);
  WK_TRR(xr[0][2], xb[0], (KS * 2 + 2) * 2304); WK_TRR(xr[1][2], xb[1], (KS * 2 + 2) * 2304); WK_TRR(xr[2][2], xb[2], (KS * 2 + 2) * 2304);
  WK_TRR(xr[0][3], xb[0], (KS * 2 + 3) * 2304); WK_TRR(xr[1][3], xb[1], (KS * 2 + 3) * 2304); WK_TRR(xr[2][3], xb[2], (KS * 2 + 3) * 2304);
}
static_assert(2304 == 18 * 128, "X brick rows");

template <int CYT>
__global__ __launch_bounds__(512, 2) void wgrad_k3_kernel(Wk3Args a) {
  typedef WkGeo<CYT> G_;
  constexpr int WDYB = G_::DYB, WNPY = G_::NPY, WNP = G_::NP, NPW = G_::NPW, WBUF = G_::BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isX = wave < 4;
  const int wx = wave & 3;                              // cx tile (16 channels) of this wave inside the workgroup's 64
  const int g = lane >> 4, r16 = lane & 15, q = (lane >> 2) & 3, p = lane & 3;

  // ---- workgroup id -> (slot, d-tap, channel tile): siblings differ by 8 (one XCD) and are dispatched back to back
  const int nsib = 3 * a.ntile;
  const int blk = blockIdx.x / (8 * nsib), rem = blockIdx.x % (8 * nsib);
  const int sib = rem >> 3, tile = sib / 3, grp = sib - tile * 3;
  const int slot = blk * 8 + (rem & 7);
  const int cy0 = (tile / a.ncxt) * (16 * CYT), cx0 = (tile % a.ncxt) * 64;
  const int S8 = a.split >> 3, xcd = slot & 7;
  const int ncol = a.B * a.nbh * a.nbw;
  const int c0 = (int)((long)ncol * xcd / 8), ncx = (int)((long)ncol * (xcd + 1) / 8) - c0;
  const int nu = ncx * a.nseg;

  // ---- the walk as ONE iterator that runs two bricks ahead of the contraction (it feeds the DMA; the contraction only needs to know
  // how many bricks there are).  Everything a brick's DMA needs is scalar state advanced by additions: the 32-bit byte offsets of the
  // brick inside its sample (soffset of the buffer loads; the descriptors are per SAMPLE and change only with b), the face mask of its
  // (h, w) position, the plane of this d-tap.  Divisions happen once per unit (a d-segment of a column), not per brick.
  const int plane_y = a.H * a.W * a.Cy * 2, plane_x = a.H * a.W * a.Cx * 2;       // bytes per d-plane (< 2^31: conv_wgk3_qualifies)
  const long long samp_y = (long long)a.D * a.H * a.W * a.Cy, samp_x = (long long)a.D * a.H * a.W * a.Cx;   // elements per sample
  int ntot = 0;
  for (int u = slot >> 3; u < nu; u += S8) { const int left = a.D - (u / ncx) * a.seg_len; ntot += left < a.seg_len ? left : a.seg_len; }
  ntot = w_uni(ntot);
  if (ntot == 0) return;
  int it_u = slot >> 3, it_left = 0, it_d = 0, it_b = -1;   // unit, bricks left in it after the current one, d of the current brick, sample
  unsigned it_ys = 0, it_xs = 0, it_m = 16u;               // soffsets of the current brick; faces of the haloed brick outside the volume
  bool it_ok = true;                                      // false: past the last brick (its DMA writes zeros into a ring slot nobody reads)
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0, 0x00020000), rx = ry;
  auto enter_unit = [&]() {                               // it_u -> first brick of that unit
    const int seg = it_u / ncx, col = c0 + it_u - seg * ncx;
    const int bw_ = col % a.nbw, t_ = col / a.nbw, bh_ = t_ % a.nbh, b_ = w_uni(t_ / a.nbh);
    const int h0 = w_uni(bh_ * 8), w0 = w_uni(bw_ * 16);
    it_d = w_uni(seg * a.seg_len);
    const int left = a.D - it_d;
    it_left = w_uni((left < a.seg_len ? left : a.seg_len) - 1);
    it_m = (h0 == 0 ? 1u : 0u) | (h0 + 8 == a.H ? 2u : 0u) | (w0 == 0 ? 4u : 0u) | (w0 + 16 == a.W ? 8u : 0u) | 16u;
    it_ys = (unsigned)(((it_d * a.H + h0) * a.W + w0) * a.Cy * 2);
    // X: the descriptor base sits one plane + one row + one voxel BEFORE the sample, so the offset of the haloed brick's first voxel
    // (d + td - 1, h0 - 1, w0 - 1) is never negative: ((d + td) H + h0) W + w0 voxels
    it_xs = (unsigned)((((it_d + grp) * a.H + h0) * a.W + w0) * a.Cx * 2);
    if (b_ != it_b) {
      it_b = b_;
      ry = __builtin_amdgcn_make_buffer_rsrc((void*)w_uni_ptr(a.dy + (long long)b_ * samp_y + cy0), 0, 0x7fffff00, 0x00020000);
      rx = __builtin_amdgcn_make_buffer_rsrc((void*)w_uni_ptr(a.x + (long long)b_ * samp_x + cx0 - (long long)(a.H * a.W + a.W + 1) * a.Cx), 0, 0x7fffff00, 0x00020000);
    }
  };
  auto advance = [&]() {
    if (it_left > 0) { --it_left; ++it_d; it_ys += (unsigned)plane_y; it_xs += (unsigned)plane_x; return; }
    it_u += S8;
    if (it_u < nu) enter_unit(); else it_ok = false;
  };
  enter_unit();

  // ---- lane constants
  // DMA: piece pi = wave + 8 k (k = 0..4; pi < 16: dY rows 8 pi .. 8 pi + 7, else X rows 8 (pi - 16) ..); lane -> (row l >> 3, LDS chunk
  // position l & 7).  Wave 7 has no piece 39: its k = 4 re-issues piece 38 (the same bytes to the same place), so that EVERY wave has exactly
  // five DMA operations per brick in flight and one counted wait serves all of them.
  constexpr unsigned OOB = 0x80000000u;
  unsigned pl[NPW], pcode = 0u;
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int pi = (wave + 8 * k) < WNP ? wave + 8 * k : WNP - 1;
    if (k < WNPY / 8) {
      if constexpr (CYT == 4) {
        const int v = 8 * pi + (lane >> 3), h = v >> 4, w = v & 15;
        pl[k] = (unsigned)(((h * a.W + w) * a.Cy + (((lane & 7) ^ (((w >> 1) & 3) << 1)) * 8)) * 2);
      } else {                                            // 64-byte rows: 16 voxels per piece, four 16-byte chunks each; key = (w >> 2) & 1
        const int v = 16 * pi + (lane >> 2), h = v >> 4, w = v & 15;
        pl[k] = (unsigned)(((h * a.W + w) * a.Cy + (((lane & 3) ^ (((w >> 2) & 1) << 1)) * 8)) * 2);
      }
    } else {
      const int rho = 8 * (pi - WNPY) + (lane >> 3), yy = rho / WEW, xx = rho - yy * WEW;
      pl[k] = (unsigned)(((yy * a.W + xx) * a.Cx + (((lane & 7) ^ (((xx >> 1) & 3) << 1)) * 8)) * 2);
      const unsigned code = (yy == 0 ? 1u : 0u) | (yy == 9 ? 2u : 0u) | (xx == 0 ? 4u : 0u) | (xx == WEW - 1 ? 8u : 0u) | (rho >= 180 ? 16u : 0u);
      pcode |= code << (5 * k);
    }
  }
  // fragment reads: the lane's voxel of a k-step is (h = 2 ks (+1 for the second half), w = wv); transposing read of 4 voxels x 16 channels per 16 lanes
  const int wv = 4 * g + q, keyA = CYT == 4 ? (wv >> 1) & 3 : (wv >> 2) & 1;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;     // LDS byte address of the carve (0: no static LDS)
  unsigned ya0[CYT];                                    // dY fragment i (16 cy): 32-byte group i ^ keyA of the voxel's row (ring slot 0)
#pragma unroll
  for (int i = 0; i < CYT; ++i) ya0[i] = lds0 + wv * (32 * CYT) + 8 * p + ((i ^ keyA) << 5);
  unsigned xb0[3];                                      // X fragment of w-shift tw: row (h, wv + tw), 32-byte group wx ^ key(wv + tw)
#pragma unroll
  for (int tw = 0; tw < 3; ++tw) xb0[tw] = lds0 + WDYB + (wv + tw) * 128 + 8 * p + ((wx ^ (((wv + tw) >> 1) & 3)) << 5);

  // pieces [k0, k1) of this wave for the iterator's brick into ring slot `buf`: per piece a mask test, a select and the load
  auto issue_pieces = [&](const int buf, const int k0, const int k1) __attribute__((always_inline)) {
    const int dx = it_d + grp - 1;                        // X plane of this d-tap
    const bool xok = it_ok && dx >= 0 && dx < a.D;
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int pi = (wave + 8 * k) < WNP ? wave + 8 * k : WNP - 1;
      if (k < WNPY / 8) {
        unsigned vo = it_ok ? pl[k] : OOB;
#ifdef AM_ABLATE
        if (a.dbg & 4) vo = OOB;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, WK_LDSP(buf * WBUF + pi * 1024), 16, vo, it_ys, 0, 0);
      } else {
        const unsigned bad = (pcode >> (5 * k)) & it_m;
        unsigned vo = (bad || !xok) ? OOB : pl[k];
#ifdef AM_ABLATE
        if (a.dbg & 4) vo = OOB;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, WK_LDSP(buf * WBUF + WDYB + (pi - WNPY) * 1024), 16, vo, it_xs, 0, 0);
      }
    }
  };

  f32x4 acc[9][CYT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CYT; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;

  // X waves contract k-steps 0, 1 (h-rows 0-3) of every brick, Y waves k-steps 2, 3: the Y offset is part of the lane constants
  if (!isX) {
#pragma unroll
    for (int i = 0; i < CYT; ++i) ya0[i] += 2 * (1024 * CYT);
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) xb0[tw] += 4 * (WEW * 128);
  }

  // ---- prologue: bricks 0 and 1 are issued (a brick past the end: zeros); brick 0 has landed when all but this wave's five pieces of
  // brick 1 have (vmcnt retires in order)
  issue_pieces(0, 0, NPW); advance();
  issue_pieces(1, 0, NPW); advance();
  if constexpr (NPW == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int cb = 0;                                            // ring slot of the brick being contracted
  for (int n = 0; n < ntot; ++n) {
    const int fb = cb >= 1 ? cb - 1 : 2;                 // ring slot (cb + 2) % 3: the one brick n - 1 has left
    unsigned ya[CYT], xb[3];
#pragma unroll
    for (int i = 0; i < CYT; ++i) ya[i] = ya0[i] + (unsigned)(cb * WBUF);
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) xb[tw] = xb0[tw] + (unsigned)(cb * WBUF);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // ---------------- L: the fragments of this wave's k-step i from LDS, then its share of the DMA of brick n + 2
      s16x4 alo[CYT], ahi[CYT], xr[3][4];
#ifdef AM_ABLATE
      if (a.dbg & 16) {
#pragma unroll
        for (int j = 0; j < CYT; ++j) { alo[j] = s16x4{1, 1, 1, 1}; ahi[j] = s16x4{1, 1, 1, 1}; }
#pragma unroll
        for (int tw = 0; tw < 3; ++tw)
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[tw][r] = s16x4{1, 1, 1, 1};
      } else
#endif
      if (i == 0) wk_read<0, CYT>(ya, xb, alo, ahi, xr); else wk_read<1, CYT>(ya, xb, alo, ahi, xr);
#ifdef AM_ABLATE
      if (!(a.dbg & 32))
#endif
      issue_pieces(fb, i == 0 ? 0 : NPW - 2, i == 0 ? NPW - 2 : NPW);
      // ONE barrier per k-step and wave, X behind its MFMAs, Y in front of them: between two barriers X runs [L(n) M(n)] and Y
      // [M(n - 1) L(n)] -- X fetches while Y multiplies, then the other way round; the program order (L, then M) is the same for both.
      // What the barriers order: brick n + 1 (DMA'd during brick n - 1's two intervals) has landed -- in front of the barrier that ends
      // brick n (X behind M(1), Y behind L(1)) every wave waits for all but its five pieces of brick n + 2, issued during brick n -- before
      // anyone reads it; all reads of brick n - 1 were complete (X's are consumed by its MFMAs, Y drains `lgkmcnt`) before the barrier
      // behind which the DMA of brick n + 2 starts to overwrite that ring slot.  The DMA runs two bricks ahead: with one brick of
      // lookahead every brick ended on the full latency of its last pieces (the launch took the SUM of its MFMA and DMA times).
      if (!isX) {
        if (i == 1) { if constexpr (NPW == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef AM_ABLATE
        if (!(a.dbg & 64))
#endif
        __builtin_amdgcn_s_barrier();
      }
      // ---------------- M: 36 MFMAs
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      // dY fragments + X rows 0, 1 have landed (reads return in order: 6 may still be out)
      if constexpr (CYT == 4)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]), "+v"(ahi[3]),
                     "+v"(xr[0][0]), "+v"(xr[1][0]), "+v"(xr[2][0]), "+v"(xr[0][1]), "+v"(xr[1][1]), "+v"(xr[2][1]) :: "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(ahi[0]), "+v"(ahi[1]),
                     "+v"(xr[0][0]), "+v"(xr[1][0]), "+v"(xr[2][0]), "+v"(xr[0][1]), "+v"(xr[1][1]), "+v"(xr[2][1]) :: "memory");
      bfx8 af[CYT];
#pragma unroll
      for (int j = 0; j < CYT; ++j)
        af[j] = __builtin_bit_cast(bfx8, s16x8{alo[j][0], alo[j][1], alo[j][2], alo[j][3], ahi[j][0], ahi[j][1], ahi[j][2], ahi[j][3]});
#ifdef AM_ABLATE
      if (!(a.dbg & 2))
#endif
#pragma unroll
      for (int t = 0; t < 9; ++t) {                      // tap (th, tw) = (t / 3, t % 3): X rows h + th, h + th + 1
        // (tied to the accumulators of the previous h-tap as well: the wait stays BEHIND those MFMAs instead of being hoisted to the top)
        if constexpr (CYT == 4) {
        if (t == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xr[0][2]), "+v"(xr[1][2]), "+v"(xr[2][2]), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                                 "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]) :: "memory");
        if (t == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0][3]), "+v"(xr[1][3]), "+v"(xr[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]),
                                 "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3]), "+v"(acc[5][0]), "+v"(acc[5][1]), "+v"(acc[5][2]), "+v"(acc[5][3]) :: "memory");
        } else {
        if (t == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xr[0][2]), "+v"(xr[1][2]), "+v"(xr[2][2]), "+v"(acc[0][0]), "+v"(acc[0][1]),
                                 "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]) :: "memory");
        if (t == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0][3]), "+v"(xr[1][3]), "+v"(xr[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]),
                                 "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[5][0]), "+v"(acc[5][1]) :: "memory");
        }
        const s16x4 lo = xr[t % 3][t / 3], hi = xr[t % 3][t / 3 + 1];
        const bfx8 bf = __builtin_bit_cast(bfx8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
        for (int j = 0; j < CYT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bf, acc[t][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (isX) {
        if (i == 1) { if constexpr (NPW == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
#ifdef AM_ABLATE
        if (!(a.dbg & 64))
#endif
        __builtin_amdgcn_s_barrier();
      }
    }
    cb = cb == 2 ? 0 : cb + 1;
    advance();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the tail's zero-fill DMA must not outlive the workgroup's LDS)

  // ---- flush: D row = cy 4 g + r, col = cx r16; tap index (td = grp, th, tw) = 9 grp + t
#ifdef AM_ABLATE
  if (a.dbg & 1) return;
#endif
  const int cx = cx0 + 16 * wx + r16;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* dst = a.dw + ((size_t)(9 * grp + t) * a.Cy + cy0) * a.Cx + cx;
#pragma unroll
    for (int i = 0; i < CYT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dst + (size_t)(16 * i + 4 * g + r) * a.Cx, acc[t][i][r]);
  }
}

}  // namespace

namespace amconv {

// shapes this kernel serves (bf16, dense, k3 s1 are the caller's part of the condition)
bool conv_wgk3_qualifies(int B, int D, int H, int W, int Cx, int Cy) {
  if (Cx % 64 || Cy % 32 || H % 8 || W % 16 || B < 1 || D < 1) return false;
  if ((size_t)(D + 2) * H * W * (Cx > Cy ? Cx : Cy) * 2 >= 0x7fffff00ull) return false;   // 32-bit byte offsets inside a sample (+ the halo plane)
  const int ncol = B * (H / 8) * (W / 16);
  return ncol >= 8 && (long)ncol * D >= 8 * 64;            // at least one column per XCD and 64 bricks per slot (a slot flushes 2 x 147 KB of atomics)
}

// 1: the launch was served; 0: the shape does not qualify (the caller goes on to conv_wgrad.hip); < 0: failure (AM_STAGE_ERR)
int conv_wgk3_launch(const void* x, const void* dy, float* dw, int B, int D, int H, int W, int Cx, int Cy, void* stream) {
  if (!conv_wgk3_qualifies(B, D, H, W, Cx, Cy)) return 0;
  const int nbh = H / 8, nbw = W / 16, ncol = B * nbh * nbw;
  const long nbrick = (long)ncol * D;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_NOK3"); if (e_ && atoi(e_)) return 0; }
#endif
  static PerDeviceOnce cu_once; static int cus_of[64];     // per device (ordinal & 63, as PerDeviceOnce keys it): the slot count follows the device that launches
  int dev_ = 0; (void)hipGetDevice(&dev_);
  cu_once.run([&](int d) { hipDeviceProp_t pr; cus_of[d & 63] = (hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; (void)hipGetLastError(); });
  const int cus_cached = cus_of[dev_ & 63] > 0 ? cus_of[dev_ & 63] : 256;
  const int cap = cus_cached / 8 > 0 ? cus_cached / 8 : 1;   // resident workgroups per XCD: one per CU
  Wk3Args a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cx = Cx; a.Cy = Cy; a.nbh = nbh; a.nbw = nbw;
  const bool wide = Cy % 64 == 0;                          // 64-wide cy tiles; else 32-wide ones (Cy = 32, 96)
  a.ncxt = Cx / 64; a.ntile = (wide ? Cy / 64 : Cy / 32) * a.ncxt;
  const int nsib = 3 * a.ntile;
  // slots per XCD: whole rounds of the resident set with the fewest idle CUs (siblings come in groups of nsib), fewer rounds preferred
  int s8 = 0; double best = 0.0;
  for (int R = 1; R <= 4; ++R) {
    const int s = cap * R / nsib;
    const double eff = s > 0 ? (double)s * nsib / (cap * R) : 0.0;
    if (s >= 1 && eff > best + 0.02) { best = eff; s8 = s; }
  }
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WGK3_S8"); if (e_ && atoi(e_) > 0) s8 = atoi(e_); }
#endif
  if (s8 < 1) s8 = 1;                                      // the tiles alone over-fill the chip: one slot per XCD
  {
    long capb = nbrick / 64 / 8;                           // >= 64 bricks per slot
    if (capb < 1) capb = 1;
    if (s8 > capb) s8 = (int)capb;
    const int colx = ncol / 8;                             // a slot needs work units: columns x segments per XCD
    (void)colx;
  }
  a.split = s8 * 8;
  {
    const int ncx = (ncol + 7) / 8;
    int nseg = (48 * s8 + ncx - 1) / ncx;                  // >= 48 work units per slot, segments of >= 8 planes (each start re-reads a d-halo)
    if (nseg > D / 8) nseg = D / 8;
    if (nseg < 1) nseg = 1;
    a.seg_len = (D + nseg - 1) / nseg;
    a.nseg = (D + a.seg_len - 1) / a.seg_len;
    // every slot of an XCD needs at least one unit
    while ((long)(ncol / 8) * a.nseg < s8 && s8 > 1) --s8;
    a.split = s8 * 8;
  }
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WGK3_DBG"); a.dbg = e_ ? atoi(e_) : 0; }
#endif
  AM_LDS_OPTIN_STAGE(wgrad_k3_kernel<4>);
  AM_LDS_OPTIN_STAGE(wgrad_k3_kernel<2>);
  if (wide) AM_LAUNCH(wgrad_k3_kernel<4>, dim3((unsigned)(a.split * nsib)), dim3(512), WkGeo<4>::LDS, (hipStream_t)stream, a);
  else AM_LAUNCH(wgrad_k3_kernel<2>, dim3((unsigned)(a.split * nsib)), dim3(512), WkGeo<2>::LDS, (hipStream_t)stream, a);
  AM_CHECK_LAUNCH_STAGE();
  return 1;
}

}  // namespace amconv
