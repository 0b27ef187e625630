// Weight gradient of the gather convolution (conv k1/k3 stride 1/2 and ConvTranspose3d k4 s2 p1)
// for gfx950 matrix cores, channels-last.  Autograd counterpart of conv_igemm.hip.
//
//   dW[widx_t][cy][cx] = sum_q  dY[q*OS + p_g][cy] * X[q*IS + shift_t][cx]        (t in tap-group g)
//
// i.e. a GEMM with M = cy (dY channels), N = cx (X channels), K = voxels.  Taps are processed in
// groups that share the dY operand: conv k3 -> 3 groups of 9 taps (same d-shift: the X brick needs
// no d-halo), ConvT -> the 8 output-parity classes of 8 taps.  A workgroup owns (group, 64 cy, 64 cx),
// walks a strided set of q-bricks, stages dY-brick and haloed X-brick channels-last in LDS (branch-free
// loads from a per-thread plan computed once) and contracts over voxels.  The contraction index (voxel)
// is NOT the contiguous one in memory, so bf16 fragments are fetched with ds_read_b64_tr_b16 (hardware
// transpose read: 4 voxels x 16 channels per 16-lane group); f32 fragments are plain 4-byte LDS reads for
// v_mfma_f32_16x16x4_f32.  The tap count is a template parameter: per k-step every fragment read of all
// taps is issued before the first MFMA, so LDS latency is paid once per k-step, not once per tap.
// Partial sums leave the workgroup as f32 atomics into the packed [tap][cy][cx] gradient
// (64-byte runs per 16 lanes; order-dependent in the last bits, like any split-K atomic reduce), or -- deterministic mode, a
// caller-provided workspace -- as plain stores of one partial sum per brick-walk slot that a second kernel folds in slot order.
// Brick shapes and walks: dense bf16 k3 s1 takes one-plane 1x8x16 bricks walked d-fastest, the columns of an XCD's slots interleaved
// (HBM sees x and dY ~1.26 times instead of 2.15); bf16 k3 s2 on grids >= 16 wide stages X at full resolution, one unit per d-tap
// (kernel variant S2); everything else walks contiguous w-fastest runs of 2x4x16 / 2x8x8 bricks.  Block-sparse launches keep the
// current sample's patch masks in LDS.
#include <mutex>
#include <stdlib.h>
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace amconv {
// conv_wgk3.hip: dense bf16 k3 s1 weight gradients with 64 x 64 channel tiles on the 8-wave LDS-DMA kernel; 1 = served, 0 = does not qualify
int conv_wgk3_launch(const void* x, const void* dy, float* dw, int B, int D, int H, int W, int Cx, int Cy, void* stream);
bool conv_wgk3_qualifies(int B, int D, int H, int W, int Cx, int Cy);
}

namespace {

struct WgArgs {
  const void* x; const void* dy; float* dw;
  int B, Dx, Hx, Wx, Cx, Dy, Hy, Wy, Cy;
  int OS, GS, ngroup;         // dY stride (ConvT parity classes), global X stride (2: strided conv, X read per parity sub-lattice)
  int QS;                     // X voxels per q step (= GS, except the full-resolution strided variant S2: GS = 1, QS = 2)
  int zmap[8];                // blockIdx.z -> unit (one launch handles the units that share a tap count)
  int upar[8];                // X parity of the unit when GS == 2
  int nbd, nbh, nbw;
  int tap_begin[9];
  int taps[64];
  int mind[8], minh[8], minw[8];
  int ed[8], eh[8], ew[8];
  int mdiv_w[8], mdiv_hw[8];  // 2^20-scaled reciprocals of ew and ew*eh
  int pofs[8];                // parity of dY voxels per group: pd<<2|ph<<1|pw
  MaskView x_mask, y_mask;
  int split, ntile;           // brick-walk slots per (tile, group) ; (cy, cx) channel tiles
  int y_uni;                  // block-sparse dY whose bricks each lie inside ONE patch, grid = whole bricks: one mask lookup per brick
  int mask_off, mask_n;       // block-sparse operands: LDS byte offset of the two cached patch-mask arrays (dY's, X's) of ONE sample, bytes each (0: not cached)
  const int* plist; int nlive, pbd, pbh, pbw;   // y_uni launches with an active-patch list: the walk runs over the LIVE bricks only (patch list entry
                              // b << 24 | pd << 16 | ph << 8 | pw, pbd x pbh x pbw bricks per patch), so every slot gets the same number of them
  int walk, seg_len, nseg;    // walk 1: d-fastest segments of seg_len bricks, columns interleaved over the slots of an XCD (see the kernel)
  float* det_ws;              // deterministic mode: [split][k^3][Cy][Cx] per-slot partial sums (plain stores), folded in slot order
  long det_stride;            //   floats per slot
#ifdef AM_ABLATE
  int dbg;                    // tools-only build (-DAM_ABLATE): AM_WG_DBG ablation bits, 1 no flush, 2 no contraction, 4 no global loads, 8 no LDS staging writes
#endif
};
#ifdef AM_ABLATE
#define AM_DBG(a_, bit_) (((a_).dbg & (bit_)) != 0)
#else
#define AM_DBG(a_, bit_) false
#endif


#ifdef AM_ABLATE
inline bool mi_is4(int Cx, int Cy) { return Cx > 32 && Cy > 32; }
inline bool wg_dma() { const char* e_ = getenv("AM_WG_DMA"); return e_ && atoi(e_); }     // tools-only build: the LDS-DMA plane-brick variant (same-process A/B)
#endif

inline int wg_slots(int occ) {                    // resident workgroups of the CURRENT device (CU count cached per device)
  static int ncu[64];
  static PerDeviceOnce once;
  int dev = 0;
  (void)hipGetDevice(&dev);
  once.run([&](int d) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256; ncu[d & 63] = n; (void)hipGetLastError(); });
  return occ * ncu[dev & 63];
}

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// MI = 16-row cy tiles per wave (CT = 16*MI dY channels per workgroup); NWX = waves along cx (KT = 16*NWX X channels); the
// remaining 4/NWX waves split the k-steps (voxels) of a brick.  64x64 tiles for the wide layers; 32-channel operands
// (STUNet-B level 0, decoder output level) get 32-wide tiles instead of multiplying zero padding.
// S2 (stride-2 conv k3, bf16): the X brick is staged at FULL resolution ((2 BH + 1) x (2 BW + 1) voxels of one plane) and the fragment
// reads step two voxel rows per q -- the 9 (kh, kw) taps of a d-tap share one staged brick and one dY brick, instead of 8 parity
// sub-lattice units that each re-stage dY for 1-8 taps.
// PF: the loads of the next live brick are issued BEFORE the contraction of the current one and land in the staging registers while
// the matrix cores run (variants with registers to spare: the small bricks of the block-sparse stride-2 layers, whose 36 MFMAs per
// wave and brick cannot hide a memory round trip behind the other workgroup of the CU).
// DMA (bf16, dense operands, one-plane 1x8x16 bricks, 64 x 64 tiles): the two bricks go global -> LDS by `buffer_load ... lds` (no staging registers,
// no ds_write phase) into one of TWO buffers, so brick n+1 is in flight while brick n is contracted and a brick costs ONE barrier:
//   s_waitcnt vmcnt(0) -> barrier -> issue brick n+1 -> contract brick n.
// An LDS-DMA instruction fills 1 KB = 8 CONSECUTIVE 128-byte rows, so the rows cannot be padded to the conflict-free 160-byte stride of the
// register-staged layout; instead the 32-byte channel groups of a row are XOR-swizzled with ((w >> 1) & 3) of the row's w coordinate -- on the SOURCE side
// (lane l of the instruction fetches the channel chunk that belongs at its LDS position) -- which spreads the 8 voxel rows of a half-wave's
// transposing read over all 64 banks; w + tap shift keeps the key a lane constant per w-tap.
template <typename T, int BD, int BH, int BW, int NTAP, int NITX, int MI = 4, int NWX = 4, bool S2 = false, bool PF = false, bool DMA = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgArgs a) {
  constexpr int EPC = TT<T>::EPC;
  constexpr int CT = 16 * MI, KT = 16 * NWX, KS = 4 / NWX;
  constexpr int MV = BD * BH * BW;
  constexpr int KSTEP = sizeof(T) == 2 ? 32 : 4;
  // LDS row strides (bytes).  bf16: 160 B = 40 dwords: the 8 consecutive voxel rows a half-wave's ds_read_b64_tr_b16
  // touches land on dword offsets {0,40,16,56,32,8,48,24} (mod 64): 8 disjoint 8-dword runs = all 64 banks, conflict-free.
  constexpr int RPAD = sizeof(T) == 2 ? 32 : 16;
  constexpr int RSY = CT * sizeof(T) + RPAD;
  constexpr int RSX = KT * sizeof(T) + RPAD;
  constexpr int CPRY = CT / EPC, CPRX = KT / EPC;        // 16-byte chunks per row
  constexpr int NITY = (MV * CPRY + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ldsY = lds;
  unsigned char* ldsX = lds + MV * RSY;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int wx = wave % NWX, wk = wave / NWX;            // cx tile of this wave ; its share of the k-steps
  // workgroup id -> (brick-walk slot s, tap group, channel tile): the workgroups that walk the SAME bricks -- the tap groups and
  // the (cy, cx) tiles of one slot -- get ids that differ by 8, i.e. the same XCD under round-robin dispatch, and are dispatched
  // back to back, so they walk in step and x / dy come from HBM once per slot instead of once per group and tile (speed only)
  const int ng_ = a.ngroup, nsib_ = ng_ * a.ntile;
  const int blk_ = blockIdx.x / (8 * nsib_), rem_ = blockIdx.x % (8 * nsib_);
  const int sib_ = rem_ >> 3, tile_ = sib_ / ng_;
  // (fewer than 8 slots -- layers whose tiles alone fill the chip: the live slots rotate with the tile index, or every live
  // workgroup would sit on XCD 0)
  const int slot = blk_ * 8 + ((rem_ & 7) + (a.split < 8 ? 8 - (tile_ & 7) : 0)) % 8;
  if (slot >= a.split) return;
  const int grp = a.zmap[sib_ - tile_ * ng_];
  const int ncxt = (a.Cx + KT - 1) / KT;
  const int cy0 = (tile_ / ncxt) * CT, cx0 = (tile_ % ncxt) * KT;
  const int pd = (a.pofs[grp] >> 2) & 1, ph = (a.pofs[grp] >> 1) & 1, pw = a.pofs[grp] & 1;
  const int ED = a.ed[grp], EH = a.eh[grp], EW = a.ew[grp];
  const int nvox = ED * EH * EW;
  const int mW = a.mdiv_w[grp], mHW = a.mdiv_hw[grp], EHW = EH * EW;
  const int tb = a.tap_begin[grp];
  const int upd = (a.upar[grp] >> 2) & 1, uph = (a.upar[grp] >> 1) & 1, upw = a.upar[grp] & 1;

  f32x4 acc[NTAP][MI];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int tapoff[NTAP];                                      // byte offset of each tap's window in the X brick (wave-uniform)
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tp = a.taps[tb + t];
    const int sd = (tp & 15) - 8, sh = ((tp >> 4) & 15) - 8, sw = ((tp >> 8) & 15) - 8;
    tapoff[t] = (((sd - a.mind[grp]) * EH + (sh - a.minh[grp])) * EW + (sw - a.minw[grp])) * RSX;
  }

  // ---- per-thread staging plans (brick-relative, computed once) ----
  // Row `it` of a thread = voxel (tid / CPR + it * 256 / CPR) of the brick, always the same 16-byte channel chunk.  Everything a
  // brick needs per row is ONE add: its global BYTE offset relative to the brick origin is precomputed here (channel chunk
  // included), the LDS destination is an immediate multiple of the row stride, and the brick-relative coordinates (needed only
  // by border / masked bricks) are packed in one register.  The brick loop itself touches no kernel argument per row.
  constexpr int VPI_Y = 256 / CPRY, VPI_X = 256 / CPRX;   // voxel rows per staging iteration
  static_assert(256 % CPRY == 0 && 256 % CPRX == 0, "a thread keeps its channel chunk across iterations");
  const int ychan = (tid % CPRY) * EPC, xchan = (tid % CPRX) * EPC;
  const bool ycok = cy0 + ychan < a.Cy, xcok = cx0 + xchan < a.Cx;
  constexpr unsigned OOB = 0x80000000u;
  unsigned yoffB[NITY], xoffB[NITX];                     // OOB: this thread has no such row (or its channel chunk is padding)
  int yq[NITY], xq[NITX];                                // brick-relative (d, h, w) of the row: d | h << 8 | w << 16
#pragma unroll
  for (int it = 0; it < NITY; ++it) {
    const int v = tid / CPRY + it * VPI_Y;
    const int vd = v / (BW * BH), vh = (v / BW) % BH, vw = v % BW;
    yq[it] = vd | (vh << 8) | (vw << 16);
    const int ych = DMA ? (((tid % CPRY) ^ (((vw >> 1) & 3) << 1)) * EPC) : ychan;      // DMA: the chunk that belongs at this lane's LDS position
    yoffB[it] = (v < MV && (DMA || ycok)) ? (unsigned)(((((vd * a.OS) * a.Hy + vh * a.OS) * a.Wy + vw * a.OS) * a.Cy + cy0 + ych) * (int)sizeof(T)) : OOB;
  }
#pragma unroll
  for (int it = 0; it < NITX; ++it) {
    const int e = tid / CPRX + it * VPI_X;
    const int ez = (e * mHW) >> 20, rem = e - ez * EHW, ey = (rem * mW) >> 20, ex = rem - ey * EW;
    xq[it] = ez | (ey << 8) | (ex << 16);
    const int xch = DMA ? (((tid % CPRX) ^ (((ex >> 1) & 3) << 1)) * EPC) : xchan;
    xoffB[it] = (e < nvox && (DMA || xcok)) ? (unsigned)((((ez * a.Hx + ey) * a.Wx + ex) * a.GS * a.Cx + cx0 + xch) * (int)sizeof(T)) : OOB;
  }
  const int ydst0 = (tid / CPRY) * RSY + (tid % CPRY) * 16, xdst0 = (tid / CPRX) * RSX + (tid % CPRX) * 16;

  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ yg = (const T*)a.dy;
  const int nbrick = a.plist ? a.nlive : a.B * a.nbd * a.nbh * a.nbw;
  const size_t yplane = (size_t)a.Hy * a.Wy * a.Cy, xplane = (size_t)a.Hx * a.Wx * a.Cx;
  const bool masked = a.x_mask.m != nullptr || a.y_mask.m != nullptr;
  const int Dy_ = a.Dy, Hy_ = a.Hy, Wy_ = a.Wy, Dx_ = a.Dx, Hx_ = a.Hx, Wx_ = a.Wx, OS_ = a.OS, GS_ = a.GS;

  // Brick walk.  walk 0: a workgroup owns a CONTIGUOUS run of bricks (w fastest): the brick coordinates advance by increment-and-carry
  // (no divisions in the loop) and consecutive bricks re-read each other's w-halo rows from L2.
  // walk 1 (dense operands): d fastest.  The work unit is a (column (b, bh, bw), segment of seg_len d-bricks); XCD x owns a contiguous
  // range of columns and its S8 slots take the units j, j + S8, ... of (segment-major, column-minor) order, so at any time the slots
  // of an XCD walk S8 ADJACENT columns through the same d range: the d-halo planes of X are re-read one brick later by the same
  // slot's tap groups and the h/w-halo rows by a neighbouring slot of the same L2 -- HBM sees X and dY about once
  // (profiles/r02_pmc_traffic.md), instead of twice with walk 0.
  const int chunk = (nbrick + a.split - 1) / a.split;
  const int brick0 = slot * chunk, brick1 = brick0 + chunk < nbrick ? brick0 + chunk : nbrick;
  int bw_, bh_, bd_, b;
  const int bpp_ = a.pbd * a.pbh * a.pbw, pbhw_ = a.pbh * a.pbw;
  {
    int bid = a.plist ? 0 : brick0;
    bw_ = bid % a.nbw; bid /= a.nbw;
    bh_ = bid % a.nbh; bid /= a.nbh;
    bd_ = bid % a.nbd; b = bid / a.nbd;
    bw_ -= 1;                                            // (pre-decrement: the loop increments first)
  }
  const int nbw_ = a.nbw, nbh_ = a.nbh, nbd_ = a.nbd;
  const int S8 = a.split >> 3, ncol = a.B * nbh_ * nbw_;
  const int c0 = (int)((long)ncol * (slot & 7) / 8), ncx = (int)((long)ncol * ((slot & 7) + 1) / 8) - c0;
  const int nu = ncx * a.nseg;
  // (two nested counted loops: the flat `for (;;)` form of this walk made hipcc spill 120 VGPRs)
  u32x4 ys[NITY], xs[NITX];                              // staging registers of one brick (dY rows, X rows)
  bool have = false;                                     // PF: ys / xs hold a brick that has not been contracted yet
  auto stage_to_lds = [&]() {
    __syncthreads();                                     // previous brick's fragment reads are done
#pragma unroll
    for (int it = 0; it < NITY; ++it)
      if (tid / CPRY + it * VPI_Y < MV && !AM_DBG(a, 8)) *(u32x4*)(ldsY + ydst0 + it * VPI_Y * RSY) = ys[it];
#pragma unroll
    for (int it = 0; it < NITX; ++it)
      if (tid / CPRX + it * VPI_X < nvox && !AM_DBG(a, 8)) *(u32x4*)(ldsX + xdst0 + it * VPI_X * RSX) = xs[it];
    __syncthreads();
  };
  constexpr int XROWS = ((BH + 2) * 18 + 7) / 8 * 8;                     // DMA: X brick rows, rounded up to whole 8-row instructions (180 -> 184, 108 -> 112)
  constexpr int DYB = MV * 128, DXB = XROWS * 128, DBUF = DYB + DXB;      // DMA: bytes of a dY brick / an X brick / one buffer
  int dcur = 0;                                          // DMA: buffer the NEXT issue fills
  auto contract = [&](int cb = 0) {
    // ---- contract over the brick's voxels ----
    if (!AM_DBG(a, 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int ks = wk; ks < (AM_DBG(a, 2) ? 0 : MV / KSTEP); ks += KS) {
      if constexpr (sizeof(T) == 2) {
        const int q = (lane >> 2) & 3, p = lane & 3;
        // contraction index k = 8g + j of the MFMA  <->  voxel ks*32 + (j < 4 ? 4g + j : 16 + 4g + j - 4): any bijection
        // works as long as A and B agree; this one makes each half-wave read 8 consecutive voxel rows
        const int v1 = ks * 32 + g * 4 + q, v2 = v1 + 16;
        constexpr int LS = S2 ? 2 : 1;                   // LDS voxel rows per q step
        const int xa1 = (((v1 / (BW * BH)) * LS * EH + ((v1 / BW) % BH) * LS) * EW + (v1 % BW) * LS) * RSX + (16 * wx + 4 * p) * 2;
        const int xa2 = (((v2 / (BW * BH)) * LS * EH + ((v2 / BW) % BH) * LS) * EW + (v2 % BW) * LS) * RSX + (16 * wx + 4 * p) * 2;
        s16x4 alo[MI], ahi[MI];
        if constexpr (DMA) {
          // unpadded 128-byte rows, 32-byte channel groups swizzled by ((w >> 1) & 3): w = 4g + q for both voxel halves (rows h and h + 1)
          const int wv = 4 * g + q, keyA = (wv >> 1) & 3;
          const unsigned char* yb = lds + cb * DBUF + v1 * 128 + 8 * p;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            alo[i] = tr_read(yb + ((i ^ keyA) << 5));
            ahi[i] = tr_read(yb + ((i ^ keyA) << 5) + 16 * 128);
          }
        } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          alo[i] = tr_read(ldsY + v1 * RSY + (16 * i + 4 * p) * 2);
          ahi[i] = tr_read(ldsY + v2 * RSY + (16 * i + 4 * p) * 2);
        }
        }
        typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
        bfx8 af[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i] = __builtin_bit_cast(bfx8, s16x8{alo[i][0], alo[i][1], alo[i][2], alo[i][3], ahi[i][0], ahi[i][1], ahi[i][2], ahi[i][3]});
        if constexpr (NTAP == 9 && BW == 16 && !S2) {
          // conv k3 unit = 9 taps (th, tw) of one d-shift (tap t = 3*th + tw, window offset (th*EW + tw) rows).  The two voxel
          // halves of a k-step are the h-rows h and h+1 of the brick, so tap th needs the X rows h+th and h+th+1: the three th of
          // a tw share 4 rows -- 12 transposing reads per k-step instead of 18
          s16x4 xr[3][4];
          if constexpr (DMA) {
            const int wv = 4 * g + q;
            const unsigned char* xb = lds + cb * DBUF + DYB + ((ks * 2) * 18 + wv) * 128 + 8 * p;     // X brick row (h = 2 ks, w) of this lane
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              const int keyB = ((wv + tw) >> 1) & 3;
#pragma unroll
              for (int r = 0; r < 4; ++r) xr[tw][r] = tr_read(xb + (r * 18 + tw) * 128 + ((wx ^ keyB) << 5));
            }
          } else {
#pragma unroll
          for (int tw = 0; tw < 3; ++tw)
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[tw][r] = tr_read(ldsX + xa1 + (r * EW + tw) * RSX);
          }
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const s16x4 lo = xr[t % 3][t / 3], hi = xr[t % 3][t / 3 + 1];
            const bfx8 bf = __builtin_bit_cast(bfx8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[t][i], 0, 0, 0);
          }
        } else {
          s16x4 blo[NTAP], bhi[NTAP];
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            blo[t] = tr_read(ldsX + xa1 + tapoff[t]);
            bhi[t] = tr_read(ldsX + xa2 + tapoff[t]);
          }
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const bfx8 bf = __builtin_bit_cast(bfx8, s16x8{blo[t][0], blo[t][1], blo[t][2], blo[t][3], bhi[t][0], bhi[t][1], bhi[t][2], bhi[t][3]});
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[t][i], 0, 0, 0);
          }
        }
      } else {
        const int v = ks * 4 + g;
        const int xa = (((v / (BW * BH)) * EH + (v / BW) % BH) * EW + v % BW) * RSX + (16 * wx + r16) * 4;
        float af[MI], bf[NTAP];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const float*)(ldsY + v * RSY + (16 * i + r16) * 4);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) bf[t] = *(const float*)(ldsX + xa + tapoff[t]);
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[t], acc[t][i], 0, 0, 0);
      }
    }
    if (!AM_DBG(a, 32)) __builtin_amdgcn_s_setprio(0);
  };
  int mask_b = -1;                                       // sample whose patch masks sit in LDS
  uint8_t* const ldsMy = lds + a.mask_off;               // (mask_n != 0 only)
  uint8_t* const ldsMx = ldsMy + a.mask_n;
  auto load_masks = [&]() {
    __syncthreads();
    for (int i = tid; i < a.mask_n; i += 256) {
      ldsMy[i] = a.y_mask.m ? a.y_mask.m[(size_t)b * a.mask_n + i] : (uint8_t)1;
      ldsMx[i] = a.x_mask.m ? a.x_mask.m[(size_t)b * a.mask_n + i] : (uint8_t)1;
    }
    __syncthreads();
    mask_b = b;
  };
  const int u_end = a.walk ? nu : (slot >> 3) + 1, u_inc = a.walk ? S8 : u_end;     // walk 0: exactly one pass (u_inc >= 1 always)
  for (int u = slot >> 3; u < u_end; u += u_inc) {
  int nstep = brick1 - brick0;
  if (a.walk) {
    const int seg = u / ncx, col = c0 + u % ncx;
    bw_ = col % nbw_; bh_ = (col / nbw_) % nbh_; b = col / (nbw_ * nbh_);
    bd_ = seg * a.seg_len - 1;
    nstep = nbd_ - seg * a.seg_len < a.seg_len ? nbd_ - seg * a.seg_len : a.seg_len;
  }
  for (int step = 0; step < nstep; ++step) {
    if (a.walk) ++bd_;
    else if (a.plist) {
      // live bricks only, (active patch, brick inside it): a contiguous run of ALL bricks holds 40 % live ones on average, but the share
      // of one slot's run spreads by +-45 % (64 patches per run at 16^3 patches) and the slowest slot is the launch
      const int i = brick0 + step, ip = i / bpp_, j = i - ip * bpp_;
      const int pk = __builtin_amdgcn_readfirstlane(a.plist[ip]);
      b = (pk >> 24) & 255;
      bd_ = ((pk >> 16) & 255) * a.pbd + j / pbhw_; bh_ = ((pk >> 8) & 255) * a.pbh + (j / a.pbw) % a.pbh; bw_ = (pk & 255) * a.pbw + j % a.pbw;
    }
    else if (++bw_ == nbw_) { bw_ = 0; if (++bh_ == nbh_) { bh_ = 0; if (++bd_ == nbd_) { bd_ = 0; ++b; } } }
    const int q0d = bd_ * BD, q0h = bh_ * BH, q0w = bw_ * BW;
    if (a.y_uni && a.mask_n && !a.plist) {
      // (uniform) block-sparse dY whose brick lies inside ONE patch of a grid of whole bricks: one lookup keeps or skips the brick,
      // before anything else is computed for it (60 % of the visits end here)
      if (b != mask_b) load_masks();
      const int alive = ldsMy[(((q0d + pd) >> a.y_mask.bs) * a.y_mask.fh + ((q0h + ph) >> a.y_mask.bs)) * a.y_mask.fw + ((q0w + pw) >> a.y_mask.bs)];
      if (!__builtin_amdgcn_readfirstlane(alive)) continue;
    }
    // X brick origin in global voxels: sub-lattice index (q0 + min shift) * GS + parity of the unit
    const int i0d = q0d * a.QS + a.mind[grp] * GS_ + upd, i0h = q0h * a.QS + a.minh[grp] * GS_ + uph, i0w = q0w * a.QS + a.minw[grp] * GS_ + upw;
    const int o0d = q0d * OS_ + pd, o0h = q0h * OS_ + ph, o0w = q0w * OS_ + pw;
    // descriptors anchored at the first d-plane of this brick (32-bit offsets span a few planes only: any tensor size works)
    const int yd0 = o0d < Dy_ ? o0d : Dy_, xd0 = i0d < 0 ? 0 : (i0d > Dx_ ? Dx_ : i0d);
    const int ybaseB = ((((o0d - yd0) * Hy_ + o0h) * Wy_ + o0w) * a.Cy) * (int)sizeof(T);   // brick origin relative to plane yd0 of sample b
    const int xbaseB = ((((i0d - xd0) * Hx_ + i0h) * Wx_ + i0w) * a.Cx) * (int)sizeof(T);
    const size_t yleft = (size_t)(Dy_ - yd0) * yplane * sizeof(T), xleft = (size_t)(Dx_ - xd0) * xplane * sizeof(T);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(yg + ((size_t)b * Dy_ + yd0) * yplane), 0,
                                                                        (int)(yleft < 0x7fffff00ull ? yleft : 0x7fffff00ull), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xg + ((size_t)b * Dx_ + xd0) * xplane), 0,
                                                                        (int)(xleft < 0x7fffff00ull ? xleft : 0x7fffff00ull), 0x00020000);
    // interior brick of a dense tensor (wave-uniform): every row in range, the row offsets are used as they are
    const bool interior = !masked && i0d >= 0 && i0h >= 0 && i0w >= 0 && i0d + (ED - 1) * GS_ < Dx_ &&
                          i0h + (EH - 1) * GS_ < Hx_ && i0w + (EW - 1) * GS_ < Wx_ && o0d + (BD - 1) * OS_ < Dy_ && o0h + (BH - 1) * OS_ < Hy_ &&
                          o0w + (BW - 1) * OS_ < Wy_;
    unsigned yo[NITY], xo[NITX];
#pragma unroll
    for (int it = 0; it < NITY; ++it) yo[it] = yoffB[it];
#pragma unroll
    for (int it = 0; it < NITX; ++it) xo[it] = xoffB[it];
    if (!interior) {
      if (!masked) {                                     // border brick of a dense tensor: range tests only, branch-free
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
          const int od = o0d + (yq[it] & 255) * OS_, oh = o0h + ((yq[it] >> 8) & 255) * OS_, ow = o0w + (yq[it] >> 16) * OS_;
          if (!(od < Dy_ && oh < Hy_ && ow < Wy_)) yo[it] = OOB;
        }
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
          const int id = i0d + (xq[it] & 255) * GS_, ih = i0h + ((xq[it] >> 8) & 255) * GS_, iw = i0w + (xq[it] >> 16) * GS_;
          if (!((unsigned)id < (unsigned)Dx_ && (unsigned)ih < (unsigned)Hx_ && (unsigned)iw < (unsigned)Wx_)) xo[it] = OOB;
        }
      } else if (a.mask_n) {                             // block-sparse operands, the sample's patch masks cached in LDS
        // (a global lookup per row puts two dependent memory round trips -- dY rows, then X rows -- in front of the brick's
        // loads, and one in front of every skipped brick: the sparse levels ran at 6 us per brick for 0.6 us of MFMAs)
        if (b != mask_b) load_masks();                   // (uniform) first brick of a sample
        const int ybs = a.y_mask.bs, xbs = a.x_mask.bs, mfh = a.y_mask.fh, mfw = a.y_mask.fw;
        // The lookups are UNCONDITIONAL LDS reads (index 0 for rows out of range), all issued before the first one is used: the
        // branchy per-row form cost one LDS round trip + ~30 instructions per row, 11 rows per brick -- three quarters of the
        // block-sparse launches' time went into deciding what to load (profiles/r03_experiments.md).
        if (a.y_uni) {
          // (decided at the top of the step: one lookup per brick, no vote, no row tests)
        } else if (a.y_mask.m) {
          uint8_t my[NITY];
#pragma unroll
          for (int it = 0; it < NITY; ++it) {
            const int od = o0d + (yq[it] & 255) * OS_, oh = o0h + ((yq[it] >> 8) & 255) * OS_, ow = o0w + (yq[it] >> 16) * OS_;
            const bool inr = yo[it] != OOB && od < Dy_ && oh < Hy_ && ow < Wy_;
            my[it] = ldsMy[inr ? ((od >> ybs) * mfh + (oh >> ybs)) * mfw + (ow >> ybs) : 0];
            if (!inr) yo[it] = OOB;
          }
          unsigned yany = 0;
#pragma unroll
          for (int it = 0; it < NITY; ++it) {
            if (my[it] == 0) yo[it] = OOB;
            yany |= yo[it] != OOB ? 1u : 0u;
          }
          if (!__syncthreads_or(yany != 0)) continue;    // nothing active in this brick (block-sparse dY)
        } else {
#pragma unroll
          for (int it = 0; it < NITY; ++it) {
            const int od = o0d + (yq[it] & 255) * OS_, oh = o0h + ((yq[it] >> 8) & 255) * OS_, ow = o0w + (yq[it] >> 16) * OS_;
            if (!(od < Dy_ && oh < Hy_ && ow < Wy_)) yo[it] = OOB;
          }
        }
        uint8_t mx[NITX];
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
          const int id = i0d + (xq[it] & 255) * GS_, ih = i0h + ((xq[it] >> 8) & 255) * GS_, iw = i0w + (xq[it] >> 16) * GS_;
          const bool inr = xo[it] != OOB && (unsigned)id < (unsigned)Dx_ && (unsigned)ih < (unsigned)Hx_ && (unsigned)iw < (unsigned)Wx_;
          mx[it] = ldsMx[inr ? ((id >> xbs) * mfh + (ih >> xbs)) * mfw + (iw >> xbs) : 0];
          if (!inr) xo[it] = OOB;
        }
#pragma unroll
        for (int it = 0; it < NITX; ++it)
          if (mx[it] == 0) xo[it] = OOB;
      } else {                                           // (masks too large for LDS: global patch-mask lookups per row)
        unsigned yany = 0;
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
          const int od = o0d + (yq[it] & 255) * OS_, oh = o0h + ((yq[it] >> 8) & 255) * OS_, ow = o0w + (yq[it] >> 16) * OS_;
          const bool ok = yo[it] != OOB && od < Dy_ && oh < Hy_ && ow < Wy_ && a.y_mask.active(b, od, oh, ow);
          if (!ok) yo[it] = OOB;
          yany |= ok ? 1u : 0u;
        }
        if (a.y_mask.m && !__syncthreads_or(yany != 0)) continue;    // nothing active in this brick (block-sparse dY)
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
          const int id = i0d + (xq[it] & 255) * GS_, ih = i0h + ((xq[it] >> 8) & 255) * GS_, iw = i0w + (xq[it] >> 16) * GS_;
          const bool ok = xo[it] != OOB && (unsigned)id < (unsigned)Dx_ && (unsigned)ih < (unsigned)Hx_ && (unsigned)iw < (unsigned)Wx_ &&
                          a.x_mask.active(b, id, ih, iw);
          if (!ok) xo[it] = OOB;
        }
      }
    }
    // buffer loads: per-lane 32-bit byte offset (row offset + the brick's scalar origin, which may be negative for halo rows that
    // are then OOB-marked); hardware zero-fill for OOB rows
    if constexpr (DMA) {
      static_assert(!DMA || (sizeof(T) == 2 && BD == 1 && (BH == 8 || BH == 4) && BW == 16 && NTAP == 9 && MI == 4 && NWX == 4 && NITX == (BH == 8 ? 6 : 4) && !S2 && !PF), "DMA variant: dense bf16 plane bricks");
      if (have) {                                        // the previous brick has landed (this wave's part; the barrier covers the others') and
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave is done reading the buffer this issue overwrites (contracted two bricks ago)
        __builtin_amdgcn_s_barrier();
      }
      const int wv_ = __builtin_amdgcn_readfirstlane(wave);
      typedef __attribute__((address_space(3))) void* ldsp_t;
#pragma unroll
      for (int it = 0; it < NITY; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (ldsp_t)(lds + dcur * DBUF + it * 4096 + wv_ * 1024), 16,
                                                 (int)((yo[it] == OOB || AM_DBG(a, 4)) ? OOB : yo[it] + (unsigned)ybaseB), 0, 0, 0);   // (an unsigned here: no stub is emitted for the host, silently)
#pragma unroll
      for (int it = 0; it < NITX; ++it)
        if (it * 32 + wv_ * 8 < XROWS)                   // (the last instruction round's rows beyond the brick are not allocated)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (ldsp_t)(lds + dcur * DBUF + DYB + it * 4096 + wv_ * 1024), 16,
                                                   (int)((xo[it] == OOB || AM_DBG(a, 4)) ? OOB : xo[it] + (unsigned)xbaseB), 0, 0, 0);
      if (have) contract(dcur ^ 1);                      // ... and flies while the previous brick is contracted
      have = true; dcur ^= 1;
    } else {
    if constexpr (PF) {
      if (have) { stage_to_lds(); }                      // the previous live brick's rows have landed: registers -> LDS
    }
#pragma unroll
    for (int it = 0; it < NITY; ++it)
      ys[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (yo[it] == OOB || AM_DBG(a, 4)) ? OOB : yo[it] + (unsigned)ybaseB, 0, 0));
#pragma unroll
    for (int it = 0; it < NITX; ++it)
      xs[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (xo[it] == OOB || AM_DBG(a, 4)) ? OOB : xo[it] + (unsigned)xbaseB, 0, 0));
    if constexpr (PF) {
      if (have) contract();                              // ... and this brick's loads fly while the previous one is contracted
      have = true;
    } else {
      stage_to_lds();
      contract();
    }
    }
  }
  }
  if constexpr (DMA) {
    if (have) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      contract(dcur ^ 1);
    }
    __syncthreads();                                     // (the deterministic flush reuses the staging area)
  }
  if constexpr (PF) {
    if (have) { stage_to_lds(); contract(); }
  }

  // ---- flush: D row = cy 4g+r, col = cx r16 ----
  const int cx = cx0 + 16 * wx + r16;
  if (a.det_ws) {
    // deterministic mode: no atomics.  The KS waves that split the k-steps fold through LDS in a fixed order, then every
    // (slot, tap, cy, cx) partial sum is stored exactly once; conv_wgrad_fold_kernel adds the slots in slot order.
    if constexpr (KS > 1) {
      float* red = (float*)lds;                              // [NWX][MI][4][64] per tap (<= 16 KB; the staging area is dead)
      for (int k = KS - 1; k > 0; --k) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          __syncthreads();
          if (wk == k) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) red[((wx * MI + i) * 4 + r) * 64 + lane] = acc[t][i][r];
          }
          __syncthreads();
          if (wk == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][i][r] += red[((wx * MI + i) * 4 + r) * 64 + lane];
          }
        }
      }
    }
    if (wk == 0 && cx < a.Cx) {
      float* ws = a.det_ws + (size_t)slot * a.det_stride;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int widx = a.taps[tb + t] >> 12;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cy = cy0 + 16 * i + 4 * g + r;
            if (cy < a.Cy) ws[((size_t)widx * a.Cy + cy) * a.Cx + cx] = acc[t][i][r];
          }
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    if (cx < a.Cx && !AM_DBG(a, 1)) {
      const int widx = a.taps[tb + t] >> 12;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cy = cy0 + 16 * i + 4 * g + r;
          if (cy < a.Cy) atomicAdd(a.dw + ((size_t)widx * a.Cy + cy) * a.Cx + cx, acc[t][i][r]);
        }
    }
  }
}

// deterministic mode: dw[tap][cy][cx] += sum over slots (in slot order) of the per-slot partial sums of this launch's taps
__global__ __launch_bounds__(256) void conv_wgrad_fold_kernel(WgArgs a, int ntap_launch) {
  const long per_tap = (long)a.Cy * a.Cx;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_tap * ntap_launch) return;
  int j = (int)(i / per_tap);                                 // j-th tap of this launch -> its widx
  const long e = i % per_tap;
  int widx = 0;
  for (int gi = 0; gi < a.ngroup; ++gi) {
    const int un = a.zmap[gi], n = a.tap_begin[un + 1] - a.tap_begin[un];
    if (j < n) { widx = a.taps[a.tap_begin[un] + j] >> 12; break; }
    j -= n;
  }
  const float* ws = a.det_ws + (size_t)widx * per_tap + e;
  float s = 0.f;
  for (int sl = 0; sl < a.split; ++sl) s += ws[(size_t)sl * a.det_stride];
  a.dw[(size_t)widx * per_tap + e] += s;
}

// Workgroups per launch: the kernel is persistent-style (each workgroup walks a contiguous run of bricks), so the grid must be a
// whole number of "rounds" of the resident set -- (resident workgroups per CU) x CUs, counted per XCD.  One workgroup more than
// two rounds costs a third round with the chip empty (measured: 1026 workgroups 2.24 ms, 1008 workgroups 1.77 ms).
constexpr int AM_WG_ROUNDS = 2;

template <typename T, int BD, int BH, int BW, int NTAP, int NITX, int MI = 4, int NWX = 4, bool S2 = false, bool PF = false, bool DMA = false>
int launch(WgArgs& a, size_t maxvox, int tiles, int nbrick, int det_slots, hipStream_t st) {
  auto kern = conv_wgrad_kernel<T, BD, BH, BW, NTAP, NITX, MI, NWX, S2, PF, DMA>;
  constexpr size_t RP = sizeof(T) == 2 ? 32 : 16;
  constexpr int CT = 16 * MI, KT = 16 * NWX;
  size_t lds = (size_t)BD * BH * BW * (CT * sizeof(T) + RP) + maxvox * (KT * sizeof(T) + RP);
  if (DMA) lds = 2 * ((size_t)BD * BH * BW * 128 + (size_t)(((BH + 2) * 18 + 7) / 8 * 8) * 128);   // two buffers of unpadded rows (1x8x16: 79 872 bytes; 1x4x16: 45 056)
  a.mask_off = 0; a.mask_n = 0;
  {
    const int pq = 1 << a.y_mask.bs;                 // patch edge in dY voxels
    a.y_uni = a.y_mask.m && a.OS == 1 && pq % BD == 0 && pq % BH == 0 && pq % BW == 0 && a.Dy % BD == 0 && a.Hy % BH == 0 && a.Wy % BW == 0;
    // (measured, profiles/r04_experiments.md: 16^3 patches -5 ... -9 % on the level-0 launches of STUNet-B / L / H, 4^3 patches -6 %; 8^3 patches
    // +-5 % either way -- a run of 8-wide bricks re-reads its w-halo from the neighbouring brick of the same patch row either way)
    if (a.y_uni && a.plist && a.nlive > 0 && pq != 8 && a.B <= 255 && a.y_mask.fd <= 255 && a.y_mask.fh <= 255 && a.y_mask.fw <= 255) {
      a.pbd = pq / BD; a.pbh = pq / BH; a.pbw = pq / BW;
      a.nlive *= a.pbd * a.pbh * a.pbw;                  // (entry: active patches) -> live bricks
      nbrick = a.nlive;
    } else {
      a.plist = nullptr; a.nlive = 0; a.pbd = a.pbh = a.pbw = 1;
    }
  }
  if (a.x_mask.m || a.y_mask.m) {                 // one sample's patch masks ride along in LDS when they are small (8^3 .. 12^3 patches)
    const MaskView& mv = a.y_mask.m ? a.y_mask : a.x_mask;
    const int n = mv.fd * mv.fh * mv.fw;
    if (n <= 4096) { a.mask_off = (int)((lds + 15) & ~(size_t)15); a.mask_n = n; lds = (size_t)a.mask_off + 2 * (size_t)n; }
  }
  if (lds > 160 * 1024) return -3;
  if (maxvox * (KT / TT<T>::EPC) > (size_t)NITX * 256) return -3;
  static PerDeviceOnce lds_cap;                   // per instantiation and device
  static int regs_occ = 2;                        // resident workgroups per CU the register file allows (LDS is accounted per launch; a property of the code object, the same on every gfx950)
  lds_cap.run([&](int) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, 0) == hipSuccess && nb > 0) regs_occ = nb;
    (void)hipGetLastError();
  });
  int occ = (int)((size_t)160 * 1024 / lds);
  if (occ > regs_occ) occ = regs_occ;
  if (occ < 1) occ = 1;
  // Workgroup ids go round-robin over the 8 XCDs and gridDim.x is a multiple of 8, so XCD x runs the slots with slot % 8 == x, times
  // `tiles`: choose slots-per-XCD so that no XCD gets one workgroup more than R whole rounds of its resident set
  int cap = wg_slots(occ) / 8;                             // resident workgroups per XCD
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_SPARE"); if (e_ && atoi(e_) > 0 && cap > atoi(e_)) cap -= atoi(e_); }   // tools: leave slots free for the other stream's small kernels
#endif
  int split8 = 0;
  double best = 0.0;
  int R0 = AM_WG_ROUNDS;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_ROUNDS"); if (e_) R0 = atoi(e_); }
#endif
  for (int R = R0; R <= 2 * R0 || (!split8 && R <= 8 * R0); ++R) {
    const int s8 = cap * R / tiles;
    const double eff = (double)s8 * tiles / (cap * R);
    if (s8 >= 1 && eff > best + 0.02) { best = eff; split8 = s8; }
  }
  // more tiles than 16 rounds of the chip hold (STUNet-H's 1536-channel layers: 1728 tiles): the tiles alone fill the chip, so only
  // as many brick-walk slots as R0 rounds need -- every extra slot is another [taps][64][64] block of atomics per tile (with the 8
  // slots the whole-round rule ended up with, those layers flushed 2 GB per launch: 7.9 ms for five launches, 3.6 with one slot).
  // Between 432 and 768 tiles (768 / 1024 channels) 8 slots measured better than 2-3.
  int split = split8 ? split8 * 8 : (cap * 8 * R0 + tiles - 1) / tiles;
  if (split < 1) split = 1;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_OLDSPLIT"); if (e_ && atoi(e_)) split = (1024 + tiles - 1) / tiles; }
#endif
  // every slot flushes a [taps][64][64] block of fp32 atomics per tile (147 KB for 9 taps: ~12 us beside ~2.4 us of work per brick): a slot
  // wants >= 64 bricks where the launch has them, even if that leaves part of the chip without a workgroup (STUNet-B 256->256 @16^3 at batch 4:
  // 32 slots of 4 bricks each = 3 rounds of mostly atomics, 213 us for 80 us of work).  tools/ab_minb.sh, same box: the batch-4 step 36.3 ms
  // without the rule, 35.6 / 35.3 / 35.1 / 36.3 with 16 / 32 / 64 / 128; batch 16, STUNet-L, STUNet-H unchanged
  {
    // (one-tap launches -- the 1x1 stride-2 shortcuts -- flush 16 KB per slot and tile, and a brick of theirs is one load latency, not 2.4 us
    // of MFMAs: 64 bricks in a row were 0.2 ms for 1 GFLOP at ANY batch size; 8 bricks per slot there)
    int minb = NTAP * a.ngroup >= 8 ? 64 : 8;
#ifdef AM_ABLATE
    { const char* e_ = getenv("AM_WG_MINB"); if (e_) minb = atoi(e_); }
#endif
    if (minb > 0 && sizeof(T) == 2) {                  // (bf16 launches: with the exact-f32 MFMA a brick is 16 x the work and the flush does not show)
      int capb = nbrick / minb / 8 * 8;
      if (capb < 8) capb = 8;
      // the tiles alone are a whole round of the chip (STUNet-B's 512 -> 512 transposed conv: 8 x 8 tiles x 8 parity groups): more slots are
      // more ROUNDS of the same bricks plus a [taps][64][64] flush each -- 8 slots of 2 bricks flushed 8 x 67 MB for 68 GFLOP at batch 4
      // (0.42 ms at 162 TFLOP/s).  No XCD needs a slot of its own there (the live slots rotate with the tile index in the kernel).
      bool tile_round = tiles >= cap * 8;
#ifdef AM_ABLATE
      { const char* e_ = getenv("AM_WG_TILECAP"); if (e_ && !atoi(e_)) tile_round = false; }
#endif
      if (tile_round) { capb = nbrick / minb; if (capb < 1) capb = 1; }
      else {
        // ... and the other way round: a launch whose slots x tiles stay UNDER one round of the chip leaves CUs idle while the few workgroups walk
        // long rows of bricks (128 -> 128 @32^3 block-sparse at batch 4: 8 slots x 12 tiles = 96 workgroups of 100 bricks, 0.31 ms at 152
        // TFLOP/s): up to one round of slots there, walks of >= 16 bricks
        int fill = cap * 8 / tiles / 8 * 8, cap16 = nbrick / 16 / 8 * 8;
#ifdef AM_ABLATE
        { const char* e_ = getenv("AM_WG_FILL"); if (e_ && !atoi(e_)) fill = 0; }
#endif
        if (fill > cap16) fill = cap16;
        if (capb < fill) capb = fill;
      }
      if (split > capb) {
        // keep the capped count on whole rounds of the resident set (32 -> 32 @128^3 block-sparse at batch 4: 200 slots x 3 tiles = 600
        // workgroups = 1.17 rounds ran as two; 168 slots = 504 workgroups run as one)
        int aligned = 0;
        if (!tile_round)
          for (int R = 1; R <= 16; ++R) { const int s8 = cap * R / tiles; if (s8 * 8 > capb) break; if (s8 >= 1) aligned = s8 * 8; }
#ifdef AM_ABLATE
        { const char* e_ = getenv("AM_WG_ALIGN"); if (e_ && !atoi(e_)) aligned = 0; }
#endif
        split = aligned ? aligned : capb;
      }
    }
  }
  if (split > nbrick) split = nbrick;
  if (a.det_ws && split > det_slots) split = det_slots;     // (the caller checked det_slots >= 1)
  a.split = split;
  // d-fastest interleaved walk for dense operands when the slots split evenly over the XCDs (segments of <= 16 d-bricks keep the
  // slots of an XCD balanced to a segment; each segment start re-reads one d-halo)
  // (measured slower on the transposed convolutions' 8 parity groups: only the one-plane bricks of dense k3 s1 take it)
  a.walk = (BD == 1 && !a.x_mask.m && !a.y_mask.m && split % 8 == 0 && split >= 8) ? 1 : 0;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_WALK"); if (e_ && !atoi(e_)) a.walk = 0; }
#endif
  // segments as long as the balance of the slots allows: >= 48 work units per slot (a unit more or less = 2 %) but at least 8
  // bricks, since every segment start re-reads one d-halo (measured on 64->64 @128^3: profiles/r02_experiments.md)
  {
    const int ncx = (a.B * a.nbh * a.nbw + 7) / 8, s8 = split / 8 > 0 ? split / 8 : 1;
    int nseg = (48 * s8 + ncx - 1) / ncx;
    if (nseg > a.nbd / 8) nseg = a.nbd / 8;
    if (nseg < 1) nseg = 1;
    a.seg_len = (a.nbd + nseg - 1) / nseg;
    a.nseg = (a.nbd + a.seg_len - 1) / a.seg_len;
  }
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_SEG"); if (e_ && atoi(e_) > 0) { a.seg_len = atoi(e_) < a.nbd ? atoi(e_) : a.nbd; a.nseg = (a.nbd + a.seg_len - 1) / a.seg_len; } }
#endif
  a.ntile = ((a.Cy + CT - 1) / CT) * ((a.Cx + KT - 1) / KT);
  dim3 grid((unsigned)(((split + 7) / 8) * 8 * a.ngroup * a.ntile), 1, 1);
  AM_LAUNCH(kern, grid, dim3(256), lds, st, a);
  AM_CHECK_LAUNCH();
  if (a.det_ws) {
    const int ntl = NTAP * a.ngroup;
    const long n = (long)a.Cy * a.Cx * ntl;
    AM_LAUNCH(conv_wgrad_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, ntl);
    AM_CHECK_LAUNCH();
  }
  return 0;
}


// ================================================================== gather form (patches smaller than a brick)
// Levels whose patches are 1 or 2 voxels wide (the two deepest of a 16x down-sampling encoder) leave the brick kernel above with 30-40 %
// live voxels in every brick it stages, a few thousand voxels in total and a [27][Cy][Cx] block of atomics per brick-walk slot:
// 65-270 TFLOP/s on the 512 ... 1536-channel layers of STUNet-L / H.  Here the ACTIVE voxels are gathered once into K-major operands
//     YT[cy][n] = dY[q_n][cy],      XT[t][cx][n] = X[q_n * stride + shift_t][cx]   (0 outside the volume / in an inactive patch),
// n = 0 .. N-1 over the active dY voxels (patch list order), padded with zeros to a multiple of 32, and the weight gradient is 27 plain
// GEMMs  dW[t] += YT . XT[t]^T  whose fragments are 16-byte global loads (both operands K-contiguous: no LDS, no transposing reads).
struct WgGather {
  const bf16_t* x; const bf16_t* dy;
  int B, Dx, Hx, Wx, Cx, Dy, Hy, Wy, Cy;
  int stride, k, pq, pq3;              // patch edge of the dY grid in voxels (1 or 2)
  const int* plist; int N, Np;         // active patches; active dY voxels; Np = N rounded up to 32
  MaskView x_mask;
  int t0, nt;                          // taps [t0, t0 + nt) of this round
  bf16_t* yt; bf16_t* xt;              // [Cy][Np], [nt][Cx][Np]
};

__global__ __launch_bounds__(256) void wg_gather_t_kernel(WgGather a) {
  __shared__ bf16_t tile[32][72];                        // 32 voxels x 64 channels (+8: the transposed 2-byte reads spread over the banks)
  const int tid = threadIdx.x, n0 = blockIdx.x * 32;
  const bool is_y = blockIdx.y == 0;
  const int t = a.t0 + (int)blockIdx.y - 1;
  const int C = is_y ? a.Cy : a.Cx;
  const bf16_t* src = is_y ? a.dy : a.x;
  bf16_t* dst = is_y ? a.yt : a.xt + (size_t)(blockIdx.y - 1) * a.Cx * a.Np;
  // source voxel of this thread's row (v = tid / 8): -1 = zeros
  const int v = tid >> 3, ch = tid & 7, n = n0 + v;
  long row = -1;
  if (n < a.N) {
    const int ip = n / a.pq3, j = n - ip * a.pq3;
    const int pk = a.plist[ip];
    const int b = (pk >> 24) & 255;
    const int qd = ((pk >> 16) & 255) * a.pq + j / (a.pq * a.pq), qh = ((pk >> 8) & 255) * a.pq + (j / a.pq) % a.pq, qw = (pk & 255) * a.pq + j % a.pq;
    if (is_y) row = (((long)b * a.Dy + qd) * a.Hy + qh) * a.Wy + qw;
    else {
      const int pad = a.k / 2, td = t / (a.k * a.k), th = (t / a.k) % a.k, tw = t % a.k;
      const int id = qd * a.stride + td - pad, ih = qh * a.stride + th - pad, iw = qw * a.stride + tw - pad;
      if (id >= 0 && id < a.Dx && ih >= 0 && ih < a.Hx && iw >= 0 && iw < a.Wx && a.x_mask.active(b, id, ih, iw))
        row = (((long)b * a.Dx + id) * a.Hx + ih) * a.Wx + iw;
    }
  }
  for (int c0 = 0; c0 < C; c0 += 64) {
    u32x4 val = u32x4{0u, 0u, 0u, 0u};
    if (row >= 0 && c0 + ch * 8 < C) val = *(const u32x4*)(src + (size_t)row * C + c0 + ch * 8);
    __syncthreads();                                     // the previous slab's column reads are done
    *(u32x4*)&tile[v][ch * 8] = val;
    __syncthreads();
    const int c = tid >> 2, vg = tid & 3;                // channel c of the slab, voxels 8 vg .. 8 vg + 7
    if (c0 + c < C) {
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (unsigned)tile[8 * vg + 2 * i][c] | ((unsigned)tile[8 * vg + 2 * i + 1][c] << 16);
      *(u32x4*)(dst + (size_t)(c0 + c) * a.Np + n0 + 8 * vg) = o;
    }
  }
}

// dW[t][cy][cx] += sum_n YT[cy][n] XT[t][cx][n]: workgroup tile 128 cy x 64 cx, 2 x 2 waves of 64 x 32 (4 x 2 MFMA tiles), k-steps of 32 with the
// next step's fragments in flight; grid (tiles, taps, k splits).  One k split: every element has one writer, plain read-modify-write
// (deterministic); more: fp32 atomics.
__global__ __launch_bounds__(256) void wg_gemm_nt_kernel(const bf16_t* __restrict__ yt, const bf16_t* __restrict__ xt, float* __restrict__ dw,
                                                         int Cy, int Cx, int Np, int k, int t0, int ksplit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
  const int ntx = Cx / 64, ty = blockIdx.x / ntx, tx = blockIdx.x % ntx;
  const int cy0 = ty * 128 + (wave >> 1) * 64, cx0 = tx * 64 + (wave & 1) * 32;
  const bf16_t* A = yt + (size_t)(cy0 + r16) * Np + 8 * g;
  const bf16_t* Bm = xt + ((size_t)blockIdx.y * Cx + cx0 + r16) * Np + 8 * g;
  const int steps = Np / 32, per = (steps + ksplit - 1) / ksplit;
  const int s0 = blockIdx.z * per, s1 = s0 + per < steps ? s0 + per : steps;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (s0 < s1) {
    u32x4 a[4], b[2], an[4], bn[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const u32x4*)(A + (size_t)(16 * i) * Np + 32 * s0);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const u32x4*)(Bm + (size_t)(16 * j) * Np + 32 * s0);
    for (int s = s0; s < s1; ++s) {
      const int sn = s + 1 < s1 ? s + 1 : s;             // (the last step re-loads itself: no branch around the loads)
#pragma unroll
      for (int i = 0; i < 4; ++i) an[i] = *(const u32x4*)(A + (size_t)(16 * i) * Np + 32 * sn);
#pragma unroll
      for (int j = 0; j < 2; ++j) bn[j] = *(const u32x4*)(Bm + (size_t)(16 * j) * Np + 32 * sn);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma_chunk<bf16_t>(a[i], b[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = an[i];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = bn[j];
    }
  }
  // D row = cy 4g + r, col = cx r16
  float* out = dw + (size_t)(t0 + blockIdx.y) * Cy * Cx;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = out + (size_t)(cy0 + 16 * i + 4 * g + r) * Cx + cx0 + 16 * j + r16;
        if (ksplit == 1) *p += acc[i][j][r]; else atomicAdd(p, acc[i][j][r]);
      }
}

// the gather form applies: bf16, conv k3 (stride 1 or 2), block-sparse dY with 1- or 2-voxel patches and an active-patch list, channel
// counts that tile by 128 x 64
inline bool wg_gather_ok(int mode, int dtype, int k, int stride, int Cx, int Cy, const uint8_t* y_mask, int y_bshift, const int32_t* list,
                         int n_active, int B, int fd, int fh, int fw) {
  // where it was measured ahead of the brick kernel (profiles/r04_experiments.md): one-voxel patches, and the strided layers into 2-voxel patches from
  // 256 input channels on (the brick kernel stages 8 parity sub-lattices there).  Its GEMM is bound by L2 bandwidth at ~160 TFLOP/s (128 x 64
  // tiles: 43 flop per byte), which the stride-1 layers of the 2-voxel-patch level already reach on the brick kernel.
  if (!(y_bshift == 0 || (y_bshift == 1 && stride == 2 && Cx >= 256))) return false;
  return mode == AM_CONV_FWD && dtype == AM_DT_BF16 && k == 3 && (stride == 1 || stride == 2) && y_mask && list && n_active > 0 &&
         Cy % 128 == 0 && Cx % 64 == 0 && B <= 255 && fd <= 255 && fh <= 255 && fw <= 255;
}

}  // namespace

extern "C" int am_conv3d_wgrad(int mode, int dtype, int ksize, int stride, const void* x, const void* dy, float* dw_packed,
                               int B, int Dx, int Hx, int Wx, int Cx, int Dy, int Hy, int Wy, int Cy,
                               const uint8_t* x_mask, int x_bshift, const uint8_t* y_mask, int y_bshift,
                               int fd, int fh, int fw, float* det_workspace, long det_workspace_floats,
                               const int32_t* active_list, int n_active, void* gather_workspace, long gather_workspace_bytes, void* stream) {
  if (Cx % 8 || Cy % 8) return -1;
  if ((size_t)8 * Hx * Wx * Cx * 4 >= 0x7fffff00ull || (size_t)8 * Hy * Wy * Cy * 4 >= 0x7fffff00ull) return -5;   // planes a brick spans
  WgArgs a;
  a.x = x; a.dy = dy; a.dw = dw_packed;
  a.det_ws = det_workspace; a.det_stride = (long)ksize * ksize * ksize * Cy * Cx;
  const int det_slots = det_workspace ? (int)(det_workspace_floats / a.det_stride < 4096 ? det_workspace_floats / a.det_stride : 4096) : 0;
  if (det_workspace && det_slots < 1) return -6;               // workspace smaller than one [k^3][Cy][Cx] slot
  a.B = B; a.Dx = Dx; a.Hx = Hx; a.Wx = Wx; a.Cx = Cx; a.Dy = Dy; a.Hy = Hy; a.Wy = Wy; a.Cy = Cy;
  a.x_mask = MaskView{x_mask, fd, fh, fw, x_bshift};
  a.y_mask = MaskView{y_mask, fd, fh, fw, y_bshift};
  const int* const plist0 = (y_mask && active_list && n_active > 0) ? active_list : nullptr;
  const int k = ksize;
  if (gather_workspace && wg_gather_ok(mode, dtype, ksize, stride, Cx, Cy, y_mask, y_bshift, active_list, n_active, B, fd, fh, fw)) {
    WgGather ga;
    ga.x = (const bf16_t*)x; ga.dy = (const bf16_t*)dy;
    ga.B = B; ga.Dx = Dx; ga.Hx = Hx; ga.Wx = Wx; ga.Cx = Cx; ga.Dy = Dy; ga.Hy = Hy; ga.Wy = Wy; ga.Cy = Cy;
    ga.stride = stride; ga.k = ksize; ga.pq = 1 << y_bshift; ga.pq3 = ga.pq * ga.pq * ga.pq;
    ga.plist = active_list; ga.N = n_active * ga.pq3; ga.Np = (ga.N + 31) / 32 * 32;
    ga.x_mask = MaskView{x_mask, fd, fh, fw, x_bshift};
    const long per_tap = (long)Cx * ga.Np * 2, ybytes = (long)Cy * ga.Np * 2;
    const int ntaps = ksize * ksize * ksize;
    int round = (int)((gather_workspace_bytes - ybytes) / per_tap);          // taps per round the workspace holds
    if (round >= 1) {
      if (round > ntaps) round = ntaps;
      hipStream_t st = (hipStream_t)stream;
      ga.yt = (bf16_t*)gather_workspace; ga.xt = (bf16_t*)((char*)gather_workspace + ybytes);
      const int tiles = (Cy / 128) * (Cx / 64);
      for (int t0 = 0; t0 < ntaps; t0 += round) {
        ga.t0 = t0; ga.nt = t0 + round <= ntaps ? round : ntaps - t0;
        // (every round rewrites the dY plane with the same values: one more row of blocks, no second kernel)
        AM_LAUNCH(wg_gather_t_kernel, dim3((unsigned)(ga.Np / 32), (unsigned)(ga.nt + 1), 1), dim3(256), 0, st, ga);
        AM_CHECK_LAUNCH();
        int ksplit = 1;
        if (!det_workspace && tiles * ga.nt < 512) { ksplit = (512 + tiles * ga.nt - 1) / (tiles * ga.nt); if (ksplit > ga.Np / 64) ksplit = ga.Np / 64 > 0 ? ga.Np / 64 : 1; }
        AM_LAUNCH(wg_gemm_nt_kernel, dim3((unsigned)tiles, (unsigned)ga.nt, (unsigned)ksplit), dim3(256), 0, st, ga.yt, ga.xt, dw_packed, Cy, Cx, ga.Np,
                  ksize, t0, ksplit);
        AM_CHECK_LAUNCH();
      }
      return 0;
    }
  }
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_DBG"); a.dbg = e_ ? atoi(e_) : 0; }
#endif
  const bool bf = dtype == AM_DT_BF16;
  if (bf && mode == AM_CONV_FWD && ksize == 3 && stride == 1 && !x_mask && !y_mask && !det_workspace && Dx == Dy && Hx == Hy && Wx == Wy) {
    const int rc = amconv::conv_wgk3_launch(x, dy, dw_packed, B, Dy, Hy, Wy, Cx, Cy, stream);
    if (rc < 0) return rc;                                 // (a failed launch: never "served", never a silent fall-through)
    if (rc == 1) return 0;
  }
  // units: taps that share the dY operand AND one dense X sub-brick
  //   conv stride 1: one unit per d-tap (9 taps, no d-halo);  ConvT: the 8 output parities (8 taps each);
  //   conv stride 2: the 8 parity sub-lattices of X (1,2,2,2,4,4,4,8 taps): X[2q + s] = X_sub[r][q + u], s = 2u + r
  int nunit;
  a.OS = 1; a.GS = 1;
  const int Qw0 = Wy, Qh0 = Hy;
  // stride-2 k3 on grids at least one 1x4x16 brick wide: full-resolution staging, one unit per d-tap (kernel variant S2)
  bool s2full = bf && mode == AM_CONV_FWD && k == 3 && stride == 2 && Qw0 >= 16 && Qh0 >= 4;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_S2FULL"); if (e_ && !atoi(e_)) s2full = false; }
#endif
  if (mode == AM_CONV_FWD) {
    if (k != 1 && k != 3) return -2;
    if (s2full) nunit = 3;
    else if (stride == 2) { a.GS = 2; nunit = 8; } else nunit = k;
  } else if (mode == AM_CONVT_FWD) {
    if (k != 4 || stride != 2) return -2;
    a.OS = 2; nunit = 8;
  } else return -2;
  const int Qd = (Dy + a.OS - 1) / a.OS, Qh = (Hy + a.OS - 1) / a.OS, Qw = (Wy + a.OS - 1) / a.OS;
  // q-brick per staging round: bf16 (k-steps of 32 voxels) 2x4x16 / 2x8x8 = 128 voxels, ~55 KB of LDS so two workgroups
  // share a CU and overlap each other's staging;  f32 (k-steps of 4): 2x8x8.
  int bd = 2, bh = 8, bw = 8;
  // block-sparse dY with patches 8 voxels wide: the 8x8 brick lies inside ONE patch (60 % of the bricks are skipped on one lookup; a
  // 16-wide brick spans two patches, is empty only 36 % of the time and needs a mask test per row)
  const bool patch8 = y_mask && (1 << y_bshift) == 8 && Hy % 8 == 0 && Wy % 8 == 0 && Dy % 2 == 0;
  if (bf && Qw >= 16 && !(patch8 && Cx > 32 && Cy > 32)) { bh = 4; bw = 16; }
  // ... with patches 4 voxels wide: the 4x4x4 brick IS the patch (64 voxels, two k-steps: short, but the 2x4x16 brick spans four patches,
  // is live 87 % of the time for 40 % of its voxels and tests the mask per row)
  bool patch4 = bf && y_mask && (1 << y_bshift) == 4 && Dy % 4 == 0 && Hy % 4 == 0 && Wy % 4 == 0 && mode == AM_CONV_FWD && k == 3 && stride == 1 && Cx > 32 && Cy > 32;
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_PATCH4"); if (e_ && !atoi(e_)) patch4 = false; }
#endif
  if (patch4) { bd = 4; bh = 4; bw = 4; }
  // dense k3 s1 with 64-wide tiles: one-plane 1x8x16 bricks -- the X brick of a tap group has no d-halo at all (10x18 voxels
  // for 8x16: 1.41x, against 1.69x for 2x4x16) and the d-fastest walk re-reads each plane from L2
  bool plane_brick = bf && Qw >= 16 && Qh >= 8 && mode == AM_CONV_FWD && k == 3 && stride == 1 && !x_mask && !y_mask && (Cx > 32 || Cy > 32);
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_PLANE"); if (e_ && !atoi(e_)) plane_brick = false; }
#endif
  if (plane_brick) { bd = 1; bh = 8; bw = 16; }
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_WG_DMA_BH4"); if (e_ && atoi(e_) && plane_brick && Cx % 64 == 0 && Cy % 64 == 0 && mi_is4(Cx, Cy)) bh = 4; }   // tools: 1x4x16 DMA bricks (45 KB of LDS per workgroup)
#endif
  if (s2full) { bd = 1; bh = 4; bw = 16; if (patch8) { bh = 8; bw = 8; } }
  // 32-channel operands: 32-wide tiles (MI = 2 cy tiles / NWX = 2 cx waves, the other waves split the voxels)
  int mi = 4, nwx = 4;
  if (s2full) nwx = 2;                                    // 32-channel X tiles: the 9 x 33 full-resolution rows stay at 96 bytes
  else if (bf && bw == 16 && mode == AM_CONV_FWD && (k == 3 || stride == 2)) {
    if (Cy <= 32 && stride == 1) mi = 2;
#ifdef AM_ABLATE
    { const char* e_ = getenv("AM_WG_MI2"); if (e_ && atoi(e_) && stride == 1 && k == 3 && !x_mask && !y_mask) mi = 2; }   // tools: 32-wide cy tiles everywhere (161 VGPRs: room for a streaming wave beside two weight-gradient waves per SIMD)
#endif
    // dY channel counts that 64-wide tiles pad by a quarter or more (STUNet-H: 96 -> 128): 32-wide cy tiles (three for 96, no padding)
    if (Cy > 64 && stride == 1 && ((Cy + 63) / 64 * 64 - Cy) * 4 >= Cy) mi = 2;
    if (Cx <= 32) nwx = 2;
    else if (mi == 4 && Cx > 64 && ((Cx + 63) / 64 * 64 - Cx) * 4 >= Cx) nwx = 2;     // likewise for X (64 x 32 tiles; not both: 32 x 32 tiles are LDS-read bound)
    if (mi == 2 && nwx == 2) bd = 4;                     // thin rows: a 256-voxel brick still fits two workgroups per CU
  }
  const int pad = (mode == AM_CONVT_FWD) ? 1 : k / 2;
  int n = 0; size_t maxvox = 0;
  int ucount[8];
  for (int gI = 0; gI < nunit; ++gI) {
    a.tap_begin[gI] = n;
    const int p[3] = {(gI >> 2) & 1, (gI >> 1) & 1, gI & 1};
    a.pofs[gI] = (mode == AM_CONVT_FWD) ? gI : 0;
    a.upar[gI] = a.GS == 2 ? gI : 0;
    int mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; bool first = true;
    for (int td = 0; td < k; ++td) for (int th = 0; th < k; ++th) for (int tw = 0; tw < k; ++tw) {
      const int t[3] = {td, th, tw};
      int u[3]; bool ok = true;
      for (int d = 0; d < 3 && ok; ++d) {
        if (mode == AM_CONVT_FWD) { const int num = p[d] + pad - t[d]; if (num & 1) ok = false; else u[d] = num / 2; }
        else if (a.GS == 2) { const int sft = t[d] - pad, r = ((sft % 2) + 2) % 2; if (r != p[d]) ok = false; else u[d] = (sft - r) / 2; }
        else u[d] = t[d] - pad;                           // (S2: shifts in full-resolution voxels around 2q)
      }
      if (ok && mode == AM_CONV_FWD && a.GS == 1 && td != gI) ok = false;
      if (!ok) continue;
      a.taps[n++] = (u[0] + 8) | ((u[1] + 8) << 4) | ((u[2] + 8) << 8) | ((td * k * k + th * k + tw) << 12);
      for (int d = 0; d < 3; ++d) { if (first || u[d] < mn[d]) mn[d] = u[d]; if (first || u[d] > mx[d]) mx[d] = u[d]; }
      first = false;
    }
    ucount[gI] = n - a.tap_begin[gI];
    a.mind[gI] = mn[0]; a.minh[gI] = mn[1]; a.minw[gI] = mn[2];
    const int ls = s2full ? 2 : 1;                        // staged voxel rows per q step
    a.ed[gI] = (bd - 1) * ls + 1 + (mx[0] - mn[0]);
    a.eh[gI] = (bh - 1) * ls + 1 + (mx[1] - mn[1]);
    a.ew[gI] = (bw - 1) * ls + 1 + (mx[2] - mn[2]);
    a.mdiv_w[gI] = (1 << 20) / a.ew[gI] + 1;
    a.mdiv_hw[gI] = (1 << 20) / (a.ew[gI] * a.eh[gI]) + 1;
    const size_t v = (size_t)a.ed[gI] * a.eh[gI] * a.ew[gI];
    if (v > maxvox) maxvox = v;
  }
  for (int gI = nunit; gI <= 8; ++gI) a.tap_begin[gI] = n;
  a.QS = s2full ? 2 : a.GS;
  a.nbd = (Qd + bd - 1) / bd; a.nbh = (Qh + bh - 1) / bh; a.nbw = (Qw + bw - 1) / bw;
  const int nbrick = B * a.nbd * a.nbh * a.nbw;
  hipStream_t st = (hipStream_t)stream;
  // one launch per tap count: the kernel's tap loop is compile-time
  static const int counts[5] = {9, 8, 4, 2, 1};
  for (int ci = 0; ci < 5; ++ci) {
    const int ntap = counts[ci];
    a.ngroup = 0;
    for (int gI = 0; gI < nunit; ++gI) if (ucount[gI] == ntap) a.zmap[a.ngroup++] = gI;
    if (!a.ngroup) continue;
    const int tiles = ((Cy + 16 * mi - 1) / (16 * mi)) * ((Cx + 16 * nwx - 1) / (16 * nwx)) * a.ngroup;
    a.plist = plist0; a.nlive = plist0 ? n_active : 0; a.pbd = a.pbh = a.pbw = 1;      // (launch() turns patches into live bricks of ITS brick shape)
    int rc = -2;
#define WG_CASE(TT_, BH_, BW_, NT_, NX_) rc = launch<TT_, 2, BH_, BW_, NT_, NX_>(a, maxvox, tiles, nbrick, det_slots, st)
    if (s2full) rc = bw == 8 ? launch<bf16_t, 1, 8, 8, 9, 5, 4, 2, true, true>(a, maxvox, tiles, nbrick, det_slots, st)
                             : launch<bf16_t, 1, 4, 16, 9, 5, 4, 2, true, true>(a, maxvox, tiles, nbrick, det_slots, st);
    else if (bf && bd == 1 && ntap == 9 && (mi == 2 || nwx == 2)) {
      if (mi == 2) rc = launch<bf16_t, 1, 8, 16, 9, 6, 2, 4>(a, maxvox, tiles, nbrick, det_slots, st);
      else rc = launch<bf16_t, 1, 8, 16, 9, 3, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
    } else if (bf && bw == 16 && (mi == 2 || nwx == 2)) {
      if (ntap == 9 && mi == 2 && nwx == 2) rc = launch<bf16_t, 4, 4, 16, 9, 7, 2, 2, false, true>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 9 && mi == 2) rc = launch<bf16_t, 2, 4, 16, 9, 7, 2, 4>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 9) rc = launch<bf16_t, 2, 4, 16, 9, 4, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 8) rc = launch<bf16_t, 2, 4, 16, 8, 4, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 4) rc = launch<bf16_t, 2, 4, 16, 4, 4, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 2) rc = launch<bf16_t, 2, 4, 16, 2, 4, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
      else rc = launch<bf16_t, 2, 4, 16, 1, 4, 4, 2>(a, maxvox, tiles, nbrick, det_slots, st);
    } else if (bf && bw == 4) {
      rc = ntap == 9 ? launch<bf16_t, 4, 4, 4, 9, 5>(a, maxvox, tiles, nbrick, det_slots, st) : -2;
    } else if (bf && bw == 16) {
#ifdef AM_ABLATE
      // LDS-DMA staging (tools build only, AM_WG_DMA=1 / AM_WG_DMA_BH4=1): +5 % on the launch alone, -1.3 ms on the STEP -- two double-buffered
      // workgroups fill a CU's LDS and nothing of the main stream runs beside them any more (profiles/r04_experiments.md section 6)
      if (ntap == 9 && bd == 1 && bh == 4 && !x_mask && !y_mask && Cx % 64 == 0 && Cy % 64 == 0 && maxvox <= 108)
        rc = launch<bf16_t, 1, 4, 16, 9, 4, 4, 4, false, false, true>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 9 && bd == 1 && !x_mask && !y_mask && Cx % 64 == 0 && Cy % 64 == 0 && maxvox <= 180 && wg_dma())
        rc = launch<bf16_t, 1, 8, 16, 9, 6, 4, 4, false, false, true>(a, maxvox, tiles, nbrick, det_slots, st);
      else
#endif
      if (ntap == 9 && bd == 1) rc = launch<bf16_t, 1, 8, 16, 9, 6>(a, maxvox, tiles, nbrick, det_slots, st);
      else if (ntap == 9) WG_CASE(bf16_t, 4, 16, 9, 7); else if (ntap == 8) WG_CASE(bf16_t, 4, 16, 8, 8); else if (ntap == 4) WG_CASE(bf16_t, 4, 16, 4, 8);
      else if (ntap == 2) WG_CASE(bf16_t, 4, 16, 2, 8); else WG_CASE(bf16_t, 4, 16, 1, 8);
    } else if (bf) {
      if (ntap == 9) WG_CASE(bf16_t, 8, 8, 9, 7); else if (ntap == 8) WG_CASE(bf16_t, 8, 8, 8, 8); else if (ntap == 4) WG_CASE(bf16_t, 8, 8, 4, 8);
      else if (ntap == 2) WG_CASE(bf16_t, 8, 8, 2, 8); else WG_CASE(bf16_t, 8, 8, 1, 8);
    } else {
      if (ntap == 9) WG_CASE(float, 8, 8, 9, 13); else if (ntap == 8) WG_CASE(float, 8, 8, 8, 16); else if (ntap == 4) WG_CASE(float, 8, 8, 4, 16);
      else if (ntap == 2) WG_CASE(float, 8, 8, 2, 16); else WG_CASE(float, 8, 8, 1, 16);
    }
#undef WG_CASE
    if (rc) return rc;
  }
  return 0;
}

extern "C" int am_conv3d_wgrad_uses_k3(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cx, int Cy, int has_masks,
                                       int deterministic) {
  return (dtype == AM_DT_BF16 && mode == AM_CONV_FWD && ksize == 3 && stride == 1 && !has_masks && !deterministic &&
          amconv::conv_wgk3_qualifies(B, D, H, W, Cx, Cy)) ? 1 : 0;
}

extern "C" int am_conv3d_wgrad_gather_bytes(int mode, int dtype, int ksize, int stride, int B, int Cx, int Cy, int y_bshift, int has_y_mask, int n_active,
                                            int fd, int fh, int fw, long* bytes) {
  static const uint8_t one = 1; static const int32_t lst = 0;
  if (!bytes) return -1;
  *bytes = 0;
  if (!wg_gather_ok(mode, dtype, ksize, stride, Cx, Cy, has_y_mask ? &one : nullptr, y_bshift, &lst, n_active, B, fd, fh, fw)) return 0;
  const long pq3 = 1L << (3 * y_bshift), Np = ((long)n_active * pq3 + 31) / 32 * 32;
  const long ntaps = (long)ksize * ksize * ksize, all = 2 * Np * (Cy + ntaps * Cx), third = 2 * Np * (Cy + (ntaps / 3) * Cx);
  *bytes = all <= (640L << 20) ? all : third;            // all 27 taps in one round when that stays under 640 MB, else three rounds of 9
  return 0;
}
