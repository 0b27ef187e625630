// Gather implicit-GEMM 3-D convolution for gfx950 (MFMA), channels-last.
//
// One kernel serves every dense / block-sparse convolution on the AnatoMask path:
//   forward conv k1/k3 stride 1/2   (P/encoder3D.py:12-15 SparseConv3d, P/decoder3D.py:20-22, P/AnatoMask.py:63-65)
//   its data gradient               (autograd of the above)
//   ConvTranspose3d k4 s2 p1        (P/decoder3D.py:19) as 8 output-parity classes of 2x2x2 taps
//   its data gradient               (a k4 s2 p1 convolution of dy)
//
// Formulation.  Output voxels are enumerated as o = q*OS + p (OS in {1,2}; p = parity class when
// OS == 2) and every tap t reads the source voxel  q*IS + shift_t  (IS in {1,2}):
//     y[o][co] = sum_t sum_ci  x[q*IS + shift_t][ci] * W[widx_t][co][ci]
// Taps are grouped into UNITS = one dense source sub-brick + the taps that read it (an output parity class when OS == 2,
// a source parity sub-lattice when IS == 2), so all taps of a unit are wave-uniform shifts inside one LDS brick.
// A workgroup (4 waves) owns a BDxBHxBW brick of q and NT = 16*NS output channels.  Per 64-byte channel slab (32 bf16 /
// 16 f32 channels) the haloed source brick is staged ONCE into LDS (branch-free coalesced 16-byte buffer loads; out-of-range
// voxels and voxels of inactive patches are the hardware's zero fill), then every tap reads its shifted window from LDS
// (the 3x3x3 stencil reuse) and contracts channels on the matrix cores:
//     D(16 cout x 16 voxel) += A(weight fragments) * B(voxel fragments), both ds_read_b128.
//   * source brick: 80-byte rows, so a fragment address is (lane constant) + (scalar tap offset): zero VALU per read.  These
//     reads are 2-way bank-conflicted; the conflict-free alternatives were measured slower (profiles/r01_conv_ablation.md).
//   * weights: XOR-swizzled 64-byte rows (conflict-free), in double-buffered groups of 3 taps shared by the 4 waves: the next
//     group's global loads fly while the current group's MFMAs issue (per-wave weight loads from L1/L2 would need
//     128 B/clk/CU -- twice what the vector memory path delivers).
//   * slabs are software-pipelined: slab k+1 and its first weight group are loaded into registers during slab k's MFMAs.
// Packed weights are zero-padded to whole tiles, so the inner loop has no bounds logic at all.
// Accumulators stay in registers across all slabs, taps and units; the epilogue adds the bias, optionally an eval-mode
// BatchNorm (scale/shift), a skip tensor and an activation, applies the output patch mask, writes 8 consecutive channels
// (16 bytes) per lane and (optionally) leaves per-workgroup per-channel partial sums (sum, sum of squares) for the norm
// that follows -- no extra pass over y.
#include <mutex>
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "../../include/anatomask_hip.h"
#include "conv_plan.h"

using namespace amconv;

namespace {

// NB: the norm-backward-reduce variant (am_conv3d_nbred, bf16): the tile of the norm's input that the epilogue needs is fetched at
// the START of the workgroup (32 more registers) -- loaded in the epilogue its HBM latency is exposed once per workgroup, which
// for a one-slab data gradient (3.5 us of MFMAs) ate everything the fused reduce saves.
// HT: the unit's LAST weight group holds exactly TG / 2 taps (k3 s1 with the 32-channel output tile: 27 taps = 4 groups of 6 + 3) and runs
// a cluster of TG / 2 taps -- compile time, no branch inside the MFMA cluster: the padded form issued 30 tap slots for 27 taps.
template <typename T, int BD, int BH, int BW, int NS, int NIT, int TGS = 3, bool HR = false, bool NB = false, bool HT = false>
__global__ __launch_bounds__(256, (TGS == 2 && BD * BH * BW <= 256) ? 3 : 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int EPC = TT<T>::EPC;
  // SPL (T = f32s_t, AM_DT_F32S): fp32 in memory, but the staged LDS row of a voxel's 16 channels is [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15]
  // in bf16 (the split happens once per staged element, here; the weights arrive pre-split from am_pack_weight) and a chunk MMA is two
  // bf16 matrix instructions on two B fragments (common.h mma_split) -- 4x the rate of the exact fp32 mode
  constexpr bool SPL = std::is_same<T, f32s_t>::value;
  constexpr int KC = (ROWB / 16) * EPC;                 // channels per slab
  constexpr int MV = BD * BH * BW;
  constexpr int VS = MV / 64;                           // 16-voxel subtiles per wave
  constexpr int NT = 16 * NS;                           // output channels per workgroup
  constexpr int TG = TG_OF(NS, TGS);
  constexpr int WBUF = TG * NT * ROWB;                  // bytes of one weight-group buffer
  constexpr int WCH = TG * NT * 4;                      // 16-byte chunks per weight group
  constexpr int WIT = (WCH + 255) / 256;
  static_assert(MV % 64 == 0, "brick must give each wave whole 16-voxel subtiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ldsW = lds + a.w_lds_off;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  // Workgroups that read the SAME source brick -- the output-channel tiles of a brick, and for two-class-stride plans (OS == 2:
  // transposed conv, strided data gradient) its 8 output-parity classes -- live next to each other in blockIdx.x:
  //   linear id = ((brick / 8) * nsib + sibling) * 8 + brick % 8,   sibling = ytile * nclass + class,
  // so a brick's siblings have ids that differ by multiples of 8 (the same XCD under round-robin dispatch -- speed only) and are
  // dispatched back to back: the source is fetched into that L2 once.  (As grid.y / grid.z they ran tile-major / class-major: the whole
  // grid of class 0, then class 1, ... -- rocprofv3 counted 4.96 GB fetched by the ConvT 64->64 @128^3 launch for a 0.54 GB source and
  // 3.2 GB by 128->128 @64^3, two channel tiles, for 1.07 GB.)
  int cls = 0, ytile = 0, bid = blockIdx.x;
  {
    const int nsib = a.nclass * a.ny;
    if (nsib > 1) {
      const int t = bid >> 3, sib = t % nsib;
      cls = sib % a.nclass; ytile = sib / a.nclass;
      bid = ((t / nsib) << 3) + (bid & 7);
      if (bid >= (a.plist ? a.nlive : a.B * a.nbd * a.nbh * a.nbw)) return;    // (grid padded to whole groups of 8 bricks)
    }
  }
  const int brick = bid;
  int bw_, bh_, bd_, b;
  if (a.plist) {
    // live bricks only: (active patch, brick inside it) -- a launch over every brick of the q grid spends a dispatch slot and a mask
    // round trip on each empty one (60 % of them at mask 0.6, times 8 for the strided data gradient's classes)
    const int bpp = a.pbd * a.pbh * a.pbw;
    const int pk = a.plist[bid / bpp], j = bid % bpp;
    b = (pk >> 24) & 255;
    bd_ = ((pk >> 16) & 255) * a.pbd + j / (a.pbh * a.pbw); bh_ = ((pk >> 8) & 255) * a.pbh + (j / a.pbw) % a.pbh; bw_ = (pk & 255) * a.pbw + j % a.pbw;
  } else {
    bw_ = bid % a.nbw; bid /= a.nbw;
    bh_ = bid % a.nbh; bid /= a.nbh;
    bd_ = bid % a.nbd; b = bid / a.nbd;
  }
  const int q0d = bd_ * BD, q0h = bh_ * BH, q0w = bw_ * BW;
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  const int co0 = ytile * NT;
  float* part = a.partials ? a.partials + ((size_t)((size_t)cls * (a.plist ? a.nlive : a.B * a.nbd * a.nbh * a.nbw) + brick) * a.Cout) * 2 : nullptr;
#ifdef AM_ABLATE
  // timing experiment (tools build): delay every second set of 256 workgroups of the FIRST dispatch round, so that the two
  // workgroups that share a CU do not run their prologue / main loop / epilogue phases in lockstep (AM_CV_DBG bits 4096.., n x s_sleep 127)
  if ((a.dbg >> 12) & 15) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (((lin >> 8) & 1) && lin < 512) for (int i = 0; i < ((a.dbg >> 12) & 15); ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // ---- skip bricks with no active output voxel (block-sparse outputs) ----
  if (a.plist) {
    // (listed bricks are live)
  } else if (a.out_mask.m && a.brick_in_patch) {         // (uniform) every voxel of the brick shares the patch of its first voxel
    const int od = q0d * a.OS + pd, oh = q0h * a.OS + ph, ow = q0w * a.OS + pw;
    if (!(od < a.Do && oh < a.Ho && ow < a.Wo && a.out_mask.active(b, od, oh, ow))) {
      if (part && tid < NT && co0 + tid < a.Cout) { part[(co0 + tid) * 2] = 0.f; part[(co0 + tid) * 2 + 1] = 0.f; }
      return;
    }
  } else if (a.out_mask.m) {
    int any = 0;
    for (int v = tid; v < MV; v += 256) {
      const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
      if (od < a.Do && oh < a.Ho && ow < a.Wo && a.out_mask.active(b, od, oh, ow)) any = 1;
    }
    if (!__syncthreads_or(any)) {
      if (part && tid < NT && co0 + tid < a.Cout) { part[(co0 + tid) * 2] = 0.f; part[(co0 + tid) * 2 + 1] = 0.f; }
      return;
    }
  }

  const int u0 = a.OS == 2 ? cls : 0, u1 = a.OS == 2 ? cls + 1 : a.nunit;
  // an output parity class without taps (k1 s2 data gradient: 7 of 8 classes) contributes zeros: when the launch accumulates into
  // y there is nothing to do -- do not read-modify-write 7/8 of the tensor
  if (a.OS == 2 && a.accumulate && !part && a.tap_begin[cls + 1] == a.tap_begin[cls]) return;

  // tap table -> one VGPR (lane t = tap t of the plan, <= 64 taps): byte offset of the tap's window inside its unit's brick | widx << 20.
  // Taps are fetched with v_readlane (no LDS round trip in front of every weight load / fragment read).
  int tapv;
  {
    const int tp = a.taps[lane];
    const int ud = (tp & 15) - 8, uh = ((tp >> 4) & 15) - 8, uw = ((tp >> 8) & 15) - 8, un = (tp >> 18) & 7;
    tapv = ((((ud - a.mind[un]) * a.eh[un] + (uh - a.minh[un])) * a.ew[un] + (uw - a.minw[un])) * LROWB) | (((tp >> 12) & 63) << 20);
  }
#define AM_TAP(T_) __builtin_amdgcn_readlane(tapv, tb + ((T_) < nt ? (T_) : nt - 1))

  // Buffer descriptors (wave-uniform): loads take a per-lane 32-bit byte offset + an SGPR offset, so the slab / tap
  // advance costs no vector instruction, and any offset >= num_records reads back as zero -- the hardware does the
  // zero-fill of halo voxels that are out of range or belong to inactive patches (sentinel offset 0x80000000).
  // The source descriptor is anchored at the first d-plane this workgroup can touch (not at the sample start), so the 32-bit
  // offsets only span the brick's few planes: tensors of any size work (STUNet-H 192^3 x 192 ch is 2.7 GB per sample).
  const size_t plane_elems = (size_t)a.Hi * a.Wi * a.Cin;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int cchunk = (tid & 3) * EPC;                    // this thread's channel offset inside the slab
  const int sdst = (tid >> 2) * LROWB + (tid & 3) * (SPL ? 8 : 16);  // LDS byte offset of iteration 0; iteration it adds it*64*LROWB (immediate)
  const int wtapB = a.Coutp * a.Cinp * (int)sizeof(T);   // bytes per weight tap slice

  // ---- per-thread staging plan for the weight groups: chunk idx -> (tap in group [wave-uniform], cout row, chunk) ----
  unsigned wsrc[WIT];
  int wdst[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int idx = tid + it * 256;
    const int row = (idx >> 2) % NT, tig = idx / (NT * 4);
    wsrc[it] = (unsigned)(((co0 + crow(row)) * a.Cinp + (idx & 3) * EPC) * (int)sizeof(T));
    wdst[it] = tig * NT * ROWB + swz(row, idx & 3);
  }
  int aoff[NS];                                          // swizzled LDS offset of this lane's weight row chunk (tap 0 of a group)
#pragma unroll
  for (int i = 0; i < NS; ++i) aoff[i] = swz(i * 16 + r16, g);
  f32x4 acc[NS][VS];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // issue the loads of weight group GI of slab KCW into WR; the tap of a chunk is wave-uniform -> scalar offset.
  // taps past the end of the unit (partial last group) and !OK requests load zeros through the out-of-range rule.
#define AM_WLOAD(WR, GI, KCW, OK)                                                                          \
  _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                                      \
    const int tt_ = __builtin_amdgcn_readfirstlane((GI) * TG + (tid + it * 256) / (NT * 4));                \
    const int wi_ = AM_TAP(tt_) >> 20;                                                                      \
    WR[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tt_ < nt && (OK) && !AM_DBG(a, 4)) ? wsrc[it] : OOB, wi_ * wtapB + (KCW) * (int)sizeof(T), 0)); \
  }
  // (UNCONDITIONAL stores: a lane-masked store puts the wait for its load inside a branch a wave may skip, and the compiler then
  // drains `vmcnt` at the head of the group loop for the path it cannot rule out)
  static_assert(WCH % 256 == 0, "every thread stages exactly WIT chunks of a weight group");
#define AM_WSTORE(WR, BUF)                                                                                 \
  _Pragma("unroll") for (int it = 0; it < WIT; ++it)                                                        \
    *(u32x4*)(ldsW + (BUF) * WBUF + wdst[it]) = WR[it];
  // issue the loads of the source brick's channel slab KCS into stg (only a partial last slab, Cin % KC != 0, has !cok lanes)
#define AM_SLOAD(KCS, OK)                                                                                  \
  {                                                                                                        \
    const bool cok_ = (KCS) + cchunk < a.Cin && (OK) && !AM_DBG(a, 2);                                       \
    _Pragma("unroll") for (int it = 0; it < NIT; ++it)                                                      \
      stg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, cok_ ? soff[it] : OOB, (KCS) * (int)sizeof(T), 0)); \
  }

  u32x4 nbx[NB ? VS : 1][NB ? NS / 2 : 1];
  if constexpr (NB) {
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      const int v = wave * (MV / 4) + j * 16 + r16;
      const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
      const bool inr = od < a.Do && oh < a.Ho && ow < a.Wo;
      const size_t ovox = ((size_t)(b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
#pragma unroll
      for (int h = 0; h < NS / 2; ++h) {
        const bool ok = inr && co0 + h * 32 + g * 8 < a.Cout;
        nbx[j][h] = *(const u32x4*)((const T*)a.nb_x + (ok ? ovox * a.Cout + co0 + g * 8 + h * 32 : (size_t)0));   // (!ok: any valid address; unused)
      }
    }
  }
  for (int un = u0; un < u1; ++un) {
    const int tb = a.tap_begin[un], nt = a.tap_begin[un + 1] - tb;
    if (nt == 0) continue;                               // (k1 s2 dgrad parities without taps write zeros)
    const int ng = (nt + TG - 1) / TG;
    const int ED = a.ed[un], EH = a.eh[un], EW = a.ew[un];
    const int nvox = ED * EH * EW;
    const int upd = (a.upar[un] >> 2) & 1, uph = (a.upar[un] >> 1) & 1, upw = a.upar[un] & 1;
    const int i0d = q0d + a.mind[un], i0h = q0h + a.minh[un], i0w = q0w + a.minw[un];   // brick origin in sub-lattice voxels
    int dbase = i0d * a.GS + upd; dbase = dbase < 0 ? 0 : (dbase > a.Di ? a.Di : dbase);
    const size_t left = (size_t)(a.Di - dbase) * plane_elems * sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)a.x + ((size_t)b * a.Di + dbase) * plane_elems), 0, (int)(left < 0x7fffff00ull ? left : 0x7fffff00ull), 0x00020000);

    // the first weight group's loads go out BEFORE the staging plan is computed (they do not depend on it): their latency runs under
    // the plan's ~300 vector instructions instead of behind them
    u32x4 stg[NIT], wr[WIT];
    AM_WLOAD(wr, 0, 0, true);

    // ---- per-thread staging plan for this unit's source brick: byte offset of this thread's chunk in each of its rows ----
    unsigned soff[NIT];
    const int mW = a.mdiv_w[un], mHW = a.mdiv_hw[un], EHW = EH * EW;
    uint8_t mb[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                   // branch-free: offsets + (block-sparse input) all patch-mask bytes in flight at once
      const int e = (tid + it * 256) >> 2;
      const int ez = (e * mHW) >> 20, rem = e - ez * EHW;         // runtime extents: reciprocal multiply instead of
      const int ey = (rem * mW) >> 20, ex = rem - ey * EW;         // ~35-instruction integer divisions
      const int id = (i0d + ez) * a.GS + upd, ih = (i0h + ey) * a.GS + uph, iw = (i0w + ex) * a.GS + upw;
      const bool ok = e < nvox && (unsigned)id < (unsigned)a.Di && (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
      soff[it] = ok ? (unsigned)(((((id - dbase) * a.Hi + ih) * a.Wi + iw) * a.Cin + cchunk) * (int)sizeof(T)) : OOB;
      mb[it] = a.in_mask.m ? a.in_mask.peek(b, id, ih, iw, ok) : (uint8_t)1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (!mb[it]) soff[it] = OOB;
    int bb[VS];                                          // LDS byte offset of this lane's voxel-row chunk, tap shift excluded
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      const int v = wave * (MV / 4) + j * 16 + r16;
      bb[j] = (((v / (BW * BH)) * EH + (v / BW) % BH) * EW + v % BW) * LROWB + (SPL ? (g & 1) : g) * 16;
    }

    // Software pipeline over the channel slabs: the source brick of slab k+1 and the first weight group of slab k+1 are
    // loaded into registers while the MFMAs of slab k issue (HBM latency of every slab but the first is hidden), and are
    // written to LDS behind the barrier that ends slab k.
    AM_SLOAD(0, true);
    int bufp = 0;                                        // LDS weight buffer of the CURRENT group (toggles every group, across slabs)
    for (int kc = 0; kc < a.Cinp; kc += KC) {
      // (every group ends with a barrier: all fragment reads of the previous slab / unit are done)
      if (kc == 0) {
        __syncthreads();
        AM_WSTORE(wr, bufp);                             // later slabs: their group 0 was stored by the previous slab's last group
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (((tid + it * 256) >> 2) < nvox) {
          if constexpr (SPL) {                           // this thread's 4 channels: hi pair at +8 c, lo pair at +32 + 8 c of the row
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
            unsigned h2[2], l2[2];
            split4_bf16(stg[it], h2, l2);
            *(u32x2_*)(lds + sdst + it * 64 * LROWB) = u32x2_{h2[0], h2[1]};
            *(u32x2_*)(lds + sdst + it * 64 * LROWB + 32) = u32x2_{l2[0], l2[1]};
          } else {
            *(u32x4*)(lds + sdst + it * 64 * LROWB) = stg[it];
          }
        }
      __syncthreads();
      const bool more_slabs = kc + KC < a.Cinp;
      // One weight group = TG taps: straight-line (padding taps multiply zero weights), no per-tap branch, so the fragment reads of
      // tap t+1 can be scheduled under the MFMAs of tap t
      auto mma_group = [&](const int gi, const int buf, auto half_tag) __attribute__((always_inline)) {
        constexpr int NTP = decltype(half_tag)::value ? TG / 2 : TG;      // taps of this cluster
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);                     // the MFMA cluster of a weight group issues ahead of the other wave's staging / address work (+1.5-3.5 %)
        if constexpr (HR) {
          // h-runs: taps 3r, 3r+1, 3r+2 of the group differ only by one h-row of the source brick, and subtile j IS h-row j of the
          // wave's d-plane (4x4x16 brick): fragment row j+th serves (subtile j, tap th) -- 6 row reads per run instead of 12
          static_assert(TG % 3 == 0 && NTP % 3 == 0 && BD == 4 && BH == 4 && BW == 16 && VS == 4, "h-run reuse needs the 4x4x16 brick and 3-tap runs");
          const int ewb = EW * LROWB;
#pragma unroll
          for (int tr = 0; tr < NTP / 3; ++tr) {
            const int tob = AM_TAP(gi * TG + tr * 3) & 0xFFFFF;
            u32x4 brow[VS + 2], brow2[SPL ? VS + 2 : 1];
#pragma unroll
            for (int r = 0; r < VS + 2; ++r) {
              brow[r] = *(const u32x4*)(lds + bb[0] + tob + r * ewb);
              if constexpr (SPL) brow2[r] = *(const u32x4*)(lds + bb[0] + tob + r * ewb + 32);
            }
#pragma unroll
            for (int th = 0; th < 3; ++th) {
              u32x4 af[NS];
#pragma unroll
              for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(ldsW + buf * WBUF + (tr * 3 + ((AM_DBG(a, 64) && th == 1) ? 0 : th)) * NT * ROWB + aoff[i]);
#pragma unroll
              for (int j = 0; j < VS; ++j)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                  if constexpr (SPL) acc[i][j] = mma_split(af[i], brow[j + th], brow2[j + th], acc[i][j]);
                  else acc[i][j] = mma_chunk<T>(af[i], brow[j + th], acc[i][j]);
                }
            }
          }
        } else {
#pragma unroll
          for (int tl = 0; tl < NTP; ++tl) {
            const int tt = gi * TG + tl;
            const int tob = AM_TAP(tt) & 0xFFFFF;
            u32x4 af[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(ldsW + buf * WBUF + tl * NT * ROWB + aoff[i]);
#pragma unroll
            for (int j = 0; j < VS; ++j) {
              // (16-wide bricks with one d-plane per wave: subtile j is h-row j, so its offset is bb[0] + j rows -- no register per subtile)
              const int bo_ = ((BW == 16 && MV / 4 == BH * BW) ? bb[0] + j * (EW * LROWB) : bb[j]) + tob;
              const u32x4 bf = *(const u32x4*)(lds + bo_);
              if constexpr (SPL) {
                const u32x4 bf2 = *(const u32x4*)(lds + bo_ + 32);
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i][j] = mma_split(af[i], bf, bf2, acc[i][j]);
              } else {
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], bf, acc[i][j]);
              }
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      };
      // Load pipeline of a slab.  Every group is the same straight line: issue the NEXT group's weight loads (group 0 of the next
      // slab after the last group; out of range -- zeros, no traffic -- past the last slab), run this group's MFMAs, wait for the
      // weights, store them into the other LDS buffer, barrier.  There is no path from an un-waited load back to the loop head (with
      // `if (more)` around the store, or lane-masked stores, the compiler saw one and drained `vmcnt` at the head: on entry that wait
      // retired the next slab's source loads issued a moment earlier, before the slab's first MFMA -- their whole latency exposed once
      // per slab and workgroup).  `vmcnt` retires loads IN ORDER, so the place of the slab loads in the issue order decides how long
      // they may fly: group 0 is peeled and issues them AFTER group 1's weight loads -- the counted wait for those weights at the end
      // of group 0 (vmcnt(NIT + 2) ...) leaves them outstanding, and the wait at the end of group 1 retires them: two groups of MFMAs
      // (96-192 of them) of cover.
      {
        const bool more = ng > 1;
        AM_WLOAD(wr, more ? 1 : 0, more ? kc : kc + KC, more || more_slabs);
        AM_SLOAD(kc + KC, more_slabs);                     // branch-free: past the last slab the loads are out-of-range (zeros, no traffic)
        mma_group(0, bufp, std::false_type{});
        if (!AM_DBG(a, 32)) { AM_WSTORE(wr, bufp ^ 1); }
        if (!AM_DBG(a, 16)) __syncthreads();
        bufp ^= 1;
      }
      if constexpr (HT) {                                  // (host: one unit, ng >= 2, the last group holds TG / 2 taps)
        for (int gi = 1; gi + 1 < ng; ++gi) {
          AM_WLOAD(wr, gi + 1, kc, true);
          mma_group(gi, bufp, std::false_type{});
          if (!AM_DBG(a, 32)) { AM_WSTORE(wr, bufp ^ 1); }
          if (!AM_DBG(a, 16)) __syncthreads();
          bufp ^= 1;
        }
        AM_WLOAD(wr, 0, kc + KC, more_slabs);
        mma_group(ng - 1, bufp, std::true_type{});
        if (!AM_DBG(a, 32)) { AM_WSTORE(wr, bufp ^ 1); }
        if (!AM_DBG(a, 16)) __syncthreads();
        bufp ^= 1;
      } else
      for (int gi = 1; gi < ng; ++gi) {
        const bool more = gi + 1 < ng;
        AM_WLOAD(wr, more ? gi + 1 : 0, more ? kc : kc + KC, more || more_slabs);
        mma_group(gi, bufp, std::false_type{});
        if (!AM_DBG(a, 32)) { AM_WSTORE(wr, bufp ^ 1); }
        if (!AM_DBG(a, 16)) __syncthreads();
        bufp ^= 1;
      }
    }
  }
#undef AM_SLOAD
#undef AM_TAP
#undef AM_WLOAD
#undef AM_WSTORE

  // ---- epilogue: D row 4g+r of tile i = channel crow(16i+4g+r), col = voxel r16 ----
  T* __restrict__ yg = (T*)a.y;
  constexpr int NH = NS / 2;                             // 8-channel runs per lane and voxel
  f32x4 bia[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int co = co0 + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
    bia[i] = (a.bias && co < a.Cout) ? *(const f32x4*)(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool fused = a.ep_scale != nullptr || a.ep_res != nullptr || a.ep_act != AM_ACT_NONE;   // (uniform) eval-mode norm / residual / activation
  const T* __restrict__ resg = (const T*)a.ep_res;
  // the lane's per-channel scale / shift are fetched ONCE, like the bias (as conditional loads inside the store loop they were 32
  // dependent round trips per lane: the teacher's eval-BatchNorm decoder convs ran 17 % slower than the student's)
  f32x4 esc[NS], esh[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int co = co0 + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
    const bool okc = a.ep_scale && co < a.Cout;
    esc[i] = okc ? *(const f32x4*)(a.ep_scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
    esh[i] = okc ? *(const f32x4*)(a.ep_shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float act_slope = a.ep_act == AM_ACT_LRELU ? 0.01f : (a.ep_act == AM_ACT_RELU6 ? 0.f : 1.f);
  const float act_hi = a.ep_act == AM_ACT_RELU6 ? 6.f : __builtin_inff();
  // 8 consecutive channels of a tensor shaped like y, as two f32x4: ONE 16-byte load for bf16 (element-wise 2-byte loads before)
  auto load8 = [&](const T* p, f32x4& lo, f32x4& hi) {
    if constexpr (sizeof(T) == 4) { lo = *(const f32x4*)p; hi = *(const f32x4*)(p + 4); }
    else {
      typedef __attribute__((ext_vector_type(8))) __bf16 bfx8_;
      typedef __attribute__((ext_vector_type(8))) float f32x8_;
      const f32x8_ f = __builtin_convertvector(*(const bfx8_*)p, f32x8_);
      lo = f32x4{f[0], f[1], f[2], f[3]}; hi = f32x4{f[4], f[5], f[6], f[7]};
    }
  };
  auto ep = [&](f32x4 o, int i, f32x4 radd) {
    if (a.ep_scale) o = o * esc[i] + esh[i];
    o += radd;
    // act(v) = min(max(v, slope * v), hi) with (slope, hi) = (1, inf) none / (0.01, inf) LeakyReLU / (0, 6) ReLU6: four instructions
    // per value on uniform constants (a per-element `if (act == ...)` chain compiled to ~8 and cost the teacher's decoder convs 9 %)
    if (a.ep_act != AM_ACT_NONE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = fmaxf(o[r], o[r] * act_slope); o[r] = t > act_hi ? act_hi : t; }   // (a NaN stays a NaN, as in torch)
    }
    return o;
  };
  const bool sparse_out = a.out_mask.m != nullptr;
  // bf16 statistics run on the matrix cores (below): the stored values are also laid out as a [voxel][channel] tile in LDS
  // row stride of that tile: 160 B (conflict-free for ds_read_b64_tr_b16, conv_wgrad.hip); the norm-backward reduce keeps a second
  // tile (the norm's input) in the same rows, [g | x | pad] = 288 B (72 dwords: 8 consecutive rows still cover all 64 banks) -- two
  // separate 160-byte tiles would be 80 KB, exactly half a CU's LDS, and cost the second resident workgroup
  constexpr int SRS = NB ? 4 * NT + 32 : 2 * NT + 32;
  const bool mstats = sizeof(T) == 2 && part && !AM_DBG(a, 512);
  if (mstats) __syncthreads();                           // (uniform) the main loop's fragment reads are done: the staging area is free
#pragma unroll
  for (int j = 0; j < VS; ++j) {
    const int v = wave * (MV / 4) + j * 16 + r16;
    const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
    const bool inr = od < a.Do && oh < a.Ho && ow < a.Wo;
    const bool act = inr && (!sparse_out || a.out_mask.active(b, od, oh, ow));
    const size_t ovox = ((size_t)(b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
    T* dstv = yg + ovox * a.Cout + co0 + g * 8;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 o0 = acc[2 * h][j] + bia[2 * h], o1 = acc[2 * h + 1][j] + bia[2 * h + 1];
      T* dst = dstv + h * 32;
      const bool wr = inr && co0 + h * 32 + g * 8 < a.Cout && !AM_DBG(a, 1);     // Cout % 8 == 0 (C % 8 == 0 contract)
      if (fused) {
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if (resg && act && wr) load8(resg + (dst - yg), r0, r1);
        o0 = ep(o0, 2 * h, r0); o1 = ep(o1, 2 * h + 1, r1);
      }
      if (sparse_out && !act) { o0 = f32x4{0.f, 0.f, 0.f, 0.f}; o1 = o0; }
      if (a.accumulate && act && wr) {
        f32x4 p0_, p1_;
        load8(dst, p0_, p1_);
        o0 += p0_; o1 += p1_;
      }
      if constexpr (sizeof(T) == 4) {
        if (wr) { *(f32x4*)dst = o0; *(f32x4*)(dst + 4) = o1; }
        acc[2 * h][j] = act && wr ? o0 : f32x4{0.f, 0.f, 0.f, 0.f};   // what was stored (for the statistics pass below)
        acc[2 * h + 1][j] = act && wr ? o1 : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
        typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
        const bfx4 p0 = __builtin_convertvector(o0, bfx4), p1 = __builtin_convertvector(o1, bfx4);   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
        if (wr) { if (a.nt_store) __builtin_nontemporal_store(__builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7), (bfx8*)dst); else *(bfx8*)dst = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7); }
        if (mstats) {
          const bfx8 z8 = __builtin_bit_cast(bfx8, u32x4{0u, 0u, 0u, 0u});
          bfx8 gk = act && wr ? __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7) : z8;
          if constexpr (NB) {
            // norm-backward reduce: second tile = the norm's input at the same voxels (exact copy); first tile = g = dy * act'(x*sc + sh)
            const bfx8 xv = act && wr ? __builtin_bit_cast(bfx8, nbx[j][h]) : z8;
            if (a.nb_act != AM_ACT_NONE) {
              const float nb_slope = a.nb_act == AM_ACT_LRELU ? 0.01f : 0.f, nb_hi = a.nb_act == AM_ACT_RELU6 ? 6.f : __builtin_inff();
              const int co = co0 + h * 32 + g * 8;
              const f32x4 sc0 = *(const f32x4*)(a.nb_scale + co), sc1 = *(const f32x4*)(a.nb_scale + co + 4);
              const f32x4 sh0 = *(const f32x4*)(a.nb_shift + co), sh1 = *(const f32x4*)(a.nb_shift + co + 4);
              typedef __attribute__((ext_vector_type(8))) float f32x8;
              const f32x8 xf = __builtin_convertvector(xv, f32x8);
              f32x8 gf = __builtin_convertvector(gk, f32x8);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float z = xf[e] * (e < 4 ? sc0[e & 3] : sc1[e & 3]) + (e < 4 ? sh0[e & 3] : sh1[e & 3]);
                gf[e] *= (z > 0.f && z < nb_hi) ? 1.f : nb_slope;
              }
              gk = __builtin_convertvector(gf, bfx8);
            }
            *(bfx8*)(lds + v * SRS + 2 * NT + (h * 32 + g * 8) * 2) = xv;
          }
          *(bfx8*)(lds + v * SRS + (h * 32 + g * 8) * 2) = gk;
        }
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    // Per-workgroup per-channel partials (sum, sum of squares of the STORED bf16 values; no atomics, deterministic) on the matrix
    // cores: with the tile in LDS as [voxel][channel], transposing reads give fragments F[channel][voxel] -- exactly the weight-
    // gradient contraction -- and   F x ones = sums,   diag(F x F^T) = sums of squares.  Wave w owns channel tile w: MV/32 k-steps
    // of 2 reads + 2 MFMAs.  (Before: 128 FMAs + 128 DPP adds per lane, 6.5 % of the dominant launch.)
    if (mstats) {
      __syncthreads();
      typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
      const bfx8 ones = __builtin_bit_cast(bfx8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
      const int q = (lane >> 2) & 3, p = lane & 3;
      for (int i = wave; i < NS; i += 4) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MV / 32; ++ks) {
          const int v1 = ks * 32 + g * 4 + q;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + v1 * SRS + (16 * i + 4 * p) * 2));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (v1 + 16) * SRS + (16 * i + 4 * p) * 2));
          const bfx8 f = __builtin_bit_cast(bfx8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
          bfx8 f2 = f;
          if constexpr (NB) {                            // norm-backward reduce: diag(G x X^T) instead of diag(F x F^T)
            const s16x4 xlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + v1 * SRS + 2 * NT + (16 * i + 4 * p) * 2));
            const s16x4 xhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (v1 + 16) * SRS + 2 * NT + (16 * i + 4 * p) * 2));
            f2 = __builtin_bit_cast(bfx8, s16x8{xlo[0], xlo[1], xlo[2], xlo[3], xhi[0], xhi[1], xhi[2], xhi[3]});
          }
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, ones, s1, 0, 0, 0);
          s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f2, s2, 0, 0, 0);
        }
        // D row 4g+r = channel, col r16: the diagonal element of channel c = 16i + r16 sits in the lane with r16 >> 2 == g, at r = r16 & 3
        if ((r16 >> 2) == g && co0 + 16 * i + r16 < a.Cout) {
          const int r = r16 & 3;
          const float v1s = r == 0 ? s1[0] : r == 1 ? s1[1] : r == 2 ? s1[2] : s1[3];
          const float v2s = r == 0 ? s2[0] : r == 1 ? s2[1] : r == 2 ? s2[2] : s2[3];
          part[(co0 + 16 * i + r16) * 2] = v1s; part[(co0 + 16 * i + r16) * 2 + 1] = v2s;
        }
      }
    }
  } else
  if (part && !AM_DBG(a, 512)) {                         // f32: per-workgroup per-channel partials on the vector ALU (no atomics, deterministic)
    if (!AM_DBG(a, 1024)) __syncthreads();
    float* red = (float*)lds;                            // [4 waves][16*NS couts][2]
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;                        // sums over this lane's VS voxels of the STORED values
#pragma unroll
        for (int j = 0; j < VS; ++j) { const float o = acc[i][j][r]; s1 += o; s2 += o * o; }
        if (!AM_DBG(a, 2048)) { s1 = row16_sum(s1); s2 = row16_sum(s2); }
        if (r16 == 0) {
          const int c = (i >> 1) * 32 + g * 8 + (i & 1) * 4 + r;
          red[(wave * 16 * NS + c) * 2] = s1; red[(wave * 16 * NS + c) * 2 + 1] = s2;
        }
      }
    if (!AM_DBG(a, 1024)) __syncthreads();
    if (tid < 16 * NS && co0 + tid < a.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += red[(w * 16 * NS + tid) * 2]; s2 += red[(w * 16 * NS + tid) * 2 + 1]; }
      part[(co0 + tid) * 2] = s1; part[(co0 + tid) * 2 + 1] = s2;
    }
  }
}

template <typename T, int BD, int BH, int BW, int NS, int NIT, int TGS = 3, bool HR = false, bool NB = false, bool HT = false>
int launch(Plan& P, hipStream_t st) {
  ConvArgs& a = P.a;
  auto kern = conv_igemm_kernel<T, BD, BH, BW, NS, NIT, TGS, HR, NB, HT>;
  static PerDeviceOnce lds_cap;                   // per instantiation and device: lift the 48 KB dynamic-LDS default to the CU's 160 KB
  lds_cap.run([&](int) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); });
  if (P.lds > 160 * 1024) return -3;
  const unsigned nbrick = a.plist ? (unsigned)a.nlive : (unsigned)(a.B * a.nbd * a.nbh * a.nbw);
  a.ny = (a.Cout + 16 * NS - 1) / (16 * NS);
  const unsigned nsib = (unsigned)(a.ny * a.nclass);
  dim3 grid(nsib > 1 ? ((nbrick + 7) / 8) * 8 * nsib : nbrick, 1, 1);       // (channel tiles and parity classes folded into x, see the kernel)
  AM_LAUNCH(kern, grid, dim3(256), P.lds, st, a);
  AM_CHECK_LAUNCH();
  return 0;
}

template <typename T, int NS>
int dispatch_nit(Plan& P, int shape, hipStream_t st) {
  const int n = P.nit;
  if (P.a.nb_x) {                                // norm-backward-reduce variants: the k3 s1 plans (one unit), bf16
    if constexpr (sizeof(T) == 2) {
      if (P.tgs != 3 || P.a.OS != 1) return -7;
      if (shape == 2) return n <= 4 ? launch<T, 4, 4, 4, NS, 4, 3, false, true>(P, st) : -7;
      if (shape == 1) return n > 7 && n <= 11 ? (P.a.hreuse ? launch<T, 4, 4, 16, NS, 11, 3, true, true>(P, st) : launch<T, 4, 4, 16, NS, 11, 3, false, true>(P, st)) : -7;
      return n > 7 && n <= 11 ? launch<T, 4, 8, 8, NS, 11, 3, false, true>(P, st) : -7;
    } else return -7;
  }
  if (P.tgs == 2) {                              // multi-unit plans: sub-bricks of at most (BD+1)(BH+1)(BW+1) voxels
    if (shape == 3) {                                                                  // 512-voxel bricks (dense transposed convs, bf16)
      if constexpr (sizeof(T) == 2) return n <= 12 ? launch<T, 4, 8, 16, NS, 12, 2>(P, st) : -3;
      else return -3;
    }
    if (shape == 2) return n <= 4 ? launch<T, 4, 4, 4, NS, 4, 2>(P, st) : -3;
    if (shape == 1) return n <= 4 ? launch<T, 4, 4, 16, NS, 4, 2>(P, st) : n <= 7 ? launch<T, 4, 4, 16, NS, 7, 2>(P, st) : -3;
    return n <= 4 ? launch<T, 4, 8, 8, NS, 4, 2>(P, st) : n <= 7 ? launch<T, 4, 8, 8, NS, 7, 2>(P, st) : -3;
  }
  if (shape == 2) {                              // 4x4x4 brick: tiny grids only (more workgroups)
    if (n <= 4) return launch<T, 4, 4, 4, NS, 4>(P, st);
    return -3;
  }
  // 32-channel output tile (TG = 6) on a 27-tap unit: 4 full groups + a half group (kernel variant HT), bf16
  const bool ht = NS == 2 && sizeof(T) == 2 && P.a.nunit == 1 && P.a.tap_begin[1] == 27;
  if constexpr (NS == 2 && sizeof(T) == 2) {
    if (ht && shape == 1 && n > 7 && n <= 11)
      return P.a.hreuse ? launch<T, 4, 4, 16, NS, 11, 3, true, false, true>(P, st) : launch<T, 4, 4, 16, NS, 11, 3, false, false, true>(P, st);
    if (ht && shape == 0 && n > 7 && n <= 11) return launch<T, 4, 8, 8, NS, 11, 3, false, false, true>(P, st);
  }
  if (shape == 1) {                              // 4x4x16: every 16-lane fragment is 16 consecutive voxels (conflict-free reads)
    if (n <= 4) return launch<T, 4, 4, 16, NS, 4>(P, st);
    if (n <= 7) return launch<T, 4, 4, 16, NS, 7>(P, st);
#ifdef AM_ABLATE
    if (n <= 11 && getenv("AM_CV_NOHR")) return launch<T, 4, 4, 16, NS, 11>(P, st);
#endif
    if (n <= 11) return P.a.hreuse ? launch<T, 4, 4, 16, NS, 11, 3, true>(P, st) : launch<T, 4, 4, 16, NS, 11>(P, st);
    return -3;
  }
  if (n <= 4) return launch<T, 4, 8, 8, NS, 4>(P, st);          // narrow grids (W < 16)
  if (n <= 7) return launch<T, 4, 8, 8, NS, 7>(P, st);
  if (n <= 11) return launch<T, 4, 8, 8, NS, 11>(P, st);
  return -3;
}

template <typename T>
int dispatch(Plan& P, int shape, hipStream_t st) {
  return P.nt_tile == 32 ? dispatch_nit<T, 2>(P, shape, st) : dispatch_nit<T, 4>(P, shape, st);
}

}  // namespace

// brick of q-space voxels per workgroup: 0 = 4x8x8 (narrow grids), 1 = 4x4x16, 2 = 4x4x4 (tiny grids, chosen by the caller).
// Block-sparse outputs whose patches are 8 q-voxels wide take the 4x8x8 brick: it lies inside ONE patch, so 60 % of the bricks
// are skipped outright (a 4x4x16 brick spans two patches and is empty only 36 % of the time).
static int brick_shape(int os, int qh, int qw, bool out_sparse, int out_bshift, int* bd, int* bh, int* bw) {
  *bd = 4;
  const int qblock = out_sparse ? ((1 << out_bshift) / os) : 0;
  // two-class-stride plans (transposed convs) on grids whose width is not a multiple of 16 (STUNet-L 160^3: q = 20; STUNet-H: 24, 12):
  // the brick that pads the (h, w) plane less (ConvT 512->512 @40^3 627 -> 834 TFLOP/s).  The k3 s1 plans stay 16-wide whatever the
  // padding: their h-run fragment reuse is worth more than the zero-operand MFMAs of the padding cost (512->512 @40^3: 1 094 vs 1 013)
  const long pad16 = (long)((qh + 3) / 4 * 4) * ((qw + 15) / 16 * 16), pad8 = (long)((qh + 7) / 8 * 8) * ((qw + 7) / 8 * 8);
  if (qw >= 16 && qblock != 8 && (os != 2 || pad16 * 100 <= pad8 * 108)) { *bh = 4; *bw = 16; return 1; }
  *bh = 8; *bw = 8; return 0;
}

// Tiling of a launch: brick shape + output-channel tile.  Small grids (deep levels: 8^3..16^3 voxels, 256-512 channels) first halve
// the channel tile (2x the workgroups, each still 4 subtiles x 2 channel tiles per wave = 24 MFMAs per weight group); only if
// that still leaves most CUs idle do they drop to 64-voxel bricks (6 MFMAs per barrier: measured 151 us for the 512->512 conv
// at 8^3 where the 256-voxel brick x 32 channels takes a third of that).
static int pick_tiling(int os, int B, int qd, int qh, int qw, int Cout, bool out_sparse, int out_bshift, int* bd, int* bh, int* bw, int* nt, bool big_ok = true) {
  int shape = brick_shape(os, qh, qw, out_sparse, out_bshift, bd, bh, bw);
  *nt = Cout <= 32 ? 32 : 64;
  const long q = (long)qd * qh * qw;
  // (bricks counted per dimension: a 12^3 grid is 3 x 2 x 2 = 12 bricks of 4x8x8, not ceil(1728 / 256) = 7)
  auto nwg = [&](int tile) { return (long)B * ((qd + *bd - 1) / *bd) * ((qh + *bh - 1) / *bh) * ((qw + *bw - 1) / *bw) * ((Cout + tile - 1) / tile) * (os == 2 ? 8 : 1); };
  if (nwg(*nt) < 256) {
    *nt = 32;
    if (nwg(32) < 192 && shape != 2) { shape = 2; *bh = 4; *bw = 4; }
  }
  // dense two-class-stride launches (ConvTranspose3d: 8 output parity classes of 8 taps) with plenty of workgroups: 4x8x16 bricks.
  // A 4x4x16 workgroup of such a plan has only 128 MFMAs per wave and slab between its prologue and its 32 KB store (64->64
  // @128^3: 731 TFLOP/s); twice the voxels halve the staging per MFMA (5x9x17 rows for 512 voxels: 1.49x, against 1.66x) and
  // every weight fragment feeds 8 MFMAs instead of 4.
  if (os == 2 && !out_sparse && shape == 1 && *nt == 64 && qh >= 8 && nwg(64) >= 4096 && big_ok) { shape = 3; *bh = 8; }
  return shape;
}

extern "C" int am_packed_dims(int dtype, int rows, int k, int* rows_padded, int* k_padded) {
  const int tile = rows <= 32 ? 32 : 64;
  const int kc = dtype == AM_DT_BF16 ? 32 : 16;
  *rows_padded = (rows + tile - 1) / tile * tile;
  *k_padded = (k + kc - 1) / kc * kc;
  return 0;
}

extern "C" int am_conv3d_partials_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int out_sparse,
                                       int out_bshift, int n_active, int* rows) {
  const int os = (mode == AM_CONV_DGRAD) ? stride : (mode == AM_CONVT_FWD ? 2 : 1);
  const int qd = (Do + os - 1) / os, qh = (Ho + os - 1) / os, qw = (Wo + os - 1) / os;
  int bd, bh, bw, nt;
  pick_tiling(os, B, qd, qh, qw, Cout, out_sparse != 0, out_bshift, &bd, &bh, &bw, &nt, dtype == AM_DT_BF16);   // (512-voxel bricks: bf16 only, as in conv3d_impl)
  *rows = B * ((qd + bd - 1) / bd) * ((qh + bh - 1) / bh) * ((qw + bw - 1) / bw) * (os == 2 ? 8 : 1);
  // upper bound over the kernels a launch of this shape may take (am_conv3d reports the rows it actually wrote)
  const int rw = conv_rw_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_sparse, out_bshift, n_active);
  if (rw > *rows) *rows = rw;
  const int rg = conv_gather_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_sparse, out_bshift, n_active);
  if (rg > *rows) *rows = rg;
  const int rk = conv_k3_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_sparse);
  if (rk > *rows) *rows = rk;
  return 0;
}

static int conv3d_impl(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed,
                       const float* bias, void* y, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo,
                       int Cout, const uint8_t* in_mask, int in_bshift, const uint8_t* out_mask, int out_bshift,
                       int fd, int fh, int fw, int accumulate, float* partials, const float* ep_scale, const float* ep_shift,
                       const void* ep_res, int ep_act, const int32_t* active_list, int n_active, int* partial_rows_written,
                       const void* nb_x, const float* nb_scale, const float* nb_shift, int nb_act, void* stream,
                       const float* in_scale = nullptr, const float* in_shift = nullptr, int in_act = 0) {
  if (Cin % 8 || Cout % 8) return -1;
  if (ep_scale && !ep_shift) return -1;
  Plan P;
  ConvArgs& a = P.a;
  a.stats_sum_only = (accumulate & AM_CONV_PARTIALS_SUM_ONLY) ? 1 : 0;      // (a hint: every kernel but conv_k3's fills both columns anyway)
  accumulate &= 1;
  a.nb_x = nb_x; a.nb_scale = nb_scale; a.nb_shift = nb_shift; a.nb_act = nb_act;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  // ---- thin layers with everything resident in LDS: conv_rw.hip ----
  {
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.partials = partials;
    a.ep_scale = ep_scale; a.ep_shift = ep_scale ? ep_shift : nullptr; a.ep_res = ep_res; a.ep_act = ep_act;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    am_packed_dims(dtype, Cout, Cin, &a.Coutp, &a.Cinp);
    a.w_bytes = ksize * ksize * ksize * a.Coutp * a.Cinp * (dtype == AM_DT_BF16 ? 2 : 4);
    a.in_mask = MaskView{in_mask, fd, fh, fw, in_bshift};
    a.out_mask = MaskView{out_mask, fd, fh, fw, out_bshift};
    a.accumulate = accumulate; a.nt_store = 0; a.brick_in_patch = 0; a.hreuse = 0;
#ifdef AM_ABLATE
    { const char* e = getenv("AM_CV_DBG"); a.dbg = e ? atoi(e) : 0; }
    if (getenv("AM_CV_NORW")) goto generic;
#endif
    a.plist = nullptr;
    if (!in_scale && mode == AM_CONVT_FWD && !partials) {   // dense transposed convs at decoder sizes: the persistent kernel's transposed instantiation
      const int rt = conv_k3t_launch(mode, dtype, ksize, stride, a, stream);
      if (rt < 0) return rt;
      if (rt == 1) return 0;
    }
    if (!in_scale) {                               // dense k3 s1 at decoder sizes: the persistent LDS-DMA kernel (conv_k3.hip)
      const int rk = conv_k3_launch(mode, dtype, ksize, stride, a, stream);
      if (rk < 0) return rk;
      if (rk == 1) {
        if (partial_rows_written) *partial_rows_written = conv_k3_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, 0);
        return 0;
      }
    }
    const int rc = nb_x ? 0 : conv_rw_launch(mode, dtype, ksize, stride, a, active_list, n_active, stream);   // (the fused reduce lives in conv_igemm_kernel only)
    if (rc < 0) return rc;
    if (rc == 1) {
      if (partial_rows_written) *partial_rows_written = conv_rw_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_mask != nullptr, out_bshift, n_active);
      return 0;
    }
    if (in_scale) return -7;                       // (the fused input norm lives in conv_rw.hip only: am_conv3d_prenorm_supported says when)
    const int rg = conv_gather_launch(mode, dtype, ksize, stride, a, active_list, n_active, stream);   // deep block-sparse levels: rows = active voxels
    if (rg < 0) return rg;
    if (rg == 1) {
      if (partial_rows_written) *partial_rows_written = conv_gather_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_mask != nullptr, out_bshift, n_active);
      return 0;
    }
  }
#ifdef AM_ABLATE
generic:
#endif
  const int os_ = (mode == AM_CONV_DGRAD) ? stride : (mode == AM_CONVT_FWD ? 2 : 1);
  bool big_ok = dtype == AM_DT_BF16;             // 512-voxel transposed-conv bricks: bf16 only (the fp32 instantiation needs 258 VGPRs and spills)
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_CV_NOBIG"); if (e_ && atoi(e_)) big_ok = false; }
#endif
  int shape = pick_tiling(os_, B, (Do + os_ - 1) / os_, (Ho + os_ - 1) / os_, (Wo + os_ - 1) / os_, Cout, out_mask != nullptr, out_bshift,
                          &P.bd, &P.bh, &P.bw, &P.nt_tile, big_ok);
  int rc = build_plan(P, mode, ksize, stride);
  if (rc) return rc;
  // k1 s2 data gradient that ACCUMULATES into y: only output parity (0, 0, 0) has a tap, the other 7 classes add nothing -- launch class 0
  // alone (as siblings in blockIdx.x the 7 idle classes sat between the live workgroups and the launch ran at the dispatcher's pace:
  // 64 -> 32 @128^3 took 0.48 ms for 0.1 ms of work)
  if (a.OS == 2 && accumulate && !partials && a.tap_begin[1] > a.tap_begin[0] && a.tap_begin[8] == a.tap_begin[1]) a.nclass = 1;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.partials = partials;
  a.ep_scale = ep_scale; a.ep_shift = ep_scale ? ep_shift : nullptr; a.ep_res = ep_res; a.ep_act = ep_act;
  a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  am_packed_dims(dtype, Cout, Cin, &a.Coutp, &a.Cinp);
  a.w_bytes = ksize * ksize * ksize * a.Coutp * a.Cinp * (dtype == AM_DT_BF16 ? 2 : 4);
  if ((size_t)24 * Hi * Wi * Cin * 4 >= 0x7fffff00ull) return -5;   // the <= 24 source planes a brick spans must stay below 2 GB (32-bit offsets)
  const int Qd = (Do + a.OS - 1) / a.OS, Qh = (Ho + a.OS - 1) / a.OS, Qw = (Wo + a.OS - 1) / a.OS;
  a.nbd = (Qd + P.bd - 1) / P.bd; a.nbh = (Qh + P.bh - 1) / P.bh; a.nbw = (Qw + P.bw - 1) / P.bw;
  a.in_mask = MaskView{in_mask, fd, fh, fw, in_bshift};
  a.out_mask = MaskView{out_mask, fd, fh, fw, out_bshift};
  a.accumulate = accumulate;
  a.plist = nullptr; a.pbd = a.pbh = a.pbw = a.nlive = 0;
  {
    const int qblock = out_mask ? ((1 << out_bshift) / a.OS) : 0;
    a.brick_in_patch = qblock > 0 && qblock % P.bd == 0 && qblock % P.bh == 0 && qblock % P.bw == 0;
    // (the list is am_mask_compact of this launch's OUT mask; whole patches inside the grid)
    bool listed = a.brick_in_patch && active_list && n_active > 0 && fd <= 255 && fh <= 255 && fw <= 255 && B <= 255 &&
                  Qd == fd * qblock && Qh == fh * qblock && Qw == fw * qblock;
#ifdef AM_ABLATE
    { const char* e = getenv("AM_CV_NOLIST"); if (e && atoi(e)) listed = false; }
#endif
    if (listed) {
      a.plist = active_list; a.pbd = qblock / P.bd; a.pbh = qblock / P.bh; a.pbw = qblock / P.bw;
      a.nlive = n_active * a.pbd * a.pbh * a.pbw;
    }
  }
#ifdef AM_ABLATE
  { const char* e = getenv("AM_CV_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  if (partials && dtype == AM_DT_BF16) {                 // the bf16 statistics epilogue lays the output tile out in LDS: [voxel][NT channels + 32 B]
    const size_t tile = (size_t)P.bd * P.bh * P.bw * (P.nt_tile * 2 + 32);
    if (P.lds < tile) P.lds = tile;                      // (only when asked for: 512-voxel bricks would lose their second workgroup per CU to it)
  }
  if (nb_x) {                                            // two [voxel][channel] tiles in the epilogue (g and the norm's input)
    const size_t tiles = (size_t)P.bd * P.bh * P.bw * (P.nt_tile * 4 + 32);
    if (P.lds < tiles) P.lds = tiles;
  }
  // outputs far larger than the 256 MB Infinity Cache bypass it (measured +2.5 % on the 1 GB decoder tensors: the halo re-reads keep L2)
  a.nt_store = ((size_t)B * Do * Ho * Wo * Cout * 2 >= ((size_t)384 << 20) && !accumulate) || AM_DBG(a, 8);
  hipStream_t st = (hipStream_t)stream;
  if (partial_rows_written) *partial_rows_written = (a.plist ? a.nlive : a.B * a.nbd * a.nbh * a.nbw) * a.nclass;
  return dtype == AM_DT_BF16 ? dispatch<bf16_t>(P, shape, st) : dtype == AM_DT_F32S ? dispatch<f32s_t>(P, shape, st) : dispatch<float>(P, shape, st);
}

extern "C" int am_conv3d(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed,
                         const float* bias, void* y, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo,
                         int Cout, const uint8_t* in_mask, int in_bshift, const uint8_t* out_mask, int out_bshift,
                         int fd, int fh, int fw, int accumulate, float* partials, const float* ep_scale, const float* ep_shift,
                         const void* ep_res, int ep_act, const int32_t* active_list, int n_active, int* partial_rows_written,
                         void* stream) {
  return conv3d_impl(mode, dtype, ksize, stride, x, w_packed, bias, y, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, in_mask, in_bshift, out_mask,
                     out_bshift, fd, fh, fw, accumulate, partials, ep_scale, ep_shift, ep_res, ep_act, active_list, n_active,
                     partial_rows_written, nullptr, nullptr, nullptr, 0, stream);
}

// Forward convolution of act(x * in_scale + in_shift) -- a (pooled sparse Instance)Norm + activation folded into the consumer's source
// staging, so that the normalised map is never written or read (the EMA teacher's stage-0 conv2 and every encoder pass that keeps no
// tape: P/STUNet_head.py:96-103 `y = LReLU(IN(conv1 x)); conv2(y)`).  Inactive / out-of-volume source rows stay zero, as the dense-conv-
// then-mask semantics of P/encoder3D.py:12-15 has them.  Served by conv_rw.hip only (bf16, Cin <= 32, k3): am_conv3d_prenorm_supported.
extern "C" int am_conv3d_prenorm_supported(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout,
                                           int sparse, int in_bshift, int out_bshift, int n_active) {
  if (mode != AM_CONV_FWD || !sparse || n_active <= 0 || in_bshift != out_bshift + (stride == 2 ? 1 : 0)) return 0;
  return conv_rw_rows(mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, sparse, out_bshift, n_active) > 0 ? 1 : 0;
}

extern "C" int am_conv3d_prenorm(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed, const float* bias, void* y,
                                 int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, const uint8_t* mask, int in_bshift,
                                 int out_bshift, int fd, int fh, int fw, float* partials, const float* in_scale, const float* in_shift,
                                 int in_act, const int32_t* active_list, int n_active, int* partial_rows_written, void* stream) {
  if (!in_scale || !in_shift || !mask || !active_list) return -1;
  if (in_act != AM_ACT_NONE && in_act != AM_ACT_LRELU && in_act != AM_ACT_RELU6) return -1;
  return conv3d_impl(mode, dtype, ksize, stride, x, w_packed, bias, y, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, mask, in_bshift, mask, out_bshift,
                     fd, fh, fw, 0, partials, nullptr, nullptr, nullptr, AM_ACT_NONE, active_list, n_active, partial_rows_written,
                     nullptr, nullptr, nullptr, 0, stream, in_scale, in_shift, in_act);
}

// A data-gradient (or any) convolution whose output y is the gradient wrt a = act(nb_x * nb_scale + nb_shift), the output of a
// norm + activation: the launch also leaves, per workgroup and channel, (sum g, sum g * nb_x) with g = y * act'(.) in `partials`
// -- what am_norm_bwd_reduce would compute in a pass of its own over y and nb_x (am_norm_bwd_from_partials turns the rows into
// the k0 / k1 / k2 coefficients and the affine gradients).  bf16 only; nb_x has y's shape and layout.
extern "C" int am_conv3d_nbred(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed, void* y, int B, int Di,
                               int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, const uint8_t* in_mask, int in_bshift,
                               const uint8_t* out_mask, int out_bshift, int fd, int fh, int fw, int accumulate, float* partials,
                               const void* nb_x, const float* nb_scale, const float* nb_shift, int nb_act,
                               int* partial_rows_written, void* stream) {
  if (dtype != AM_DT_BF16 || !partials || !nb_x) return -1;
  if (nb_act != AM_ACT_NONE && (!nb_scale || !nb_shift)) return -1;
  if (nb_act != AM_ACT_NONE && nb_act != AM_ACT_LRELU && nb_act != AM_ACT_RELU6) return -1;
  return conv3d_impl(mode, dtype, ksize, stride, x, w_packed, nullptr, y, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, in_mask, in_bshift, out_mask,
                     out_bshift, fd, fh, fw, accumulate, partials, nullptr, nullptr, nullptr, AM_ACT_NONE, nullptr, 0,
                     partial_rows_written, nb_x, nb_scale, nb_shift, nb_act, stream);
}
