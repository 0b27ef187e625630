// Resident-weight gather convolution for THIN layers (gfx950, bf16): Cin <= 32 (one 64-byte channel slab per voxel) and
// taps * Cout small enough that EVERY tap's weights fit in LDS next to the source brick:
//   STUNet stage 0  conv2 32->32 k3 s1 @128^3 (block-sparse, 16^3 patches)  and its data gradient      (P/STUNet_head.py:86, P/encoder3D.py:12-15)
//   STUNet stage 1  conv1 32->64 k3 s2 -> 64^3 (block-sparse, 8^3 output patches)                       (P/STUNet_head.py:81)
// The generic kernel (conv_igemm.hip) gives such a layer one workgroup per brick: a prologue (tap table, staging plan, buffer
// descriptors), ONE channel slab, and nine 3-tap weight groups that each cost a barrier and a trip to L2 -- ~200 us of weight
// traffic and ~120 us of fixed costs around 84 us of MFMA work (profiles/r01_encoder_fwd.md).  Here a PERSISTENT workgroup
//   * loads all taps' weights into LDS once (27 x 32 x 64 B = 54 KB, or 27 x 64 x 64 B = 108 KB),
//   * walks a contiguous run of ACTIVE bricks taken from the active-patch list (am_mask_compact) -- empty bricks are never
//     launched, the activity of a brick's 27 neighbour patches is one ballot per brick (fetched two bricks ahead), so the
//     staging plan of a source row is pure ALU,
//   * software-pipelines the stages (brick x unit): the next stage's source rows are in flight (registers) while the current
//     stage's taps issue; per stage there are two barriers and no weight traffic at all,
//   * keeps the per-channel (sum, sum of squares) of what it stored in registers across its bricks and leaves ONE partials row.
// Fragment layouts, LDS images, tap plan and epilogue channel order are those of conv_igemm.hip (conv_plan.h).
#include <mutex>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/anatomask_hip.h"
#include "conv_plan.h"

using namespace amconv;

namespace {

struct RwArgs {
  const int* plist;      // active-patch list (block-sparse in/out) or nullptr (dense: every brick of the q grid)
  int nbrick;            // bricks to process in total
  int pbd, pbh, pbw;     // bricks per patch along d, h, w   (block-sparse)
  int pq;                // patch edge in q (= output) voxels (block-sparse)
  int ntaps;
};

typedef bf16_t T;
constexpr int EPC = 8;
constexpr unsigned OOB = 0x80000000u;

template <int NW, int BD, int BH, int BW, int NS, int NIT, bool HR, int LR>
__global__ __launch_bounds__(NW * 64) void conv_rw_kernel(ConvArgs a, RwArgs r) {
  constexpr int NTH = NW * 64;
  constexpr int MV = BD * BH * BW;
  constexpr int VS = MV / (16 * NW);                     // 16-voxel subtiles per wave
  constexpr int NT = 16 * NS;
  constexpr int RPI = NTH / 4;                           // source rows staged per iteration
  static_assert(MV % (16 * NW) == 0, "brick must give each wave whole 16-voxel subtiles");
  static_assert(!HR || (BH == 4 && BW == 16 && VS == 4), "h-run reuse: a wave owns one 4x16 d-plane");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ldsW = lds + a.w_lds_off;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int co0 = blockIdx.y * NT;
  const bool sparse = r.plist != nullptr;
  const int ibs = a.in_mask.bs, obs = a.out_mask.bs;

  // ---- this workgroup's contiguous run of bricks ----
  const int chunk = (r.nbrick + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * chunk, b1 = min(b0 + chunk, r.nbrick);

  // ---- all taps' weights -> LDS (once) ----
  {
    const int wtapB = a.Coutp * a.Cinp * (int)sizeof(T);
    const int total = r.ntaps * NT * 4;
    for (int idx = tid; idx < total; idx += NTH) {
      const int tap = idx / (NT * 4), rem = idx - tap * (NT * 4), row = rem >> 2, ch = rem & 3;
      const int widx = (a.taps[tap] >> 12) & 63;
      const u32x4 v = *(const u32x4*)((const unsigned char*)a.w + (size_t)widx * wtapB + ((co0 + crow(row)) * a.Cinp + ch * EPC) * (int)sizeof(T));
      *(u32x4*)(ldsW + tap * NT * ROWB + swz(row, ch)) = v;
    }
  }
  // Strided (8-unit) block-sparse launches stage EVERY unit with the same (BD+1) x (BH+1) x (BW+1) sub-brick (a unit whose parity
  // needs no halo along an axis gets one unused row there: +25 % staged rows), so that the staging plan of a row is the same for
  // all units and can be precomputed like the single-unit one.
  const bool uni = !HR && sparse && a.GS == 2 && a.nunit == 8 && ibs == 4;
  // tap table -> one VGPR (lane t = tap t): byte offset of the tap's window inside its unit's brick
  int tapv;
  {
    const int tp = a.taps[lane];
    const int ud = (tp & 15) - 8, uh = ((tp >> 4) & 15) - 8, uw = ((tp >> 8) & 15) - 8, un = (tp >> 18) & 7;
    const int eh_ = uni ? BH + 1 : a.eh[un], ew_ = uni ? BW + 1 : a.ew[un];
    tapv = (((ud - a.mind[un]) * eh_ + (uh - a.minh[un])) * ew_ + (uw - a.minw[un])) * LR;
  }
  // per-unit plan parameters -> lane tables (lane u = unit u): a v_readlane per use instead of a scalar load from the kernel
  // argument segment (indexed by the runtime unit number those are s_load + s_waitcnt chains, ~20 of them per stage)
  const int ul = lane & 7;
  const int u_eh = a.eh[ul], u_ew = a.ew[ul], u_ed = a.ed[ul], u_mdw = a.mdiv_w[ul], u_mdhw = a.mdiv_hw[ul], u_par = a.upar[ul];
  const int u_md = a.mind[ul], u_mh = a.minh[ul], u_mw = a.minw[ul], u_tb = a.tap_begin[ul], u_te = a.tap_begin[ul + 1];
#define UP(V_, UN_) __builtin_amdgcn_readlane(V_, UN_)
  int aoff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) aoff[i] = swz(i * 16 + r16, g);
  f32x4 acc[NS][VS];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1a[NS][4], s2a[NS][4];                          // statistics of the stored values, across all bricks of this workgroup
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) s1a[i][q] = s2a[i][q] = 0.f;
  f32x4 bia[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int co = co0 + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
    bia[i] = (a.bias && co < a.Cout) ? *(const f32x4*)(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int cchunk = (tid & 3) * EPC;
  const int sdst = (tid >> 2) * LR + (tid & 3) * 16;
  // fused input norm + activation (am_conv3d_prenorm): the thread's channel chunk never changes, its 8 (scale, shift) pairs live in registers
  const bool prenorm = a.in_scale != nullptr;
  float isc[EPC], ish[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) { isc[i] = (prenorm && cchunk + i < a.Cin) ? a.in_scale[cchunk + i] : 1.f; ish[i] = (prenorm && cchunk + i < a.Cin) ? a.in_shift[cchunk + i] : 0.f; }
  const float in_slope = a.in_act == AM_ACT_LRELU ? 0.01f : (a.in_act == AM_ACT_RELU6 ? 0.f : 1.f);
  const float in_hi = a.in_act == AM_ACT_RELU6 ? 6.f : __builtin_inff();
  const size_t plane_elems = (size_t)a.Hi * a.Wi * a.Cin;
  const bool cok = cchunk < a.Cin;
  T* __restrict__ yg = (T*)a.y;
  const int bpp = r.pbd * r.pbh * r.pbw;

  // ---- brick table of this workgroup (LDS, built once): sample, q origin, fmap patch, activity bits of the 27 neighbour patches.
  // Decoding a brick index takes runtime integer divisions, and the CU's single scalar unit executing them three times per stage for
  // eight waves was the largest cost of this kernel's skeleton (tools/rw_ablate.py); the stages only read the table.
  int* tbl = (int*)(ldsW + r.ntaps * NT * ROWB);
  for (int i = tid; i < b1 - b0; i += NTH) {
    const int bi = b0 + i;
    int b, q0d, q0h, q0w, pd = 0, ph = 0, pw = 0, nbm = 0;
    if (sparse) {
      const int ai = bi / bpp, s_ = bi - ai * bpp;
      const int pk = r.plist[ai];
      b = (pk >> 24) & 255; pd = (pk >> 16) & 255; ph = (pk >> 8) & 255; pw = pk & 255;
      const int sd = s_ / (r.pbh * r.pbw), sh = (s_ / r.pbw) % r.pbh, sw = s_ % r.pbw;
      q0d = pd * r.pq + sd * BD; q0h = ph * r.pq + sh * BH; q0w = pw * r.pq + sw * BW;
      for (int l = 0; l < 27; ++l) {
        const int nd = pd + l / 9 - 1, nh = ph + (l / 3) % 3 - 1, nw = pw + l % 3 - 1;
        const bool ok = (unsigned)nd < (unsigned)a.in_mask.fd && (unsigned)nh < (unsigned)a.in_mask.fh && (unsigned)nw < (unsigned)a.in_mask.fw;
        if (ok && a.in_mask.m[((b * a.in_mask.fd + nd) * a.in_mask.fh + nh) * a.in_mask.fw + nw]) nbm |= 1 << l;
      }
    } else {
      int t = bi;
      const int bw_ = t % a.nbw; t /= a.nbw;
      const int bh_ = t % a.nbh; t /= a.nbh;
      const int bd_ = t % a.nbd; b = t / a.nbd;
      q0d = bd_ * BD; q0h = bh_ * BH; q0w = bw_ * BW;
    }
    int* e = tbl + i * 8;
    e[0] = b; e[1] = q0d; e[2] = q0h; e[3] = q0w; e[4] = pd | (ph << 8) | (pw << 16); e[5] = nbm;
  }
  __syncthreads();
  // table entry of brick bi -> scalars
  auto fetch = [&](int bi, int& b, int& q0d, int& q0h, int& q0w, int& pd, int& ph, int& pw, int& nbm) {
    const int* e = tbl + (bi - b0) * 8;
    const int p = __builtin_amdgcn_readfirstlane(e[4]);
    b = __builtin_amdgcn_readfirstlane(e[0]); q0d = __builtin_amdgcn_readfirstlane(e[1]); q0h = __builtin_amdgcn_readfirstlane(e[2]);
    q0w = __builtin_amdgcn_readfirstlane(e[3]); nbm = __builtin_amdgcn_readfirstlane(e[5]);
    pd = p & 255; ph = (p >> 8) & 255; pw = (p >> 16) & 255;
  };

  // Source rows travel global -> registers -> LDS; TWO register sets alternate so that the rows of stage s+2 are requested while
  // stage s computes (one stage of cover -- a few hundred to 3 500 cycles -- does not hide an HBM round trip under load).
  u32x4 stgA[NIT], stgB[NIT];
  unsigned vmA = 0u, vmB = 0u;                           // rows of the register set that exist (bit it): the fused input transform leaves the others zero
  // staging plan of stage (brick bi, unit un) + issue of its loads into stg.  nbm: activity bits of the 27 neighbour patches.
  // `valid` = false (past the last stage): the same number of loads is issued, all out of range (zeros, no traffic) -- the
  // compiler counts vector-memory operations statically, and a CONDITIONAL prefetch makes it wait for everything in flight.
  // Single-unit (k3 s1) block-sparse launches: a row's position inside the haloed brick never changes, so its packed (d, h, w)
  // and its byte offset relative to the brick origin are computed ONCE per thread; per stage a row then costs one packed add, the
  // 3 x 3 x 3 neighbour-patch index (bit fields of that sum) and a bit test -- 11 vector instructions instead of ~35
  // (tools/rw_ablate.py: the staging-plan arithmetic was 100 us of this kernel's 205 us skeleton).
  int relf[NIT];
  unsigned roff[NIT];
  {
    const int EH = HR ? UP(u_eh, 0) : BH + 1, EW = HR ? UP(u_ew, 0) : BW + 1, ED = HR ? UP(u_ed, 0) : BD + 1;
    const int nvox = ED * EH * EW, EHW = EH * EW, gs = HR ? 1 : 2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = (tid >> 2) + it * RPI;
      const int ez = e / EHW, rem = e - ez * EHW, ey = rem / EW, ex = rem - ey * EW;       // (once per thread)
      relf[it] = (cok && e < nvox) ? ((ez * gs) | ((ey * gs) << 8) | ((ex * gs) << 16)) : (int)0x80000000;   // fine-grid steps
      roff[it] = (unsigned)((((ez * gs * a.Hi + ey * gs) * a.Wi + ex * gs) * a.Cin + cchunk) * (int)sizeof(T));
    }
  }
  auto plan_and_load = [&](u32x4 (&stg)[NIT], unsigned& vm, int bi_, int un_, bool valid) {
    vm = 0u;
    // everything that shapes the stage is wave-uniform: say so (scalar registers, scalar buffer descriptor)
    const int bi = __builtin_amdgcn_readfirstlane(valid ? bi_ : b0), un = __builtin_amdgcn_readfirstlane(valid ? un_ : 0);
    int b, q0d, q0h, q0w, pd, ph, pw, nbm;
    fetch(bi, b, q0d, q0h, q0w, pd, ph, pw, nbm);
    const int EH = UP(u_eh, un), EW = UP(u_ew, un), nvox = UP(u_ed, un) * EH * EW, EHW = EH * EW;
    const int mW = UP(u_mdw, un), mHW = UP(u_mdhw, un), par = UP(u_par, un);
    const int upd = (par >> 2) & 1, uph = (par >> 1) & 1, upw = par & 1;
    const int i0d = q0d + UP(u_md, un), i0h = q0h + UP(u_mh, un), i0w = q0w + UP(u_mw, un);
    int dbase = i0d * a.GS + upd; dbase = dbase < 0 ? 0 : (dbase > a.Di ? a.Di : dbase);
    const size_t left = (size_t)(a.Di - dbase) * plane_elems * sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)a.x + ((size_t)b * a.Di + dbase) * plane_elems), 0, (int)(left < 0x7fffff00ull ? left : 0x7fffff00ull), 0x00020000);
    {
      if ((HR && sparse && ibs == 4) || uni) {             // (uniform)
        const int P = 16;
        const int f0d = i0d * a.GS + upd, f0h = i0h * a.GS + uph, f0w = i0w * a.GS + upw;        // fine-grid origin of the (sub-)brick
        const unsigned sbase = (unsigned)(((((f0d - dbase) * a.Hi + f0h) * a.Wi + f0w) * a.Cin) * (int)sizeof(T));   // (wraps for halo origins; valid rows come out >= 0)
        const int sb = (f0d - pd * P + P) | ((f0h - ph * P + P) << 8) | ((f0w - pw * P + P) << 16);
        const bool live = valid & !AM_DBG(a, 2);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int t = relf[it] + sb;
          const int idx = ((t >> 4) & 3) * 9 + ((t >> 12) & 3) * 3 + ((t >> 20) & 3);
          const bool ok = live & (relf[it] >= 0) & (((nbm >> idx) & 1) != 0);
          vm |= ok ? (1u << it) : 0u;
          stg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? roff[it] + sbase : OOB, 0, 0));
        }
        return;
      }
    }
    if (AM_DBG(a, 16)) {                                  // (ablation: no staging-plan arithmetic)
#pragma unroll
      for (int it = 0; it < NIT; ++it) stg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, OOB, 0, 0));
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = (tid >> 2) + it * RPI;
      const int ez = (e * mHW) >> 20, rem = e - ez * EHW;
      const int ey = (rem * mW) >> 20, ex = rem - ey * EW;
      const int id = (i0d + ez) * a.GS + upd, ih = (i0h + ey) * a.GS + uph, iw = (i0w + ex) * a.GS + upw;
      // branch-free (bitwise &, not &&: the short-circuit form compiles to an exec-mask branch in front of every load)
      bool ok = valid & !AM_DBG(a, 2) & cok & (e < nvox) & ((unsigned)id < (unsigned)a.Di) & ((unsigned)ih < (unsigned)a.Hi) & ((unsigned)iw < (unsigned)a.Wi);
      if (sparse) {
        const int pidx = ((id >> ibs) - pd + 1) * 9 + ((ih >> ibs) - ph + 1) * 3 + ((iw >> ibs) - pw + 1);
        ok = ok & (((nbm >> (pidx & 31)) & 1) != 0);
      }
      const unsigned off = ok ? (unsigned)(((((id - dbase) * a.Hi + ih) * a.Wi + iw) * a.Cin + cchunk) * (int)sizeof(T)) : OOB;
      vm |= ok ? (1u << it) : 0u;
      stg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
  };

  if (b0 < b1) {
    // Pipeline state lives in plain locals and the stage body is a macro expanded once per register set: a lambda that mutates
    // captured state sends that state to scratch memory, and scratch traffic shares the vector-memory counter with the prefetches
    // (every scratch read would drain them).
    // stage cursor: (brick, unit), skipping units without taps
#define AM_RW_FIRST(UN_) while ((UN_) < a.nunit && UP(u_te, UN_) == UP(u_tb, UN_)) ++(UN_);
#define AM_RW_ADVANCE(BI_, UN_) { ++(UN_); AM_RW_FIRST(UN_) if ((UN_) >= a.nunit) { (UN_) = 0; AM_RW_FIRST(UN_) ++(BI_); } }
    int bi = b0, un = 0;
    AM_RW_FIRST(un)
    int bi1 = bi, un1 = un; AM_RW_ADVANCE(bi1, un1)     // stage s+1
    int bi2 = bi1, un2 = un1; AM_RW_ADVANCE(bi2, un2)   // stage s+2
    plan_and_load(stgA, vmA, bi, un, true);
    plan_and_load(stgB, vmB, bi1, un1, bi1 < b1);
#define AM_RW_STAGE(STG, VM) \
    { \
      bi = __builtin_amdgcn_readfirstlane(bi); un = __builtin_amdgcn_readfirstlane(un); \
      const int tb = UP(u_tb, un), nt = UP(u_te, un) - tb; \
      const int EH = uni ? BH + 1 : UP(u_eh, un), EW = uni ? BW + 1 : UP(u_ew, un), nvox = (uni ? BD + 1 : UP(u_ed, un)) * EH * EW; \
      if (prenorm) {                                      /* (uniform) x' -> act(x' * scale + shift) on the rows that exist */ \
_Pragma("unroll") \
        for (int it = 0; it < NIT; ++it) { \
          float f_[EPC]; \
          chunk_to_f<T>(STG[it], f_); \
_Pragma("unroll") \
          for (int e_ = 0; e_ < EPC; ++e_) { const float y_ = f_[e_] * isc[e_] + ish[e_]; const float t_ = fmaxf(y_, y_ * in_slope); f_[e_] = t_ > in_hi ? in_hi : t_; } \
          const u32x4 n_ = f_to_chunk<T>(f_); \
          if ((VM >> it) & 1u) STG[it] = n_; \
        } \
      } \
      __syncthreads(); \
_Pragma("unroll") \
      for (int it = 0; it < NIT; ++it) \
        if ((tid >> 2) + it * RPI < nvox && !AM_DBG(a, 8)) *(u32x4*)(lds + sdst + it * RPI * LR) = STG[it]; \
      __syncthreads(); \
      plan_and_load(STG, VM, bi2, un2, bi2 < b1); \
      int bb[VS]; \
_Pragma("unroll") \
      for (int j = 0; j < VS; ++j) { \
        const int v = wave * (MV / NW) + j * 16 + r16; \
        bb[j] = (((v / (BW * BH)) * EH + (v / BW) % BH) * EW + v % BW) * LR + g * 16; \
      } \
      __builtin_amdgcn_sched_barrier(0); \
      if (AM_DBG(a, 4)) { } else if constexpr (HR) { \
        const int ewb = EW * LR; \
_Pragma("unroll") \
        for (int tr = 0; tr < 9; ++tr) { \
          const int tob = __builtin_amdgcn_readlane(tapv, tr * 3); \
          u32x4 brow[VS + 2]; \
_Pragma("unroll") \
          for (int q = 0; q < VS + 2; ++q) brow[q] = *(const u32x4*)(lds + bb[0] + tob + q * ewb); \
_Pragma("unroll") \
          for (int th = 0; th < 3; ++th) { \
            u32x4 af[NS]; \
_Pragma("unroll") \
            for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(ldsW + (tr * 3 + th) * NT * ROWB + aoff[i]); \
_Pragma("unroll") \
            for (int j = 0; j < VS; ++j) \
_Pragma("unroll") \
              for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], brow[j + th], acc[i][j]); \
          } \
        } \
      } else { \
        auto taps = [&](auto NTAP_) { \
          constexpr int NTAP = decltype(NTAP_)::value; \
_Pragma("unroll") \
          for (int tl = 0; tl < NTAP; ++tl) { \
            const int tob = __builtin_amdgcn_readlane(tapv, tb + tl); \
            u32x4 af[NS]; \
_Pragma("unroll") \
            for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(ldsW + (tb + tl) * NT * ROWB + aoff[i]); \
_Pragma("unroll") \
            for (int j = 0; j < VS; ++j) { \
              const u32x4 bf = *(const u32x4*)(lds + bb[j] + tob); \
_Pragma("unroll") \
              for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], bf, acc[i][j]); \
            } \
          } \
        }; \
        if (nt == 8) taps(std::integral_constant<int, 8>{}); \
        else if (nt == 4) taps(std::integral_constant<int, 4>{}); \
        else if (nt == 2) taps(std::integral_constant<int, 2>{}); \
        else if (nt == 1) taps(std::integral_constant<int, 1>{}); \
        else { \
          for (int tl = 0; tl < nt; ++tl) { \
            const int tob = __builtin_amdgcn_readlane(tapv, tb + tl); \
            u32x4 af[NS]; \
_Pragma("unroll") \
            for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(ldsW + (tb + tl) * NT * ROWB + aoff[i]); \
_Pragma("unroll") \
            for (int j = 0; j < VS; ++j) { \
              const u32x4 bf = *(const u32x4*)(lds + bb[j] + tob); \
_Pragma("unroll") \
              for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], bf, acc[i][j]); \
            } \
          } \
        } \
      } \
      __builtin_amdgcn_sched_barrier(0); \
      if (bi1 != bi && !AM_DBG(a, 32)) { \
        int b, q0d, q0h, q0w, pd, ph, pw, nbm_; \
        fetch(bi, b, q0d, q0h, q0w, pd, ph, pw, nbm_); \
_Pragma("unroll") \
        for (int j = 0; j < VS; ++j) { \
          const int v = wave * (MV / NW) + j * 16 + r16; \
          const int od = q0d + v / (BW * BH), oh = q0h + (v / BW) % BH, ow = q0w + v % BW; \
          const bool inr = od < a.Do && oh < a.Ho && ow < a.Wo; \
          const size_t ovox = ((size_t)(b * a.Do + od) * a.Ho + oh) * a.Wo + ow; \
          T* dstv = yg + ovox * a.Cout + co0 + g * 8; \
_Pragma("unroll") \
          for (int h = 0; h < NS / 2; ++h) { \
            const f32x4 o0 = acc[2 * h][j] + bia[2 * h], o1 = acc[2 * h + 1][j] + bia[2 * h + 1]; \
            const bool wr = inr && co0 + h * 32 + g * 8 < a.Cout; \
            typedef __attribute__((ext_vector_type(4))) __bf16 bfx4; \
            typedef __attribute__((ext_vector_type(8))) __bf16 bfx8; \
            const bfx4 p0 = __builtin_convertvector(o0, bfx4), p1 = __builtin_convertvector(o1, bfx4); \
            if (wr && !AM_DBG(a, 1)) *(bfx8*)(dstv + h * 32) = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7); \
            const f32x4 q0 = __builtin_convertvector(p0, f32x4), q1 = __builtin_convertvector(p1, f32x4); \
_Pragma("unroll") \
            for (int q = 0; q < 4; ++q) { \
              if (wr) { s1a[2 * h][q] += q0[q]; s2a[2 * h][q] += q0[q] * q0[q]; s1a[2 * h + 1][q] += q1[q]; s2a[2 * h + 1][q] += q1[q] * q1[q]; } \
            } \
            acc[2 * h][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * h + 1][j] = f32x4{0.f, 0.f, 0.f, 0.f}; \
          } \
        } \
      } \
      bi = bi1; un = un1; bi1 = bi2; un1 = un2; \
      if (bi2 < b1) { AM_RW_ADVANCE(bi2, un2); } \
    } \

    while (bi < b1) {
      AM_RW_STAGE(stgA, vmA)
      if (bi >= b1) break;
      AM_RW_STAGE(stgB, vmB)
    }
#undef AM_RW_STAGE
#undef AM_RW_ADVANCE
#undef AM_RW_FIRST
#undef UP
  }
  // ---- ONE partials row per workgroup ----
  if (a.partials) {
    __syncthreads();
    float* red = (float*)lds;                            // [NW waves][16*NS couts][2]
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s1 = s1a[i][q], s2 = s2a[i][q];
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        if (r16 == 0) {
          const int c = (i >> 1) * 32 + g * 8 + (i & 1) * 4 + q;
          red[(wave * 16 * NS + c) * 2] = s1; red[(wave * 16 * NS + c) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < 16 * NS && co0 + tid < a.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { s1 += red[(w * 16 * NS + tid) * 2]; s2 += red[(w * 16 * NS + tid) * 2 + 1]; }
      float* part = a.partials + ((size_t)blockIdx.x * a.Cout + co0 + tid) * 2;
      part[0] = s1; part[1] = s2;
    }
  }
}

// which shapes qualify, and with what geometry
struct RwGeo { int nw, bd, bh, bw, nt_tile, nwg, nbrick, pbd, pbh, pbw, pq; bool hr; };

bool rw_geometry(RwGeo& G, int mode, int dtype, int k, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, bool sparse, int out_bshift,
                 int n_active) {
  if (dtype != AM_DT_BF16 || k != 3 || Cin > 32 || Cin % 8 || Cout > 64 || Cout % 8) return false;
  const bool s1 = stride == 1 && (mode == AM_CONV_FWD || mode == AM_CONV_DGRAD);
  const bool s2 = stride == 2 && mode == AM_CONV_FWD;
  if (!s1 && !s2) return false;
  G.nt_tile = Cout <= 32 ? 32 : 64;
  const int patch = sparse ? (1 << out_bshift) : 0;
  if (s1) {                                              // 27 taps, one unit: h-run layout, wave = one 4x16 d-plane
    if (G.nt_tile != 32) return false;                   // 27 x 64 x 64 B of weights + an 8x4x16 brick do not fit; 4 waves measured no gain
    if (sparse && patch != 16) return false;
    if (Wo % 16 || Ho % 4 || Do % 8) return false;
    G.nw = 8; G.bd = 8; G.bh = 4; G.bw = 16; G.hr = true;
  } else {                                               // 8 parity sub-lattices of the fine grid
    if (sparse && patch != 8) return false;
    if (Wo % 8 || Ho % 8 || Do % 4) return false;
    G.nw = 4; G.bd = 4; G.bh = 8; G.bw = 8; G.hr = false;
  }
  if (sparse) {
    if (n_active <= 0) return false;
    G.pq = patch; G.pbd = patch / G.bd; G.pbh = patch / G.bh; G.pbw = patch / G.bw;
    G.nbrick = n_active * G.pbd * G.pbh * G.pbw;
  } else {
    G.pq = 0; G.pbd = G.pbh = G.pbw = 1;
    G.nbrick = B * (Do / G.bd) * (Ho / G.bh) * (Wo / G.bw);
  }
  const int slices = (Cout + G.nt_tile - 1) / G.nt_tile;
  int nwg = 256 / slices; if (nwg < 1) nwg = 1;            // one persistent workgroup per CU (LDS: weights + brick > 80 KB)
  if (nwg > G.nbrick) nwg = G.nbrick;
  G.nwg = nwg;
  return G.nbrick > 0;
}

template <int NW, int BD, int BH, int BW, int NS, int NIT, bool HR, int LR>
int rw_launch(ConvArgs& a, RwArgs& r, const RwGeo& G, size_t lds, hipStream_t st) {
  auto kern = conv_rw_kernel<NW, BD, BH, BW, NS, NIT, HR, LR>;
  AM_LDS_OPTIN_STAGE(kern);
  dim3 grid(G.nwg, (a.Cout + 16 * NS - 1) / (16 * NS), 1);
  AM_LAUNCH(kern, grid, dim3(NW * 64), lds, st, a, r);
  AM_CHECK_LAUNCH_STAGE();
  return 1;
}

// EVERY acceptance condition of the resident-weight kernel that follows from the shape alone: the geometry, the tap plan (h-run order for
// the 8x4x16 brick), the LDS footprint (brick + all taps' weights + the workgroup's brick table, which grows with n_active: level 0 of
// STUNet-B 128^3 stops fitting above ~4864 active patches) and the staging iteration count.  conv_rw_rows and am_conv3d_prenorm_supported
// PREDICT with it what conv_rw_launch will do, so they cannot disagree with the launch.  P.a must hold the caller's arguments (or be default).
struct RwFit { size_t lds; int ntaps; };
constexpr int RW_LR = 96;                                // conflict-free AND additive row stride (the generic kernel's 80 B is 2-way conflicted: it must fit two workgroups per CU)

static int rw_fit(RwGeo& G, Plan& P, RwFit& F, int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout,
                  bool sparse, int out_bshift, int n_active) {
  if (!rw_geometry(G, mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, sparse, out_bshift, n_active)) return 0;
  P.bd = G.bd; P.bh = G.bh; P.bw = G.bw; P.nt_tile = G.nt_tile;
  const int rc = build_plan(P, mode, ksize, stride);
  if (rc) return rc;
  ConvArgs& a = P.a;
  if (G.hr && !a.hreuse) return 0;
  F.ntaps = a.tap_begin[a.nunit];
  size_t mxv = 0;
  for (int c = 0; c < a.nunit; ++c) { const size_t v = (size_t)a.ed[c] * a.eh[c] * a.ew[c]; if (v > mxv) mxv = v; }
  size_t brick = mxv * RW_LR;
  if (brick < 8192) brick = 8192;                        // the statistics fold reuses the head of the brick
  a.w_lds_off = (int)brick;
  const int chunk = (G.nbrick + G.nwg - 1) / G.nwg;      // bricks per workgroup -> 32-byte rows of its brick table
  F.lds = brick + (size_t)F.ntaps * G.nt_tile * ROWB + (size_t)chunk * 32;
  if (F.lds > 160 * 1024) return 0;
  const int nit = (int)((mxv * 4 + G.nw * 64 - 1) / (G.nw * 64));
  // 8x4x16 brick, haloed 10x6x18 = 1080 rows -> 9 staging iterations of 128 rows; 4x8x8 brick, sub-lattice sub-bricks of <= 5x9x9 = 405 rows -> 7 of 64
  if (nit > (G.hr ? 9 : 7)) return 0;
  return 1;
}

}  // namespace

namespace amconv {

int conv_rw_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int out_sparse, int out_bshift,
                 int n_active) {
  RwGeo G; Plan P; RwFit F;
  return rw_fit(G, P, F, mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_sparse != 0, out_bshift, n_active) == 1 ? G.nwg : 0;
}

int conv_rw_launch(int mode, int dtype, int ksize, int stride, ConvArgs& a0, const int* active_list, int n_active, void* stream) {
  const bool sparse = a0.out_mask.m != nullptr;
  if (sparse != (a0.in_mask.m != nullptr) || (sparse && a0.in_mask.m != a0.out_mask.m)) return 0;
  if (a0.accumulate || a0.ep_scale || a0.ep_res || a0.ep_act != AM_ACT_NONE) return 0;
  if (sparse && !active_list) return 0;
  if (a0.in_scale && (!a0.in_shift || mode != AM_CONV_FWD)) return 0;
  if (sparse) {                                          // the source halo must stay inside the 3x3x3 patch neighbourhood
    const int in_patch = 1 << a0.in_mask.bs;
    if (in_patch < 4 || (in_patch != (1 << a0.out_mask.bs) * (mode == AM_CONV_FWD ? stride : 1))) return 0;
  }
  RwGeo G; Plan P; RwFit F;
  P.a = a0;
  const int fit = rw_fit(G, P, F, mode, dtype, ksize, stride, a0.B, a0.Do, a0.Ho, a0.Wo, a0.Cin, a0.Cout, sparse, a0.out_mask.bs, n_active);
  if (fit != 1) return fit;
  ConvArgs& a = P.a;
  a.nbd = a.Do / G.bd; a.nbh = a.Ho / G.bh; a.nbw = a.Wo / G.bw;
  RwArgs r;
  r.plist = sparse ? active_list : nullptr; r.nbrick = G.nbrick; r.pbd = G.pbd; r.pbh = G.pbh; r.pbw = G.pbw; r.pq = G.pq; r.ntaps = F.ntaps;
  hipStream_t st = (hipStream_t)stream;
  if (G.hr) return rw_launch<8, 8, 4, 16, 2, 9, true, RW_LR>(a, r, G, F.lds, st);
  return G.nt_tile == 32 ? rw_launch<4, 4, 8, 8, 2, 7, false, RW_LR>(a, r, G, F.lds, st) : rw_launch<4, 4, 8, 8, 4, 7, false, RW_LR>(a, r, G, F.lds, st);
}

}  // namespace amconv
