// Gather implicit-GEMM 3-D convolution for gfx950 (MFMA), channels-last.
//
// One kernel serves every dense / block-sparse convolution on the AnatoMask path:
//   forward conv k1/k3 stride 1/2   (P/encoder3D.py:12-15 SparseConv3d, P/decoder3D.py:20-22, P/AnatoMask.py:63-65)
//   its data gradient               (autograd of the above)
//   ConvTranspose3d k4 s2 p1        (P/decoder3D.py:19) as 8 output-parity classes of 2x2x2 taps
//   its data gradient               (a k4 s2 p1 convolution of dy)
//
// Formulation.  Output voxels are enumerated as o = q*OS + p (OS in {1,2}; p = parity class when
// OS == 2) and every tap t of the class reads the source voxel  q*IS + shift_t  (IS in {1,2}):
//     y[o][co] = sum_t sum_ci  x[q*IS + shift_t][ci] * W[widx_t][co][ci]
// so all taps of a class are wave-uniform.  A workgroup (4 waves) owns a BDxBHxBW brick of q and
// NT = 16*NS output channels.  Per 64-byte channel slab (32 bf16 / 16 f32 channels) the haloed
// source brick is staged ONCE into LDS (coalesced 16-byte loads, zero for out-of-range voxels and
// for voxels of inactive 16^3-patches), then every tap reads its shifted window from LDS
// (the 3x3x3 stencil reuse) and contracts channels on the matrix cores:
//     D(16 cout x 16 voxel) += A(weights, 16-byte row fragments straight from L1/L2) * B(LDS fragments).
// Accumulators stay in registers across all slabs and taps; the epilogue adds the bias, applies the
// output patch mask and writes 4 consecutive channels per lane.
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

struct ConvArgs {
  const void* x; const void* w; const float* bias; void* y;
  int B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout;
  int OS, IS, nclass;
  int nbd, nbh, nbw;          // bricks per dim of the q grid
  int Qd, Qh, Qw;             // q grid extent per class
  int tap_begin[9];
  int taps[64];               // (sd+8) | (sh+8)<<4 | (sw+8)<<8 | widx<<12
  int mind[8], minh[8], minw[8];
  int ed[8], eh[8], ew[8];    // LDS source-brick extents per class
  MaskView in_mask, out_mask;
  int accumulate;
};

constexpr int ROWB = 64;      // channel-slab bytes staged per voxel
constexpr int LROW = 80;      // LDS row stride (16 B pad: spreads 16-lane b128 reads over banks)
constexpr int MAXIT = 16;

template <typename T, int BD, int BH, int BW, int NS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int EPC = TT<T>::EPC;
  constexpr int KC = (ROWB / 16) * EPC;                 // channels per slab
  constexpr int MV = BD * BH * BW;
  constexpr int VS = MV / 64;                           // 16-voxel subtiles per wave
  static_assert(MV % 64 == 0, "brick must give each wave whole 16-voxel subtiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int cls = blockIdx.z;
  int bid = blockIdx.x;
  const int bw_ = bid % a.nbw; bid /= a.nbw;
  const int bh_ = bid % a.nbh; bid /= a.nbh;
  const int bd_ = bid % a.nbd; const int b = bid / a.nbd;
  const int q0d = bd_ * BD, q0h = bh_ * BH, q0w = bw_ * BW;
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  const int co0 = blockIdx.y * (16 * NS);

  // ---- skip bricks with no active output voxel (block-sparse outputs) ----
  if (a.out_mask.m) {
    int any = 0;
    for (int v = tid; v < MV; v += 256) {
      const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
      if (od < a.Do && oh < a.Ho && ow < a.Wo && a.out_mask.active(b, od, oh, ow)) any = 1;
    }
    if (!__syncthreads_or(any)) return;
  }

  const int ED = a.ed[cls], EH = a.eh[cls], EW = a.ew[cls];
  const int nvox = ED * EH * EW;
  const int nit = (nvox * (ROWB / 16) + 255) >> 8;
  const int i0d = q0d * a.IS + a.mind[cls], i0h = q0h * a.IS + a.minh[cls], i0w = q0w * a.IS + a.minw[cls];

  // ---- per-thread staging plan (voxel -> global voxel index), computed once ----
  int svox[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    svox[it] = -1;
    const int idx = tid + it * 256;
    const int e = idx >> 2;
    if (it < nit && e < nvox) {
      const int ex = e % EW, ey = (e / EW) % EH, ez = e / (EW * EH);
      const int id = i0d + ez, ih = i0h + ey, iw = i0w + ex;
      if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi && a.in_mask.active(b, id, ih, iw))
        svox[it] = ((b * a.Di + id) * a.Hi + ih) * a.Wi + iw;
    }
  }

  // ---- per-lane fragment bases ----
  int vb[VS];                                            // LDS voxel index of this lane's voxel, tap shift excluded
#pragma unroll
  for (int j = 0; j < VS; ++j) {
    const int v = wave * (MV / 4) + j * 16 + r16;
    const int lw = v % BW, lh = (v / BW) % BH, ld = v / (BW * BH);
    vb[j] = ((ld * a.IS) * EH + lh * a.IS) * EW + lw * a.IS;
  }
  f32x4 acc[NS][VS];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ wg = (const T*)a.w;
  const int tb = a.tap_begin[cls], te = a.tap_begin[cls + 1];
  const int cchunk = (tid & 3) * EPC;                    // this thread's channel offset inside the slab when staging

  for (int kc = 0; kc < a.Cin; kc += KC) {
    __syncthreads();                                     // all fragment reads of the previous slab are done
    u32x4 stg[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      stg[it] = u32x4{0u, 0u, 0u, 0u};
      if (it < nit && svox[it] >= 0 && kc + cchunk < a.Cin)
        stg[it] = *(const u32x4*)(xg + (size_t)svox[it] * a.Cin + kc + cchunk);
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = tid + it * 256;
      if (it < nit && (idx >> 2) < nvox) *(u32x4*)(lds + (idx >> 2) * LROW + (idx & 3) * 16) = stg[it];
    }
    __syncthreads();
    const bool kvalid = (kc + g * EPC) < a.Cin;
    for (int t = tb; t < te; ++t) {
      const int tp = a.taps[t];
      const int sd = (tp & 15) - 8, sh = ((tp >> 4) & 15) - 8, sw = ((tp >> 8) & 15) - 8, widx = tp >> 12;
      const int tapoff = ((sd - a.mind[cls]) * EH + (sh - a.minh[cls])) * EW + (sw - a.minw[cls]);
      u32x4 af[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int co = co0 + i * 16 + r16;
        af[i] = u32x4{0u, 0u, 0u, 0u};
        if (kvalid && co < a.Cout) af[i] = *(const u32x4*)(wg + ((size_t)widx * a.Cout + co) * a.Cin + kc + g * EPC);
      }
#pragma unroll
      for (int j = 0; j < VS; ++j) {
        const u32x4 bf = *(const u32x4*)(lds + (vb[j] + tapoff) * LROW + g * 16);
#pragma unroll
        for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], bf, acc[i][j]);
      }
    }
  }

  // ---- epilogue: D row = cout 4g+r, col = voxel r16 ----
  T* __restrict__ yg = (T*)a.y;
#pragma unroll
  for (int j = 0; j < VS; ++j) {
    const int v = wave * (MV / 4) + j * 16 + r16;
    const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
    if (od >= a.Do || oh >= a.Ho || ow >= a.Wo) continue;
    const bool act = a.out_mask.active(b, od, oh, ow);
    const size_t ovox = ((size_t)(b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int co = co0 + i * 16 + g * 4;
      if (co >= a.Cout) continue;                        // Cout % 4 == 0 (C % 8 == 0 contract)
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float vv = acc[i][j][r] + (a.bias ? a.bias[co + r] : 0.f);
        o[r] = act ? vv : 0.f;
      }
      T* dst = yg + ovox * a.Cout + co;
      if (a.accumulate && act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += TT<T>::ld(dst + r);
      }
      if (sizeof(T) == 4) {
        *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
      } else {
        uint2 pk;
        pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
        pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
        *(uint2*)dst = pk;
      }
    }
  }
}

// host: tap tables ------------------------------------------------------------------------------
struct Plan { ConvArgs a; int bd, bh, bw; size_t lds; };

int build_plan(Plan& P, int mode, int k, int stride) {
  ConvArgs& a = P.a;
  const int pad = (mode == AM_CONVT_FWD || mode == AM_CONVT_DGRAD) ? 1 : k / 2;
  a.OS = 1; a.IS = 1; a.nclass = 1;
  if (mode == AM_CONV_FWD) a.IS = stride;
  else if (mode == AM_CONV_DGRAD) a.OS = stride;
  else if (mode == AM_CONVT_FWD) { a.OS = 2; if (k != 4 || stride != 2) return -2; }
  else if (mode == AM_CONVT_DGRAD) { a.IS = 2; if (k != 4 || stride != 2) return -2; }
  else return -2;
  if (a.OS == 2) a.nclass = 8;
  if (k * k * k > 64) return -2;
  int n = 0;
  for (int c = 0; c < a.nclass; ++c) {
    a.tap_begin[c] = n;
    const int p[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
    int mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    bool first = true;
    for (int td = 0; td < k; ++td) for (int th = 0; th < k; ++th) for (int tw = 0; tw < k; ++tw) {
      const int t[3] = {td, th, tw};
      int s[3]; bool ok = true;
      for (int d = 0; d < 3; ++d) {
        if (mode == AM_CONV_FWD || mode == AM_CONVT_DGRAD) s[d] = t[d] - pad;
        else if (a.OS == 1) s[d] = pad - t[d];
        else { const int num = p[d] + pad - t[d]; if (num & 1) { ok = false; break; } s[d] = num / 2; }
      }
      if (!ok) continue;
      if (n >= 64) return -2;
      a.taps[n++] = (s[0] + 8) | ((s[1] + 8) << 4) | ((s[2] + 8) << 8) | ((td * k * k + th * k + tw) << 12);
      for (int d = 0; d < 3; ++d) { if (first || s[d] < mn[d]) mn[d] = s[d]; if (first || s[d] > mx[d]) mx[d] = s[d]; }
      first = false;
    }
    a.mind[c] = mn[0]; a.minh[c] = mn[1]; a.minw[c] = mn[2];
    a.ed[c] = (P.bd - 1) * a.IS + (mx[0] - mn[0]) + 1;
    a.eh[c] = (P.bh - 1) * a.IS + (mx[1] - mn[1]) + 1;
    a.ew[c] = (P.bw - 1) * a.IS + (mx[2] - mn[2]) + 1;
  }
  for (int c = a.nclass; c <= 8; ++c) a.tap_begin[c] = n;
  size_t mxv = 0;
  for (int c = 0; c < a.nclass; ++c) { size_t v = (size_t)a.ed[c] * a.eh[c] * a.ew[c]; if (v > mxv) mxv = v; }
  if (mxv * (ROWB / 16) > (size_t)MAXIT * 256) return -3;
  P.lds = mxv * LROW;
  return 0;
}

template <typename T, int BD, int BH, int BW, int NS>
int launch(Plan& P, hipStream_t st) {
  ConvArgs& a = P.a;
  auto kern = conv_igemm_kernel<T, BD, BH, BW, NS>;
  static size_t attr_lds = 48 * 1024;             // raise the dynamic-LDS cap only when a launch needs it
  if (P.lds > attr_lds) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds) == hipSuccess) attr_lds = P.lds;
    (void)hipGetLastError();
  }
  dim3 grid(a.B * a.nbd * a.nbh * a.nbw, (a.Cout + 16 * NS - 1) / (16 * NS), a.nclass);
  AM_LAUNCH(kern, grid, dim3(256), P.lds, st, a);
  AM_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int dispatch(Plan& P, bool small, hipStream_t st) {
  const int co = P.a.Cout;
  if (small) {                                   // 4x4x4 brick (IS == 2: haloed source brick would not fit otherwise)
    if (co <= 16) return launch<T, 4, 4, 4, 1>(P, st);
    if (co <= 32) return launch<T, 4, 4, 4, 2>(P, st);
    return launch<T, 4, 4, 4, 4>(P, st);
  }
  if (co <= 16) return launch<T, 4, 8, 8, 1>(P, st);
  if (co <= 32) return launch<T, 4, 8, 8, 2>(P, st);
  return launch<T, 4, 8, 8, 4>(P, st);
}

}  // namespace

extern "C" int am_conv3d(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed,
                         const float* bias, void* y, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo,
                         int Cout, const uint8_t* in_mask, int in_bshift, const uint8_t* out_mask, int out_bshift,
                         int fd, int fh, int fw, int accumulate, void* stream) {
  if (Cin % 8 || Cout % 8) return -1;
  Plan P;
  ConvArgs& a = P.a;
  const bool small = (mode == AM_CONV_FWD && stride == 2) || mode == AM_CONVT_DGRAD;
  P.bd = 4; P.bh = small ? 4 : 8; P.bw = small ? 4 : 8;
  int rc = build_plan(P, mode, ksize, stride);
  if (rc) return rc;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
  a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.Qd = (Do + a.OS - 1) / a.OS; a.Qh = (Ho + a.OS - 1) / a.OS; a.Qw = (Wo + a.OS - 1) / a.OS;
  a.nbd = (a.Qd + P.bd - 1) / P.bd; a.nbh = (a.Qh + P.bh - 1) / P.bh; a.nbw = (a.Qw + P.bw - 1) / P.bw;
  a.in_mask = MaskView{in_mask, fd, fh, fw, in_bshift};
  a.out_mask = MaskView{out_mask, fd, fh, fw, out_bshift};
  a.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  return dtype == AM_DT_BF16 ? dispatch<bf16_t>(P, small, st) : dispatch<float>(P, small, st);
}
