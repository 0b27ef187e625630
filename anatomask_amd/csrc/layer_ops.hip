// Sparse layer zoo of the SparK encoder (reference: P/encoder3D.py = nnunetv2/training/nnUNetTrainer/variants/
// pretrain/encoder3D.py) for gfx950, on the same block-sparse channels-last tensors as the STUNet path: [B][D][H][W][C] + uint8 patch
// mask + block shift + active-patch list (am_mask_compact).  Inactive voxels hold don't-care bits; every kernel here reads them as 0
// (the reference's tensors carry explicit zeros there) and writes only active voxels.  All of these are HBM / VALU bound streaming
// or stencil kernels: no matrix cores.
//   am_voxel_norm_fwd/bwd   per-voxel normalisation over channel groups: SparseConvNeXtLayerNorm (:181-232, one group),
//                           SparseGroupNorm (:47-78: GroupNorm applied to the (N, C) matrix of active voxels, i.e. per voxel) and
//                           SparseGRN's sparse branch (:100-135: per-voxel L2 norm over C, Nx = G / (G + 1e-6))
//   am_pool3d_fwd/bwd       SparseMaxPooling / SparseAvgPooling (:31-36): dense pooling of the zero-filled tensor, output masked
//   am_dwconv3d             depthwise k^3 convolution (SparseConvNeXtBlock.dwconv :247, MedNeXt blocks) forward / data gradient
//   am_dwconv3d_wgrad       its weight (and bias) gradient
//   am_gelu_fwd/bwd, am_scale_residual_fwd/bwd   the pointwise tail of SparseConvNeXtBlock.forward (:258-275)
//   am_masked_mean_fwd/bwd  SparseAdaptiveAvgPooling (:171-179)
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename T> __device__ __forceinline__ void chunk_to_f2(const u32x4& c, f32x2* f);
template <> __device__ __forceinline__ void chunk_to_f2<float>(const u32x4& c, f32x2* f) {
  f[0] = f32x2{__uint_as_float(c[0]), __uint_as_float(c[1])}; f[1] = f32x2{__uint_as_float(c[2]), __uint_as_float(c[3])};
}
template <> __device__ __forceinline__ void chunk_to_f2<bf16_t>(const u32x4& c, f32x2* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = f32x2{__uint_as_float(c[i] << 16), __uint_as_float(c[i] & 0xffff0000u)};
}

// ---- enumeration of the ACTIVE voxels of a tensor: active-patch list x the voxels of a mask cell (w fastest), or all voxels (dense)
struct VoxGeo {
  int B, D, H, W, C;
  const int* plist;   // packed b<<24 | pd<<16 | ph<<8 | pw at the mask-grid resolution, or nullptr (dense)
  int bs;             // mask cell = (1 << bs) voxels per dim at this tensor's resolution
  long nvox;          // number of voxels to walk
  __device__ __forceinline__ long vox(long i, int& b, int& d, int& h, int& w) const {   // -> linear voxel index
    if (plist) {
      const int cs = 1 << bs, r = (int)(i & ((1 << (3 * bs)) - 1));
      const int pk = plist[i >> (3 * bs)];
      b = (pk >> 24) & 255;
      d = (((pk >> 16) & 255) << bs) | (r >> (2 * bs));
      h = (((pk >> 8) & 255) << bs) | ((r >> bs) & (cs - 1));
      w = ((pk & 255) << bs) | (r & (cs - 1));
    } else {
      w = (int)(i % W); long t = i / W;
      h = (int)(t % H); t /= H;
      d = (int)(t % D); b = (int)(t / D);
    }
    return (((long)b * D + d) * H + h) * W + w;
  }
};
inline VoxGeo mkvox(int B, int D, int H, int W, int C, const int* plist, int n_active, int bs) {
  VoxGeo g{B, D, H, W, C, plist, bs, 0};
  g.nvox = plist ? ((long)n_active << (3 * bs)) : (long)B * D * H * W;
  return g;
}

// sum of `val` over the L consecutive threads of a segment (pos = index inside the segment), fixed tree order: deterministic.
// Every thread of the workgroup calls it (L is uniform); sm holds 256 floats.
__device__ __forceinline__ float seg_sum(float* sm, int tid, int pos, int L, int P2, float val) {
  sm[tid] = val;
  __syncthreads();
  for (int s = P2 >> 1; s > 0; s >>= 1) {
    if (pos < s && pos + s < L) sm[tid] += sm[tid + s];
    __syncthreads();
  }
  const float r = sm[tid - pos];
  __syncthreads();
  return r;
}

constexpr int VN_GROUP = 0, VN_GRN = 1;

// y = (x - mean_g) * rstd_g * gamma_c + beta_c per voxel and channel group (biased variance, two passes), or the sparse GRN.
// thread = (voxel of the tile, 16-byte channel chunk); a tile is NV = 256 / CPV voxels.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void voxel_norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, VoxGeo g, int cg,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float sm[256];
  const int tid = threadIdx.x, CPV = g.C / EPC, NV = 256 / CPV;
  const int vl = tid / CPV, c = tid % CPV;
  const bool lane_ok = vl < NV;
  const int L = cg >= EPC ? cg / EPC : 1;                     // threads per group (cg % EPC == 0 or EPC % cg == 0, checked by the launcher)
  int P2 = 1; while (P2 < L) P2 <<= 1;
  const int pos = cg >= EPC ? c % L : 0;
  float ga[EPC], be[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) { ga[j] = lane_ok ? gamma[c * EPC + j] : 0.f; be[j] = (lane_ok && beta) ? beta[c * EPC + j] : 0.f; }
  const long ntile = (g.nvox + NV - 1) / NV;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long i = tile * NV + vl;
    const bool ok = lane_ok && i < g.nvox;
    int b, d, h, w;
    const long v = ok ? g.vox(i, b, d, h, w) : 0;
    float f[EPC];
    if (ok) chunk_to_f<T>(*(const u32x4*)(x + v * g.C + c * EPC), f);
    else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) f[j] = 0.f;
    }
    float o[EPC];
    if constexpr (MODE == VN_GRN) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) s += f[j] * f[j];
      const float G = sqrtf(seg_sum(sm, tid, pos, L, P2, s));
      const float n = G / (G + 1e-6f);
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = ga[j] * (f[j] * n) + be[j];
    } else if (cg >= EPC) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) s += f[j];
      const float mean = seg_sum(sm, tid, pos, L, P2, s) / cg;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) q += (f[j] - mean) * (f[j] - mean);
      const float rstd = rsqrtf(seg_sum(sm, tid, pos, L, P2, q) / cg + eps);
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = (f[j] - mean) * rstd * ga[j] + be[j];
    } else {                                                  // groups of 1, 2 or 4 channels inside the chunk
      for (int j0 = 0; j0 < EPC; j0 += cg) {
        float s = 0.f, q = 0.f;
        for (int j = j0; j < j0 + cg; ++j) s += f[j];
        const float mean = s / cg;
        for (int j = j0; j < j0 + cg; ++j) q += (f[j] - mean) * (f[j] - mean);
        const float rstd = rsqrtf(q / cg + eps);
        for (int j = j0; j < j0 + cg; ++j) o[j] = (f[j] - mean) * rstd * ga[j] + be[j];
      }
    }
    if (ok) *(u32x4*)(y + v * g.C + c * EPC) = f_to_chunk<T>(o);
  }
}

// backward: dx, and dgamma / dbeta accumulated over the voxels (fp32 atomics, one per workgroup and channel)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void voxel_norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, VoxGeo g,
                                                             int cg, const float* __restrict__ gamma, float eps,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float sm[256];
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, CPV = g.C / EPC, NV = 256 / CPV;
  const int vl = tid / CPV, c = tid % CPV;
  const bool lane_ok = vl < NV;
  const int L = cg >= EPC ? cg / EPC : 1;
  int P2 = 1; while (P2 < L) P2 <<= 1;
  const int pos = cg >= EPC ? c % L : 0;
  float ga[EPC], ag[EPC], ab[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) { ga[j] = lane_ok ? gamma[c * EPC + j] : 0.f; ag[j] = 0.f; ab[j] = 0.f; }
  const long ntile = (g.nvox + NV - 1) / NV;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long i = tile * NV + vl;
    const bool ok = lane_ok && i < g.nvox;
    int b, d, h, w;
    const long v = ok ? g.vox(i, b, d, h, w) : 0;
    float f[EPC], gy[EPC], o[EPC];
    if (ok) { chunk_to_f<T>(*(const u32x4*)(x + v * g.C + c * EPC), f); chunk_to_f<T>(*(const u32x4*)(dy + v * g.C + c * EPC), gy); }
    else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) { f[j] = 0.f; gy[j] = 0.f; }
    }
    if constexpr (MODE == VN_GRN) {
      float s = 0.f, t = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) { s += f[j] * f[j]; t += ga[j] * f[j] * gy[j]; }
      const float G = sqrtf(seg_sum(sm, tid, pos, L, P2, s));
      const float S = seg_sum(sm, tid, pos, L, P2, t);
      const float n = G / (G + 1e-6f);
      const float k = G > 0.f ? S * 1e-6f / ((G + 1e-6f) * (G + 1e-6f)) / G : 0.f;    // dn/dG * dG/dx_k = eps / (G+eps)^2 * x_k / G
#pragma unroll
      for (int j = 0; j < EPC; ++j) { o[j] = ga[j] * n * gy[j] + k * f[j]; ag[j] += gy[j] * f[j] * n; ab[j] += gy[j]; }
    } else if (cg >= EPC) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) s += f[j];
      const float mean = seg_sum(sm, tid, pos, L, P2, s) / cg;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) q += (f[j] - mean) * (f[j] - mean);
      const float rstd = rsqrtf(seg_sum(sm, tid, pos, L, P2, q) / cg + eps);
      float a = 0.f, bb = 0.f;
#pragma unroll
      for (int j = 0; j < EPC; ++j) { const float xh = (f[j] - mean) * rstd; a += ga[j] * gy[j]; bb += ga[j] * gy[j] * xh; }
      a = seg_sum(sm, tid, pos, L, P2, a) / cg;
      bb = seg_sum(sm, tid, pos, L, P2, bb) / cg;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        const float xh = (f[j] - mean) * rstd;
        o[j] = rstd * (ga[j] * gy[j] - a - xh * bb);
        ag[j] += gy[j] * xh; ab[j] += gy[j];
      }
    } else {
      for (int j0 = 0; j0 < EPC; j0 += cg) {
        float s = 0.f, q = 0.f, a = 0.f, bb = 0.f;
        for (int j = j0; j < j0 + cg; ++j) s += f[j];
        const float mean = s / cg;
        for (int j = j0; j < j0 + cg; ++j) q += (f[j] - mean) * (f[j] - mean);
        const float rstd = rsqrtf(q / cg + eps);
        for (int j = j0; j < j0 + cg; ++j) { const float xh = (f[j] - mean) * rstd; a += ga[j] * gy[j]; bb += ga[j] * gy[j] * xh; }
        a /= cg; bb /= cg;
        for (int j = j0; j < j0 + cg; ++j) {
          const float xh = (f[j] - mean) * rstd;
          o[j] = rstd * (ga[j] * gy[j] - a - xh * bb);
          ag[j] += gy[j] * xh; ab[j] += gy[j];
        }
      }
    }
    if (ok) *(u32x4*)(dx + v * g.C + c * EPC) = f_to_chunk<T>(o);
  }
  // fold the NV voxel lanes of every channel chunk (fixed order), one atomic per workgroup and channel
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPC; ++j) red[tid * 8 + j] = pass == 0 ? ag[j] : ab[j];
    __syncthreads();
    float* dst = pass == 0 ? dgamma : dbeta;
    if (dst && tid < CPV) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        float s = 0.f;
        for (int q = 0; q < NV; ++q) s += red[(q * CPV + tid) * 8 + j];
        atomicAdd(&dst[(size_t)(blockIdx.x % AM_LAYER_REP) * g.C + tid * EPC + j], s);
      }
    }
  }
}

// ---- pooling: thread = (active output voxel, channel chunk)
template <typename T, int OP>   // OP 0 = max (ties: first in (d, h, w) scan order, as torch), 1 = average
__global__ __launch_bounds__(256) void pool3d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int* __restrict__ idx, VoxGeo go, int Di,
                                                         int Hi, int Wi, int k, int st, int pad, int dil, int count_include_pad, MaskView min) {
  constexpr int EPC = TT<T>::EPC;
  const int CPV = go.C / EPC;
  const long n = go.nvox * CPV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = (int)(e % CPV);
    int b, od, oh, ow;
    const long vo = go.vox(e / CPV, b, od, oh, ow);
    float acc[EPC]; int am[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) { acc[j] = OP == 0 ? -INFINITY : 0.f; am[j] = -1; }
    int cnt = 0;
    for (int td = 0; td < k; ++td) for (int th = 0; th < k; ++th) for (int tw = 0; tw < k; ++tw) {
      const int id = od * st - pad + td * dil, ih = oh * st - pad + th * dil, iw = ow * st - pad + tw * dil;
      if ((unsigned)id >= (unsigned)Di || (unsigned)ih >= (unsigned)Hi || (unsigned)iw >= (unsigned)Wi) continue;
      ++cnt;
      float f[EPC];
      if (min.active(b, id, ih, iw)) chunk_to_f<T>(*(const u32x4*)(x + ((((long)b * Di + id) * Hi + ih) * Wi + iw) * go.C + c * EPC), f);
      else {
#pragma unroll
        for (int j = 0; j < EPC; ++j) f[j] = 0.f;
      }
      const int li = (id * Hi + ih) * Wi + iw;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        if (OP == 0) { if (f[j] > acc[j] || am[j] < 0) { acc[j] = f[j]; am[j] = li; } }
        else acc[j] += f[j];
      }
    }
    if (OP == 1) {
      // count_include_pad: the window clipped to the PADDED extent (torch: a ceil_mode window may hang over the padding's end)
      auto ext = [&](int o, int n_) { const int a = o * st - pad, hi_ = a + k > n_ + pad ? n_ + pad : a + k; return hi_ - a; };
      const float div = count_include_pad ? (float)(ext(od, Di) * ext(oh, Hi) * ext(ow, Wi)) : (float)cnt;
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] /= div;
    }
    *(u32x4*)(y + vo * go.C + c * EPC) = f_to_chunk<T>(acc);
    if (OP == 0 && idx) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) idx[vo * go.C + c * EPC + j] = am[j];
    }
  }
}

// gather form of the pooling backward (no atomics): thread = (active input voxel, chunk), loops over the windows that cover it
template <typename T, int OP>
__global__ __launch_bounds__(256) void pool3d_bwd_kernel(const T* __restrict__ dy, const int* __restrict__ idx, T* __restrict__ dx, VoxGeo gi,
                                                         int Do, int Ho, int Wo, int k, int st, int pad, int dil, int count_include_pad, MaskView mout) {
  constexpr int EPC = TT<T>::EPC;
  const int CPV = gi.C / EPC;
  const long n = gi.nvox * CPV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = (int)(e % CPV);
    int b, d, h, w;
    const long vi = gi.vox(e / CPV, b, d, h, w);
    const int li = (d * gi.H + h) * gi.W + w;
    float acc[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
    // (dilated max windows: a superset of the windows that hold the voxel -- the argmax comparison picks the real ones)
    auto lo = [&](int p) { const int a = p + pad - (k - 1) * dil; return a <= 0 ? 0 : (a + st - 1) / st; };
    auto hi = [&](int p, int n_) { const int a = (p + pad) / st; return a < n_ - 1 ? a : n_ - 1; };
    for (int od = lo(d); od <= hi(d, Do); ++od) for (int oh = lo(h); oh <= hi(h, Ho); ++oh) for (int ow = lo(w); ow <= hi(w, Wo); ++ow) {
      if (!mout.active(b, od, oh, ow)) continue;
      const long vo = (((long)b * Do + od) * Ho + oh) * Wo + ow;
      float gy[EPC];
      chunk_to_f<T>(*(const u32x4*)(dy + vo * gi.C + c * EPC), gy);
      if (OP == 0) {
#pragma unroll
        for (int j = 0; j < EPC; ++j) if (idx[vo * gi.C + c * EPC + j] == li) acc[j] += gy[j];
      } else {
        float div;
        if (!count_include_pad) {
          auto ext = [&](int o, int n_) { const int a = o * st - pad, lo_ = a < 0 ? 0 : a, hi_ = a + k > n_ ? n_ : a + k; return hi_ - lo_; };
          div = (float)(ext(od, gi.D) * ext(oh, gi.H) * ext(ow, gi.W));
        } else {
          auto ext = [&](int o, int n_) { const int a = o * st - pad, hi_ = a + k > n_ + pad ? n_ + pad : a + k; return hi_ - a; };
          div = (float)(ext(od, gi.D) * ext(oh, gi.H) * ext(ow, gi.W));
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] += gy[j] / div;
      }
    }
    *(u32x4*)(dx + vi * gi.C + c * EPC) = f_to_chunk<T>(acc);
  }
}

// Branch-free staging of a haloed brick of one 16-byte channel chunk: every thread computes the addresses of all its rows, issues all
// patch-mask bytes, then all data loads (clamped to a valid address), and only then selects -- two global round trips per workgroup
// instead of two per row (a conditional load inside the loop serialises its latency).
template <typename T, int ED, int EH, int EW, int P>
__device__ __forceinline__ void stage_brick(u32x4* __restrict__ xb, const T* __restrict__ x, int b, int d0, int h0, int w0, int D, int H, int W, int C,
                                            int c0, const MaskView& mask, int tid) {
  constexpr int NE = ED * EH * EW, NIT = (NE + 255) / 256;
  uint8_t mb[NIT]; bool inr[NIT]; size_t off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * 256;
    const int ex = e % EW, ey = (e / EW) % EH, ez = e / (EW * EH);
    const int id = d0 + ez - P, ih = h0 + ey - P, iw = w0 + ex - P;
    inr[it] = e < NE && (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    mb[it] = mask.m ? mask.peek(b, id, ih, iw, inr[it]) : (uint8_t)1;
    off[it] = inr[it] ? ((((size_t)b * D + id) * H + ih) * W + iw) * C + c0 : (size_t)c0;
  }
  u32x4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) v[it] = *(const u32x4*)(x + off[it]);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * 256;
    if (e < NE) xb[e] = (inr[it] && mb[it]) ? v[it] : u32x4{0u, 0u, 0u, 0u};
  }
}

// ---- depthwise k^3 convolution, stride 1, padding k/2.  One workgroup = one 8x8x16 brick x one 16-byte channel chunk: the haloed
// brick of that chunk is staged once in LDS (bounds and the patch mask applied), a thread owns 4 consecutive w outputs and per
// (td, th) row reads 4 + k - 1 chunks for 4 k multiply-adds per channel.  Weights of the chunk live in LDS as [tap][EPC] floats
// (broadcast reads).  flip: data gradient (taps mirrored, x = dy).
constexpr int DWD = 8, DWH = 8, DWW = 16;
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     T* __restrict__ y, int B, int D, int H, int W, int C, MaskView mask, int flip) {
  constexpr int EPC = TT<T>::EPC, P = K / 2, NT = K * K * K;
  constexpr int ED = DWD + K - 1, EH = DWH + K - 1, EW = DWW + K - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32x4* xb = (u32x4*)lds;                                    // [ED][EH][EW] chunks
  float* wl = (float*)(lds + (size_t)ED * EH * EW * 16);      // [NT][EPC]
  // workgroup id -> (brick, channel chunk): the chunks of one brick get ids that differ by 8, i.e. the same XCD under round-robin
  // dispatch and neighbouring dispatch slots: they share the brick's cache lines in that XCD's L2 (each reads 16 of a voxel's C*2
  // bytes; brick-major order with the chunk in blockIdx.y re-read the whole tensor from HBM once per chunk)
  const int tid = threadIdx.x, nchunk = C / EPC;
  const int nbw = (W + DWW - 1) / DWW, nbh = (H + DWH - 1) / DWH, nbd = (D + DWD - 1) / DWD;
  const int t_ = blockIdx.x >> 3;
  const int c0 = (t_ % nchunk) * EPC;
  int bid = (t_ / nchunk) * 8 + (blockIdx.x & 7);
  if (bid >= B * nbd * nbh * nbw) return;
  const int bw_ = bid % nbw; bid /= nbw;
  const int bh_ = bid % nbh; bid /= nbh;
  const int bd_ = bid % nbd; const int b = bid / nbd;
  const int d0 = bd_ * DWD, h0 = bh_ * DWH, w0 = bw_ * DWW;
  // any active output voxel in this brick? (mask cells are powers of two: test one voxel per cell-sized step)
  if (mask.m) {
    const int cs = 1 << mask.bs;
    int any = 0;
    for (int e = tid; e < DWD * DWH * DWW; e += 256) {
      const int lw = e % DWW, lh = (e / DWW) % DWH, ld = e / (DWW * DWH);
      if ((lw % cs) && lw) continue;
      if ((lh % cs) && lh) continue;
      if ((ld % cs) && ld) continue;
      if (d0 + ld < D && h0 + lh < H && w0 + lw < W && mask.active(b, d0 + ld, h0 + lh, w0 + lw)) any = 1;
    }
    if (!__syncthreads_or(any)) return;
  }
  for (int i = tid; i < NT * EPC; i += 256) {
    const int t = i / EPC, j = i % EPC;
    wl[i] = w[(size_t)(c0 + j) * NT + (flip ? NT - 1 - t : t)];
  }
  stage_brick<T, ED, EH, EW, P>(xb, x, b, d0, h0, w0, D, H, W, C, c0, mask, tid);
  __syncthreads();
  const int lw0 = (tid & 3) * 4, lh = (tid >> 2) & 7, ld = tid >> 5;      // 4 w-quads x 8 h x 8 d
  // two channels per register pair: the multiply-adds compile to v_pk_fma_f32 (the packed rate is the 157 TFLOP/s vector peak)
  f32x2 acc[4][EPC / 2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < EPC / 2; ++j) acc[q][j] = (bias && !flip) ? f32x2{bias[c0 + 2 * j], bias[c0 + 2 * j + 1]} : f32x2{0.f, 0.f};
#pragma unroll 1
  for (int td = 0; td < K; ++td)
#pragma unroll 1
    for (int th = 0; th < K; ++th) {
      const u32x4* row = xb + ((ld + td) * EH + lh + th) * EW + lw0;
      f32x2 xr[4 + K - 1][EPC / 2];
#pragma unroll
      for (int i = 0; i < 4 + K - 1; ++i) chunk_to_f2<T>(row[i], xr[i]);
      const f32x4* wr = (const f32x4*)(wl + (td * K + th) * K * EPC);
#pragma unroll
      for (int tw = 0; tw < K; ++tw) {
#pragma unroll
        for (int j4 = 0; j4 < EPC / 4; ++j4) {
          const f32x4 w4 = wr[tw * (EPC / 4) + j4];              // one 16-byte broadcast read = 4 channels of this tap
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2 wv = h2 ? f32x2{w4[2], w4[3]} : f32x2{w4[0], w4[1]};
            const int j = j4 * 2 + h2;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][j] = __builtin_elementwise_fma(wv, xr[q + tw][j], acc[q][j]);
          }
        }
      }
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int od = d0 + ld, oh = h0 + lh, ow = w0 + lw0 + q;
    float o[EPC];
#pragma unroll
    for (int j = 0; j < EPC / 2; ++j) { o[2 * j] = acc[q][j][0]; o[2 * j + 1] = acc[q][j][1]; }
    if (od < D && oh < H && ow < W && mask.active(b, od, oh, ow))
      *(u32x4*)(y + ((((size_t)b * D + od) * H + oh) * W + ow) * C + c0) = f_to_chunk<T>(o);
  }
}

// weight gradient: dw[c][t] += sum_v dy[v][c] * xm[v + t - P][c].  Lane = (one (td, th) row of taps, one of NSUB = 64 / K^2 interleaved
// subsets of the brick's w-quads), holding K x EPC partial sums across ALL bricks of its workgroup; the 4 waves split the (d, h)
// rows of a brick.  Per group of 4 consecutive w voxels a lane reads 4 dy chunks and 4 + K - 1 x chunks of its own row.  One atomic
// flush per workgroup.  Workgroup id -> (slot, chunk) as in dwconv_kernel (the chunks of a brick share an XCD's L2).
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                           float* __restrict__ db, int B, int D, int H, int W, int C, MaskView mask, int nslot) {
  constexpr int EPC = TT<T>::EPC, P = K / 2, NT = K * K * K, KK = K * K, NSUB = 64 / KK;
  constexpr int ED = DWD + K - 1, EH = DWH + K - 1, EW = DWW + K - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32x4* xb = (u32x4*)lds;
  u32x4* yb = xb + ED * EH * EW;                              // [DWD][DWH][DWW] chunks of dy
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nchunk = C / EPC;
  const int t_ = blockIdx.x >> 3;
  const int c0 = (t_ % nchunk) * EPC, slot = (t_ / nchunk) * 8 + (blockIdx.x & 7);
  const bool live = lane < KK * NSUB;
  const int sub = live ? lane / KK : 0, tl = lane % KK, td = tl / K, th = tl % K;
  const int nbw = (W + DWW - 1) / DWW, nbh = (H + DWH - 1) / DWH, nbd = (D + DWD - 1) / DWD;
  const int nbrick = B * nbd * nbh * nbw;
  f32x2 acc[K][EPC / 2];
  float sb[EPC];
#pragma unroll
  for (int t = 0; t < K; ++t)
#pragma unroll
    for (int j = 0; j < EPC / 2; ++j) acc[t][j] = f32x2{0.f, 0.f};
#pragma unroll
  for (int j = 0; j < EPC; ++j) sb[j] = 0.f;
  for (int brick = slot; brick < nbrick; brick += nslot) {
    int bid = brick;
    const int bw_ = bid % nbw; bid /= nbw;
    const int bh_ = bid % nbh; bid /= nbh;
    const int bd_ = bid % nbd; const int b = bid / nbd;
    const int d0 = bd_ * DWD, h0 = bh_ * DWH, w0 = bw_ * DWW;
    __syncthreads();                                          // previous brick's reads are done
    int any = 0;
    {
      constexpr int NY = DWD * DWH * DWW / 256;
      uint8_t mb[NY]; bool inr[NY]; size_t off[NY];
#pragma unroll
      for (int it = 0; it < NY; ++it) {
        const int e = tid + it * 256;
        const int od = d0 + e / (DWW * DWH), oh = h0 + (e / DWW) % DWH, ow = w0 + e % DWW;
        inr[it] = od < D && oh < H && ow < W;
        mb[it] = mask.m ? mask.peek(b, od, oh, ow, inr[it]) : (uint8_t)1;
        off[it] = inr[it] ? ((((size_t)b * D + od) * H + oh) * W + ow) * C + c0 : (size_t)c0;
      }
      u32x4 v[NY];
#pragma unroll
      for (int it = 0; it < NY; ++it) v[it] = *(const u32x4*)(dy + off[it]);
#pragma unroll
      for (int it = 0; it < NY; ++it) {
        const bool on = inr[it] && mb[it];
        yb[tid + it * 256] = on ? v[it] : u32x4{0u, 0u, 0u, 0u};
        any |= on ? 1 : 0;
      }
    }
    if (!__syncthreads_or(any)) continue;                     // no active voxel: dy == 0 on the whole brick
    stage_brick<T, ED, EH, EW, P>(xb, x, b, d0, h0, w0, D, H, W, C, c0, mask, tid);
    __syncthreads();
    // this wave's (d, h) rows r = wave, wave + 4, ...; a row has DWW / 4 w-quads; work item i = (row index, quad), lane takes i % NSUB == sub
    constexpr int NQ = DWW / 4, NITEM = (DWD * DWH / 4) * NQ;
#pragma unroll 1
    for (int i = sub; i < NITEM; i += NSUB) {
      const int r = wave + 4 * (i / NQ), wq = (i % NQ) * 4;
      const int ld = r / DWH, lh = r % DWH;
      const u32x4* xrow = xb + ((ld + td) * EH + lh + th) * EW + wq;
      const u32x4* yrow = yb + (ld * DWH + lh) * DWW + wq;
      f32x2 gy[4][EPC / 2], xr[4 + K - 1][EPC / 2];
#pragma unroll
      for (int q = 0; q < 4; ++q) chunk_to_f2<T>(yrow[q], gy[q]);
#pragma unroll
      for (int q = 0; q < 4 + K - 1; ++q) chunk_to_f2<T>(xrow[q], xr[q]);
#pragma unroll
      for (int t = 0; t < K; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < EPC / 2; ++j) acc[t][j] = __builtin_elementwise_fma(gy[q][j], xr[q + t][j], acc[t][j]);
      if (tl == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < EPC / 2; ++j) { sb[2 * j] += gy[q][j][0]; sb[2 * j + 1] += gy[q][j][1]; }
      }
    }
  }
  // fold the 4 waves x NSUB subsets (fixed order), then one atomic per workgroup and (channel, tap)
  __syncthreads();
  float* red = (float*)lds;                                   // [4][64][K][EPC], then [4][NSUB][EPC] bias sums
  if (live) {
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
      for (int j = 0; j < EPC; ++j) red[((wave * 64 + lane) * K + t) * EPC + j] = acc[t][j / 2][j & 1];
    if (tl == 0) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) red[4 * 64 * K * EPC + (wave * NSUB + sub) * EPC + j] = sb[j];
    }
  }
  __syncthreads();
  for (int i = tid; i < NT * EPC; i += 256) {
    const int j = i % EPC, t = i / EPC, tw = t % K, row = t / K;    // t = (td*K + th)*K + tw
    float s = 0.f;
    for (int wv = 0; wv < 4; ++wv)
      for (int sb_ = 0; sb_ < NSUB; ++sb_) s += red[((wv * 64 + sb_ * KK + row) * K + tw) * EPC + j];
    if (s != 0.f) atomicAdd(&dw[(size_t)(c0 + j) * NT + t], s);
  }
  if (db && tid < EPC) {
    float s = 0.f;
    for (int q = 0; q < 4 * NSUB; ++q) s += red[4 * 64 * K * EPC + q * EPC + tid];
    if (s != 0.f) atomicAdd(&db[c0 + tid], s);
  }
}

// ---- depthwise k^3 convolution with stride 2 (MedNeXtDownBlock.conv1, P/MedNeXt_head.py:335-341): the down blocks touch 1/8 of the
// voxels of the stride-1 blocks around them and an input voxel is shared by only ~(k/2)^3 outputs, so these are direct kernels: one
// thread per (active voxel, 16-byte channel chunk), the k^3 taps read through L1 / L2, the chunk's weights in LDS.
// MODE 0: forward  y[o] = b + sum_t w[t] xm[2o + t - P];  MODE 1: data gradient  dx[i] = sum_{t : (i + P - t) even} w[t] dym[(i + P - t) / 2].
template <typename T, int K, int MODE>
__global__ __launch_bounds__(256) void dwconv_s2_kernel(const T* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ dst, VoxGeo g, int Ds, int Hs, int Ws, MaskView msrc) {
  constexpr int EPC = TT<T>::EPC, P = K / 2, NT = K * K * K;
  __shared__ float wl[NT * 8];
  const int c0 = blockIdx.y * EPC;
  for (int i = threadIdx.x; i < NT * EPC; i += 256) wl[i] = w[(size_t)(c0 + i % EPC) * NT + i / EPC];
  __syncthreads();
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < g.nvox; e += (long)gridDim.x * 256) {
    int b, d, h, wv;
    const long v = g.vox(e, b, d, h, wv);
    float acc[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] = (MODE == 0 && bias) ? bias[c0 + j] : 0.f;
    if (MODE == 0) {
      for (int td = 0; td < K; ++td) for (int th = 0; th < K; ++th) for (int tw = 0; tw < K; ++tw) {
        const int id = 2 * d + td - P, ih = 2 * h + th - P, iw = 2 * wv + tw - P;
        if ((unsigned)id >= (unsigned)Ds || (unsigned)ih >= (unsigned)Hs || (unsigned)iw >= (unsigned)Ws || !msrc.active(b, id, ih, iw)) continue;
        float f[EPC];
        chunk_to_f<T>(*(const u32x4*)(src + ((((size_t)b * Ds + id) * Hs + ih) * Ws + iw) * g.C + c0), f);
        const float* wt = wl + ((td * K + th) * K + tw) * EPC;
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] += wt[j] * f[j];
      }
    } else {
      for (int td = (d + P) & 1; td < K; td += 2) for (int th = (h + P) & 1; th < K; th += 2) for (int tw = (wv + P) & 1; tw < K; tw += 2) {
        const int od = (d + P - td) / 2, oh = (h + P - th) / 2, ow = (wv + P - tw) / 2;      // numerators are even and may be negative only by -? no: checked below
        if (d + P - td < 0 || h + P - th < 0 || wv + P - tw < 0 || od >= Ds || oh >= Hs || ow >= Ws || !msrc.active(b, od, oh, ow)) continue;
        float f[EPC];
        chunk_to_f<T>(*(const u32x4*)(src + ((((size_t)b * Ds + od) * Hs + oh) * Ws + ow) * g.C + c0), f);
        const float* wt = wl + ((td * K + th) * K + tw) * EPC;
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] += wt[j] * f[j];
      }
    }
    *(u32x4*)(dst + v * g.C + c0) = f_to_chunk<T>(acc);
  }
}

// weight gradient of the stride-2 depthwise convolution: dw[c][t] += sum_o dy[o][c] * xm[2o + t - P][c].  One workgroup = one channel
// chunk x a strided share of the active output voxels; thread t owns taps t and t + 256 (k = 7: 343 taps) with EPC partial sums each,
// every thread reads the same dy chunk (broadcast) and its own x chunk.  One atomic flush per workgroup.
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_s2_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                              float* __restrict__ db, VoxGeo go, int Dx, int Hx, int Wx, MaskView mx) {
  constexpr int EPC = TT<T>::EPC, P = K / 2, NT = K * K * K, TPT = (NT + 255) / 256;
  const int c0 = blockIdx.y * EPC, tid = threadIdx.x;
  float acc[TPT][EPC], sb[EPC];
#pragma unroll
  for (int q = 0; q < TPT; ++q)
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[q][j] = 0.f;
#pragma unroll
  for (int j = 0; j < EPC; ++j) sb[j] = 0.f;
  for (long e = blockIdx.x; e < go.nvox; e += gridDim.x) {
    int b, od, oh, ow;
    const long vo = go.vox(e, b, od, oh, ow);
    float gy[EPC];
    chunk_to_f<T>(*(const u32x4*)(dy + vo * go.C + c0), gy);
    if (tid == 0) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) sb[j] += gy[j];
    }
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int t = tid + q * 256;
      if (t >= NT) continue;
      const int id = 2 * od + t / (K * K) - P, ih = 2 * oh + (t / K) % K - P, iw = 2 * ow + t % K - P;
      if ((unsigned)id >= (unsigned)Dx || (unsigned)ih >= (unsigned)Hx || (unsigned)iw >= (unsigned)Wx || !mx.active(b, id, ih, iw)) continue;
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + ((((size_t)b * Dx + id) * Hx + ih) * Wx + iw) * go.C + c0), f);
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[q][j] += gy[j] * f[j];
    }
  }
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int t = tid + q * 256;
    if (t < NT) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) if (acc[q][j] != 0.f) atomicAdd(&dw[(size_t)(c0 + j) * NT + t], acc[q][j]);
    }
  }
  if (db && tid == 0) {
#pragma unroll
    for (int j = 0; j < EPC; ++j) if (sb[j] != 0.f) atomicAdd(&db[c0 + j], sb[j]);
  }
}

// ---- pointwise tail of the ConvNeXt block
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float v) {
  return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}
template <typename T, int BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, VoxGeo g) {
  constexpr int EPC = TT<T>::EPC;
  const int CPV = g.C / EPC;
  const long n = g.nvox * CPV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    int b, d, h, w;
    const long off = g.vox(e / CPV, b, d, h, w) * g.C + (e % CPV) * EPC;
    float f[EPC], o[EPC];
    chunk_to_f<T>(*(const u32x4*)(x + off), f);
    if (BWD) {
      float gy[EPC];
      chunk_to_f<T>(*(const u32x4*)(dy + off), gy);
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = gy[j] * gelu_d(f[j]);
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = gelu_f(f[j]);
    }
    *(u32x4*)(out + off) = f_to_chunk<T>(o);
  }
}

// y = res + gamma_c * x on active voxels (layer scale + residual).  BWD: dx = gamma * dy, dgamma += sum_v dy * x (dres = dy: the caller's)
template <typename T, int BWD>
__global__ __launch_bounds__(256) void scale_residual_kernel(const T* __restrict__ x, const T* __restrict__ other, const float* __restrict__ gamma,
                                                             T* __restrict__ out, float* __restrict__ dgamma, VoxGeo g) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[256 * 8];
  const int CPV = g.C / EPC, NV = 256 / CPV, tid = threadIdx.x, vl = tid / CPV, c = tid % CPV;
  const bool lane_ok = vl < NV;
  float ga[EPC], ag[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) { ga[j] = (lane_ok && gamma) ? gamma[c * EPC + j] : 1.f; ag[j] = 0.f; }
  const long ntile = (g.nvox + NV - 1) / NV;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long i = tile * NV + vl;
    if (!lane_ok || i >= g.nvox) continue;
    int b, d, h, w;
    const long off = g.vox(i, b, d, h, w) * g.C + c * EPC;
    float f[EPC], r[EPC], o[EPC];
    chunk_to_f<T>(*(const u32x4*)(x + off), f);
    chunk_to_f<T>(*(const u32x4*)(other + off), r);             // fwd: residual input; bwd: dy
    if (BWD) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) { o[j] = ga[j] * r[j]; ag[j] += r[j] * f[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = r[j] + ga[j] * f[j];
    }
    *(u32x4*)(out + off) = f_to_chunk<T>(o);
  }
  if (BWD && dgamma) {
#pragma unroll
    for (int j = 0; j < EPC; ++j) red[tid * 8 + j] = ag[j];
    __syncthreads();
    if (tid < CPV) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        float s = 0.f;
        for (int q = 0; q < NV; ++q) s += red[(q * CPV + tid) * 8 + j];
        atomicAdd(&dgamma[(size_t)(blockIdx.x % AM_LAYER_REP) * g.C + tid * EPC + j], s);
      }
    }
  }
}

// ---- SparseAdaptiveAvgPooling: mean[b][c] = sum over the active voxels of sample b / (count_b + 1e-6).  One workgroup per sample and
// 16-byte chunk (deterministic: fixed walk + tree).  BWD: dx[v][c] = dmean[b][c] / (count_b + 1e-6) on active voxels.
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_fwd_kernel(const T* __restrict__ x, float* __restrict__ mean, int B, int D, int H, int W, int C,
                                                              MaskView mask) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[256 * 8];
  __shared__ float cntl[256];
  const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const long nv = (long)D * H * W;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
  float cnt = 0.f;
  for (long v = tid; v < nv; v += 256) {
    const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
    if (!mask.active(b, d, h, w)) continue;
    float f[EPC];
    chunk_to_f<T>(*(const u32x4*)(x + ((size_t)b * nv + v) * C + c * EPC), f);
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] += f[j];
    cnt += 1.f;
  }
#pragma unroll
  for (int j = 0; j < EPC; ++j) red[tid * 8 + j] = acc[j];
  cntl[tid] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) red[tid * 8 + j] += red[(tid + s) * 8 + j];
      cntl[tid] += cntl[tid + s];
    }
    __syncthreads();
  }
  if (tid < EPC) mean[(size_t)b * C + c * EPC + tid] = red[tid] / (cntl[0] + 1e-6f);
}
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ dmean, const float* __restrict__ count, T* __restrict__ dx,
                                                              VoxGeo g) {
  constexpr int EPC = TT<T>::EPC;
  const int CPV = g.C / EPC;
  const long n = g.nvox * CPV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    int b, d, h, w;
    const int c = (int)(e % CPV);
    const long off = g.vox(e / CPV, b, d, h, w) * g.C + c * EPC;
    float o[EPC];
    const float inv = 1.f / (count[b] + 1e-6f);
#pragma unroll
    for (int j = 0; j < EPC; ++j) o[j] = dmean[(size_t)b * g.C + c * EPC + j] * inv;
    *(u32x4*)(dx + off) = f_to_chunk<T>(o);
  }
}

inline int nblocks(long work) { long nb = (work + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb)); }
inline bool list_ok(const uint8_t* mask, const int* plist, int n_active) { return !mask || (plist && n_active > 0); }

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) do { if ((dtype) == AM_DT_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)
// (a voxel row must fit the 256 threads of a workgroup as 16-byte chunks: C <= 2048 in bf16, C <= 1024 in fp32 -- refuse, never mis-compute)
#define CHK_C(C) do { if ((C) % 8 || (C) > 2048 || (C) <= 0 || (dtype != AM_DT_BF16 && (C) > 1024)) return -1; } while (0)

extern "C" {

int am_voxel_norm_fwd(int dtype, int kind, const void* x, void* y, int B, int D, int H, int W, int C, int groups, const float* gamma,
                      const float* beta, float eps, const uint8_t* mask, int bshift, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if (!gamma || groups < 1 || C % groups || (kind != 0 && kind != 1)) return -1;
  if (!list_ok(mask, active_list, n_active)) return -2;         // block-sparse tensors are walked through their active-patch list
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  int cg = kind == 1 ? C : C / groups;
  if (!((cg >= epc && cg % epc == 0) || (cg < epc && epc % cg == 0))) return -2;
  const VoxGeo g = mkvox(B, D, H, W, C, mask ? active_list : nullptr, n_active, bshift);
  if (g.nvox == 0) return 0;
  const int nv = 256 / (C / epc);
  const int nb = nblocks((g.nvox + nv - 1) / nv * 256);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    DISPATCH_T(dtype, AM_LAUNCH((voxel_norm_fwd_kernel<float, VN_GROUP>), dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, g, cg, gamma, beta, eps),
               AM_LAUNCH((voxel_norm_fwd_kernel<bf16_t, VN_GROUP>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, g, cg, gamma, beta, eps));
  else
    DISPATCH_T(dtype, AM_LAUNCH((voxel_norm_fwd_kernel<float, VN_GRN>), dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, g, cg, gamma, beta, eps),
               AM_LAUNCH((voxel_norm_fwd_kernel<bf16_t, VN_GRN>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, g, cg, gamma, beta, eps));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_voxel_norm_bwd(int dtype, int kind, const void* x, const void* dy, void* dx, int B, int D, int H, int W, int C, int groups,
                      const float* gamma, float eps, float* dgamma_accum, float* dbeta_accum, const uint8_t* mask, int bshift,
                      const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if (!gamma || groups < 1 || C % groups || (kind != 0 && kind != 1)) return -1;
  if (!list_ok(mask, active_list, n_active)) return -2;
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  int cg = kind == 1 ? C : C / groups;
  if (!((cg >= epc && cg % epc == 0) || (cg < epc && epc % cg == 0))) return -2;
  const VoxGeo g = mkvox(B, D, H, W, C, mask ? active_list : nullptr, n_active, bshift);
  if (g.nvox == 0) return 0;
  const int nv = 256 / (C / epc);
  long nbl = (g.nvox + nv - 1) / nv;
  const int nb = (int)(nbl > 1024 ? 1024 : nbl);                // one atomic flush per workgroup and channel
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    DISPATCH_T(dtype, AM_LAUNCH((voxel_norm_bwd_kernel<float, VN_GROUP>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, g, cg, gamma, eps, dgamma_accum, dbeta_accum),
               AM_LAUNCH((voxel_norm_bwd_kernel<bf16_t, VN_GROUP>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, g, cg, gamma, eps, dgamma_accum, dbeta_accum));
  else
    DISPATCH_T(dtype, AM_LAUNCH((voxel_norm_bwd_kernel<float, VN_GRN>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, g, cg, gamma, eps, dgamma_accum, dbeta_accum),
               AM_LAUNCH((voxel_norm_bwd_kernel<bf16_t, VN_GRN>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, g, cg, gamma, eps, dgamma_accum, dbeta_accum));
  AM_CHECK_LAUNCH();
  return 0;
}

// output extent of a torch pooling layer (floor, or ceil_mode: the last window must start inside the input or its left padding)
static int pool_out(int n, int k, int s, int p, int dil, int ceil_mode) {
  const int num = n + 2 * p - dil * (k - 1) - 1;
  if (num < 0) return 0;
  int o = (ceil_mode ? (num + s - 1) / s : num / s) + 1;
  if (ceil_mode && (o - 1) * s >= n + p) --o;
  return o;
}

int am_pool3d_fwd(int dtype, int op, const void* x, void* y, int32_t* argmax, int B, int Di, int Hi, int Wi, int C, int ksize, int stride,
                  int pad, int dilation, int count_include_pad, int Do, int Ho, int Wo, const uint8_t* mask, int in_bshift, int out_bshift,
                  int fd, int fh, int fw, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if ((op != 0 && op != 1) || ksize < 1 || ksize > 7 || stride < 1 || pad < 0 || 2 * pad > ksize || dilation < 1 || (op == 1 && dilation != 1)) return -2;
  {                                                        // (Do, Ho, Wo) = torch's floor OR ceil_mode extents, all three by the same rule
    bool okf = Do == pool_out(Di, ksize, stride, pad, dilation, 0) && Ho == pool_out(Hi, ksize, stride, pad, dilation, 0) && Wo == pool_out(Wi, ksize, stride, pad, dilation, 0);
    bool okc = Do == pool_out(Di, ksize, stride, pad, dilation, 1) && Ho == pool_out(Hi, ksize, stride, pad, dilation, 1) && Wo == pool_out(Wi, ksize, stride, pad, dilation, 1);
    if (!okf && !okc) return -2;
  }
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo go = mkvox(B, Do, Ho, Wo, C, mask ? active_list : nullptr, n_active, out_bshift);
  if (go.nvox == 0) return 0;
  const MaskView mi{mask, fd, fh, fw, in_bshift};
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int nb = nblocks(go.nvox * (C / epc));
  hipStream_t st = (hipStream_t)stream;
  if (op == 0)
    DISPATCH_T(dtype, AM_LAUNCH((pool3d_fwd_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, argmax, go, Di, Hi, Wi, ksize, stride, pad, dilation, count_include_pad, mi),
               AM_LAUNCH((pool3d_fwd_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, go, Di, Hi, Wi, ksize, stride, pad, dilation, count_include_pad, mi));
  else
    DISPATCH_T(dtype, AM_LAUNCH((pool3d_fwd_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, argmax, go, Di, Hi, Wi, ksize, stride, pad, dilation, count_include_pad, mi),
               AM_LAUNCH((pool3d_fwd_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, go, Di, Hi, Wi, ksize, stride, pad, dilation, count_include_pad, mi));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_pool3d_bwd(int dtype, int op, const void* dy, const int32_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int C, int ksize,
                  int stride, int pad, int dilation, int count_include_pad, int Do, int Ho, int Wo, const uint8_t* mask, int in_bshift,
                  int out_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if ((op != 0 && op != 1) || (op == 0 && !argmax) || ksize < 1 || ksize > 7 || stride < 1 || dilation < 1 || (op == 1 && dilation != 1)) return -2;
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo gi = mkvox(B, Di, Hi, Wi, C, mask ? active_list : nullptr, n_active, in_bshift);
  if (gi.nvox == 0) return 0;
  const MaskView mo{mask, fd, fh, fw, out_bshift};
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int nb = nblocks(gi.nvox * (C / epc));
  hipStream_t st = (hipStream_t)stream;
  if (op == 0)
    DISPATCH_T(dtype, AM_LAUNCH((pool3d_bwd_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, gi, Do, Ho, Wo, ksize, stride, pad, dilation, count_include_pad, mo),
               AM_LAUNCH((pool3d_bwd_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, gi, Do, Ho, Wo, ksize, stride, pad, dilation, count_include_pad, mo));
  else
    DISPATCH_T(dtype, AM_LAUNCH((pool3d_bwd_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, gi, Do, Ho, Wo, ksize, stride, pad, dilation, count_include_pad, mo),
               AM_LAUNCH((pool3d_bwd_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, gi, Do, Ho, Wo, ksize, stride, pad, dilation, count_include_pad, mo));
  AM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

namespace {
template <typename T, int K>
int launch_dw(const void* x, const float* w, const float* bias, void* y, int B, int D, int H, int W, int C, MaskView mv, int flip, hipStream_t st) {
  constexpr int EPC = TT<T>::EPC;
  auto kern = dwconv_kernel<T, K>;
  const size_t sm = (size_t)(DWD + K - 1) * (DWH + K - 1) * (DWW + K - 1) * 16 + (size_t)K * K * K * EPC * 4;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nbrick = B * ((D + DWD - 1) / DWD) * ((H + DWH - 1) / DWH) * ((W + DWW - 1) / DWW);
  AM_LAUNCH(kern, dim3((unsigned)((nbrick + 7) / 8 * 8 * (C / EPC))), dim3(256), sm, st, (const T*)x, w, bias, (T*)y, B, D, H, W, C, mv, flip);
  AM_CHECK_LAUNCH();
  return 0;
}
template <typename T, int K>
int launch_dw_wgrad(const void* x, const void* dy, float* dw, float* db, int B, int D, int H, int W, int C, MaskView mv, hipStream_t st) {
  constexpr int EPC = TT<T>::EPC;
  auto kern = dwconv_wgrad_kernel<T, K>;
  size_t sm = ((size_t)(DWD + K - 1) * (DWH + K - 1) * (DWW + K - 1) + (size_t)DWD * DWH * DWW) * 16;
  const size_t fold = (size_t)(4 * 64 * K * EPC + 4 * (64 / (K * K)) * EPC) * 4;
  if (sm < fold) sm = fold;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nbrick = B * ((D + DWD - 1) / DWD) * ((H + DWH - 1) / DWH) * ((W + DWW - 1) / DWW);
  int nslot = 1024 / (C / EPC) / 8 * 8;                       // ~four workgroups per CU over all channel chunks, a multiple of 8
  if (nslot < 8) nslot = 8;
  if (nslot > (nbrick + 7) / 8 * 8) nslot = (nbrick + 7) / 8 * 8;
  AM_LAUNCH(kern, dim3((unsigned)(nslot * (C / EPC))), dim3(256), sm, st, (const T*)x, (const T*)dy, dw, db, B, D, H, W, C, mv, nslot);
  AM_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" {

int am_dwconv3d(int dtype, int data_grad, const void* x, const float* w, const float* bias, void* y, int B, int D, int H, int W, int C,
                int ksize, const uint8_t* mask, int bshift, int fd, int fh, int fw, void* stream) {
  CHK_C(C);
  const MaskView mv{mask, fd, fh, fw, bshift};
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == AM_DT_BF16;
  switch (ksize) {
    case 3: return bf ? launch_dw<bf16_t, 3>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st) : launch_dw<float, 3>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st);
    case 5: return bf ? launch_dw<bf16_t, 5>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st) : launch_dw<float, 5>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st);
    case 7: return bf ? launch_dw<bf16_t, 7>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st) : launch_dw<float, 7>(x, w, bias, y, B, D, H, W, C, mv, data_grad, st);
    default: return -2;
  }
}

int am_dwconv3d_wgrad(int dtype, const void* x, const void* dy, float* dw_accum, float* db_accum, int B, int D, int H, int W, int C, int ksize,
                      const uint8_t* mask, int bshift, int fd, int fh, int fw, void* stream) {
  CHK_C(C);
  const MaskView mv{mask, fd, fh, fw, bshift};
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == AM_DT_BF16;
  switch (ksize) {
    case 3: return bf ? launch_dw_wgrad<bf16_t, 3>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st) : launch_dw_wgrad<float, 3>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st);
    case 5: return bf ? launch_dw_wgrad<bf16_t, 5>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st) : launch_dw_wgrad<float, 5>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st);
    case 7: return bf ? launch_dw_wgrad<bf16_t, 7>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st) : launch_dw_wgrad<float, 7>(x, dy, dw_accum, db_accum, B, D, H, W, C, mv, st);
    default: return -2;
  }
}

int am_dwconv3d_s2(int dtype, int data_grad, const void* src, const float* w, const float* bias, void* dst, int B, int Df, int Hf, int Wf,
                   int C, int ksize, const uint8_t* mask, int fine_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active,
                   void* stream) {
  CHK_C(C);
  if (Df % 2 || Hf % 2 || Wf % 2 || (mask && fine_bshift < 1)) return -2;
  if (!list_ok(mask, active_list, n_active)) return -2;
  const int Dc = Df / 2, Hc = Hf / 2, Wc = Wf / 2;
  // forward walks the active COARSE (output) voxels and reads the fine tensor; the data gradient walks the active FINE voxels and reads dy
  const VoxGeo g = data_grad ? mkvox(B, Df, Hf, Wf, C, mask ? active_list : nullptr, n_active, fine_bshift)
                             : mkvox(B, Dc, Hc, Wc, C, mask ? active_list : nullptr, n_active, fine_bshift - 1);
  if (g.nvox == 0) return 0;
  const MaskView ms{mask, fd, fh, fw, data_grad ? fine_bshift - 1 : fine_bshift};
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const dim3 grid(nblocks(g.nvox), C / epc);
  hipStream_t st = (hipStream_t)stream;
  const int sd = data_grad ? Dc : Df, sh = data_grad ? Hc : Hf, sw = data_grad ? Wc : Wf;
#define AM_S2(KK)                                                                                                                                   \
  if (data_grad) DISPATCH_T(dtype, AM_LAUNCH((dwconv_s2_kernel<float, KK, 1>), grid, dim3(256), 0, st, (const float*)src, w, bias, (float*)dst, g, sd, sh, sw, ms), \
                            AM_LAUNCH((dwconv_s2_kernel<bf16_t, KK, 1>), grid, dim3(256), 0, st, (const bf16_t*)src, w, bias, (bf16_t*)dst, g, sd, sh, sw, ms));     \
  else DISPATCH_T(dtype, AM_LAUNCH((dwconv_s2_kernel<float, KK, 0>), grid, dim3(256), 0, st, (const float*)src, w, bias, (float*)dst, g, sd, sh, sw, ms),            \
                  AM_LAUNCH((dwconv_s2_kernel<bf16_t, KK, 0>), grid, dim3(256), 0, st, (const bf16_t*)src, w, bias, (bf16_t*)dst, g, sd, sh, sw, ms))
  if (ksize == 3) { AM_S2(3); } else if (ksize == 5) { AM_S2(5); } else if (ksize == 7) { AM_S2(7); } else return -2;
#undef AM_S2
  AM_CHECK_LAUNCH();
  return 0;
}

int am_dwconv3d_s2_wgrad(int dtype, const void* x, const void* dy, float* dw_accum, float* db_accum, int B, int Df, int Hf, int Wf, int C,
                         int ksize, const uint8_t* mask, int fine_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active,
                         void* stream) {
  CHK_C(C);
  if (Df % 2 || Hf % 2 || Wf % 2 || (mask && fine_bshift < 1)) return -2;
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo go = mkvox(B, Df / 2, Hf / 2, Wf / 2, C, mask ? active_list : nullptr, n_active, fine_bshift - 1);
  if (go.nvox == 0) return 0;
  const MaskView mx{mask, fd, fh, fw, fine_bshift};
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  long per = 2048 / (C / epc); if (per < 1) per = 1; if (per > go.nvox) per = go.nvox;
  const dim3 grid((unsigned)per, C / epc);
  hipStream_t st = (hipStream_t)stream;
#define AM_S2W(KK)                                                                                                                                   \
  DISPATCH_T(dtype, AM_LAUNCH((dwconv_s2_wgrad_kernel<float, KK>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, dw_accum, db_accum, go, Df, Hf, Wf, mx), \
             AM_LAUNCH((dwconv_s2_wgrad_kernel<bf16_t, KK>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, dw_accum, db_accum, go, Df, Hf, Wf, mx))
  if (ksize == 3) { AM_S2W(3); } else if (ksize == 5) { AM_S2W(5); } else if (ksize == 7) { AM_S2W(7); } else return -2;
#undef AM_S2W
  AM_CHECK_LAUNCH();
  return 0;
}

int am_gelu(int dtype, const void* x, const void* dy, void* out, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift,
            const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo g = mkvox(B, D, H, W, C, mask ? active_list : nullptr, n_active, bshift);
  if (g.nvox == 0) return 0;
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int nb = nblocks(g.nvox * (C / epc));
  hipStream_t st = (hipStream_t)stream;
  if (dy)
    DISPATCH_T(dtype, AM_LAUNCH((gelu_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, g),
               AM_LAUNCH((gelu_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, g));
  else
    DISPATCH_T(dtype, AM_LAUNCH((gelu_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)nullptr, (float*)out, g),
               AM_LAUNCH((gelu_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)out, g));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_scale_residual(int dtype, int backward, const void* x, const void* other, const float* gamma, void* out, float* dgamma_accum, int B, int D,
                      int H, int W, int C, const uint8_t* mask, int bshift, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo g = mkvox(B, D, H, W, C, mask ? active_list : nullptr, n_active, bshift);
  if (g.nvox == 0) return 0;
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int nv = 256 / (C / epc);
  long nbl = (g.nvox + nv - 1) / nv;
  const int nb = (int)(nbl > 2048 ? 2048 : nbl);
  hipStream_t st = (hipStream_t)stream;
  if (backward)
    DISPATCH_T(dtype, AM_LAUNCH((scale_residual_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)other, gamma, (float*)out, dgamma_accum, g),
               AM_LAUNCH((scale_residual_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)other, gamma, (bf16_t*)out, dgamma_accum, g));
  else
    DISPATCH_T(dtype, AM_LAUNCH((scale_residual_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)other, gamma, (float*)out, dgamma_accum, g),
               AM_LAUNCH((scale_residual_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)other, gamma, (bf16_t*)out, dgamma_accum, g));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_masked_mean_fwd(int dtype, const void* x, float* mean, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                       int fw, void* stream) {
  CHK_C(C);
  const MaskView mv{mask, fd, fh, fw, bshift};
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, AM_LAUNCH(masked_mean_fwd_kernel<float>, dim3(B, C / epc), dim3(256), 0, st, (const float*)x, mean, B, D, H, W, C, mv),
             AM_LAUNCH(masked_mean_fwd_kernel<bf16_t>, dim3(B, C / epc), dim3(256), 0, st, (const bf16_t*)x, mean, B, D, H, W, C, mv));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_masked_mean_bwd(int dtype, const float* dmean, const float* count, void* dx, int B, int D, int H, int W, int C, const uint8_t* mask,
                       int bshift, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  if (!list_ok(mask, active_list, n_active)) return -2;
  const VoxGeo g = mkvox(B, D, H, W, C, mask ? active_list : nullptr, n_active, bshift);
  if (g.nvox == 0) return 0;
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int nb = nblocks(g.nvox * (C / epc));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, AM_LAUNCH(masked_mean_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, dmean, count, (float*)dx, g),
             AM_LAUNCH(masked_mean_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, dmean, count, (bf16_t*)dx, g));
  AM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
