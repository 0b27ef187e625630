// Device-side spatial augmentation of the data feed (gfx950, VALU / HBM bound): what batchgenerators' SpatialTransform + MirrorTransform
// do on the host in the reference's worker processes (P/pretrain_AntoMask.py:57-150: rotation +-30 deg and isotropic scaling 0.7-1.4,
// p = 0.2 each, order-3 spline interpolation of the data, constant border 0; then mirroring of every axis with p = 0.5), applied to
// the ENLARGED patch the loader crops (nnunetv2/training/data_augmentation/compute_initial_patch_size.py:4-24):
//   am_spline_prefilter   cubic B-spline coefficients of a volume, in place, separable IIR with mirror boundaries -- exactly
//                         scipy.ndimage.spline_filter(order=3, mode='mirror'), which is what map_coordinates(order=3, mode='constant')
//                         applies before interpolating (batchgenerators.augmentations.utils.interpolate_img calls map_coordinates)
//   am_resample_affine    out[o] = interp(src, A * (o, 1)): per-sample 3x4 affine (rotation, scale, centring, mirror flips folded
//                         in); order 0 = integer crop / flip, 1 = trilinear, 3 = cubic B-spline over prefiltered coefficients with
//                         scipy's rules (outside [0, n-1] -> cval, mirrored coefficient access at the edges)
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

// one thread per line along `axis`; lines are enumerated so that consecutive threads take consecutive innermost positions
// (axis 0 / 1: coalesced across the wave; axis 2: each thread walks its own contiguous line, served by L2)
__global__ __launch_bounds__(256) void spline_prefilter_kernel(float* __restrict__ v, int D, int H, int W, int axis) {
  const long nline = axis == 0 ? (long)H * W : axis == 1 ? (long)D * W : (long)D * H;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= nline) return;
  const int n = axis == 0 ? D : axis == 1 ? H : W;
  long base, stride;
  if (axis == 0) { base = t; stride = (long)H * W; }
  else if (axis == 1) { base = (t / W) * (long)H * W + t % W; stride = W; }
  else { base = t * (long)W; stride = 1; }
  if (n < 2) return;
  const double z = -0.26794919243112270647;                  // sqrt(3) - 2
  const float zf = (float)z, lambda = (float)((1.0 - z) * (1.0 - 1.0 / z));
  float* p = v + base;
  // causal initialisation for a mirror boundary (exact sum over the 2n - 2 periodic samples; terms vanish below fp32 after ~24)
  double zi = z, c0 = (double)p[0] * lambda;
  const int m = n < 40 ? n : 40;
  for (int i = 1; i < m; ++i) { c0 += zi * (double)p[i * stride] * lambda; zi *= z; }
  if (m == n) {
    for (int i = n - 2; i > 0; --i) { c0 += zi * (double)p[i * stride] * lambda; zi *= z; }
    c0 /= (1.0 - zi);
  }
  float prev = (float)c0;
  p[0] = prev;
  for (int i = 1; i < n; ++i) { prev = p[i * stride] * lambda + zf * prev; p[i * stride] = prev; }
  prev = (float)((z / (z * z - 1.0)) * ((double)p[(n - 1) * stride] + z * (double)p[(n - 2) * stride]));
  p[(n - 1) * stride] = prev;
  for (int i = n - 2; i >= 0; --i) { prev = zf * (prev - p[i * stride]); p[i * stride] = prev; }
}

__device__ __forceinline__ int mirror_idx(int i, int n) {
  i = i < 0 ? -i : i;
  i = i > n - 1 ? 2 * (n - 1) - i : i;
  return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
__device__ __forceinline__ void bspline_w(float f, float* w) {   // weights of the 4 coefficients around floor(x), f = x - floor(x)
  const float f2 = f * f, f3 = f2 * f, g = 1.f - f;
  w[0] = g * g * g * (1.f / 6.f);
  w[1] = 2.f / 3.f - f2 + 0.5f * f3;
  w[2] = 2.f / 3.f - g * g + 0.5f * g * g * g;
  w[3] = f3 * (1.f / 6.f);
}

struct Aff { float a[12]; };

__global__ __launch_bounds__(256) void resample_affine_kernel(const float* __restrict__ src, int Ds, int Hs, int Ws, float* __restrict__ dst,
                                                              int D, int H, int W, Aff A, int order, float cval) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)D * H * W) return;
  const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((long)W * H));
  const float pz = A.a[0] * d + A.a[1] * h + A.a[2] * w + A.a[3];
  const float py = A.a[4] * d + A.a[5] * h + A.a[6] * w + A.a[7];
  const float px = A.a[8] * d + A.a[9] * h + A.a[10] * w + A.a[11];
  float out = cval;
  if (order == 0) {
    const int iz = (int)floorf(pz + 0.5f), iy = (int)floorf(py + 0.5f), ix = (int)floorf(px + 0.5f);
    if ((unsigned)iz < (unsigned)Ds && (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws) out = src[((long)iz * Hs + iy) * Ws + ix];
  } else if (pz >= 0.f && pz <= Ds - 1.f && py >= 0.f && py <= Hs - 1.f && px >= 0.f && px <= Ws - 1.f) {   // scipy: outside [0, n-1] -> cval
    const int bz = (int)floorf(pz), by = (int)floorf(py), bx = (int)floorf(px);
    if (order == 1) {
      const float fz = pz - bz, fy = py - by, fx = px - bx;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = min(bz + (k >> 2), Ds - 1), iy = min(by + ((k >> 1) & 1), Hs - 1), ix = min(bx + (k & 1), Ws - 1);
        const float wt = ((k >> 2) ? fz : 1.f - fz) * (((k >> 1) & 1) ? fy : 1.f - fy) * ((k & 1) ? fx : 1.f - fx);
        s += wt * src[((long)iz * Hs + iy) * Ws + ix];
      }
      out = s;
    } else {
      float wz[4], wy[4], wx[4];
      bspline_w(pz - bz, wz); bspline_w(py - by, wy); bspline_w(px - bx, wx);
      int ix[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ix[c] = mirror_idx(bx - 1 + c, Ws);
      float s = 0.f;
#pragma unroll
      for (int a_ = 0; a_ < 4; ++a_) {
        const long zo = (long)mirror_idx(bz - 1 + a_, Ds) * Hs;
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) {
          const float* row = src + (zo + mirror_idx(by - 1 + b_, Hs)) * Ws;
          const float r = wx[0] * row[ix[0]] + wx[1] * row[ix[1]] + wx[2] * row[ix[2]] + wx[3] * row[ix[3]];
          s += wz[a_] * wy[b_] * r;
        }
      }
      out = s;
    }
  }
  dst[i] = out;
}

}  // namespace

extern "C" {

int am_spline_prefilter(float* vol, int D, int H, int W, void* stream) {
  if (D < 1 || H < 1 || W < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  for (int axis = 0; axis < 3; ++axis) {
    const long nline = axis == 0 ? (long)H * W : axis == 1 ? (long)D * W : (long)D * H;
    AM_LAUNCH(spline_prefilter_kernel, dim3((unsigned)((nline + 255) / 256)), dim3(256), 0, st, vol, D, H, W, axis);
    AM_CHECK_LAUNCH();
  }
  return 0;
}

int am_resample_affine(const float* src, int Ds, int Hs, int Ws, float* dst, int D, int H, int W, const float* affine_host12, int order,
                       float cval, void* stream) {
  if (order != 0 && order != 1 && order != 3) return -2;
  Aff A;
  for (int i = 0; i < 12; ++i) A.a[i] = affine_host12[i];
  const long n = (long)D * H * W;
  AM_LAUNCH(resample_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, Ds, Hs, Ws, dst, D, H, W, A,
            order, cval);
  AM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
