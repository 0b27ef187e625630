// Step-level kernels of the AnatoMask iteration (gfx950): per-patch reconstruction loss (forward,
// scalar reduce, backward), reconstruction-guided hard-mask sampler, global grad-norm,
// fused clip + AdamW + teacher-EMA update.  All HBM / latency bound; VALU only.
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {   // 256 threads
  v = warp_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// One workgroup per 16^3 patch (P/AnatoMask.py:190-202 forward_loss; P/pretrain_AntoMask.py:423-425 teacher l2).
// normalized=1: target = (inp-mean)/sqrt(var_unbiased+1e-6) per patch; normalized=0: raw target.
// l2m[b][l] = mean_N((rec-target)^2) * (1-active[b][l])
__global__ __launch_bounds__(256) void patch_loss_fwd_kernel(const float* __restrict__ inp, const float* __restrict__ rec,
                                                             const uint8_t* __restrict__ active, int D, int H, int W, int fd,
                                                             int fh, int fw, int normalized, float* __restrict__ l2m,
                                                             float* __restrict__ pmean, float* __restrict__ prstd) {
  __shared__ float sh[4];
  const int L = fd * fh * fw;
  const int b = blockIdx.x / L, l = blockIdx.x % L;
  const int pw = l % fw, ph = (l / fw) % fh, pd = l / (fw * fh);
  const int z = threadIdx.x >> 4, y = threadIdx.x & 15;
  if (active[blockIdx.x] && !pmean) {                      // visible patch of a loss-only pass: contributes 0, its rec voxels may be unwritten
    if (threadIdx.x == 0) l2m[blockIdx.x] = 0.f;
    return;
  }
  const size_t base = (((size_t)b * D + pd * 16 + z) * H + ph * 16 + y) * W + pw * 16;
  float xi[16], xr[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a = *(const f32x4*)(inp + base + 4 * q), r = *(const f32x4*)(rec + base + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xi[4 * q + i] = a[i]; xr[4 * q + i] = r[i]; }
  }
  float mean = 0.f, rstd = 1.f;
  if (normalized) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += xi[i];
    mean = block_sum(s, sh) * (1.f / 4096.f);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = xi[i] - mean; s2 += d * d; }
    const float var = block_sum(s2, sh) * (1.f / 4095.f);
    rstd = 1.f / sqrtf(var + 1e-6f);
  }
  float e = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const float d = xr[i] - (xi[i] - mean) * rstd; e += d * d; }
  e = block_sum(e, sh) * (1.f / 4096.f);
  if (threadIdx.x == 0) {
    l2m[blockIdx.x] = active[blockIdx.x] ? 0.f : e;
    if (pmean) { pmean[blockIdx.x] = mean; prstd[blockIdx.x] = rstd; }
  }
}

// loss = sum(l2m) / (count(non-active) + 1e-8)   -> out[0] = loss, out[1] = 1/(count+1e-8)
__global__ void patch_loss_reduce_kernel(const float* __restrict__ l2m, const uint8_t* __restrict__ active, int n, float* out) {
  __shared__ float sh[4];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { s += l2m[i]; c += active[i] ? 0.f : 1.f; }
  s = block_sum(s, sh); c = block_sum(c, sh);
  if (threadIdx.x == 0) { out[0] = s / (c + 1e-8f); out[1] = 1.f / (c + 1e-8f); }
}

// drec = gout * (1-active) * inv_count * 2*(rec - target)/4096   (zeros on visible patches)
__global__ __launch_bounds__(256) void patch_loss_bwd_kernel(const float* __restrict__ inp, const float* __restrict__ rec,
                                                             const uint8_t* __restrict__ active, int D, int H, int W, int fd,
                                                             int fh, int fw, const float* __restrict__ pmean,
                                                             const float* __restrict__ prstd, const float* __restrict__ lossinfo,
                                                             const float* __restrict__ gout, float* __restrict__ drec) {
  const int L = fd * fh * fw;
  const int b = blockIdx.x / L, l = blockIdx.x % L;
  const int pw = l % fw, ph = (l / fw) % fh, pd = l / (fw * fh);
  const int z = threadIdx.x >> 4, y = threadIdx.x & 15;
  const size_t base = (((size_t)b * D + pd * 16 + z) * H + ph * 16 + y) * W + pw * 16;
  const bool vis = active[blockIdx.x] != 0;
  // a non-finite loss (P/pretrain_AntoMask.py:441-446 stops on it) must reach the optimizer's guard on EVERY rank: NaN gradients survive the
  // all-reduce, an overflowed loss with finite gradients would not -- so the gradient of a non-finite loss is NaN by construction
  const float lossv = lossinfo[0];
  const float k = vis ? 0.f : (lossv - lossv == 0.f ? (gout ? gout[0] : 1.f) * lossinfo[1] * (2.f / 4096.f) : __builtin_nanf(""));
  const float mean = pmean[blockIdx.x], rstd = prstd[blockIdx.x];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!vis) {
      const f32x4 a = *(const f32x4*)(inp + base + 4 * q), r = *(const f32x4*)(rec + base + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = k * (r[i] - (a[i] - mean) * rstd);
    }
    *(f32x4*)(drec + base + 4 * q) = o;
  }
}

// ------------------------------------------------------------------ hard-mask sampler (P/AnatoMask.py:81-128)
__device__ __forceinline__ uint32_t f2ord(float f) {          // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ void bitonic_sort(unsigned long long* a, int n) {   // n power of two, ascending, 256 threads
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = a[i], y = a[p];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[p] = x; }
        }
      }
    }
  __syncthreads();
}
// One workgroup per sample.  The len_loss highest-loss patches are never visible; among the others the
// len_keep smallest keys (ties broken by patch id, = stable argsort) are visible.
__global__ __launch_bounds__(256) void mask_sampler_kernel(const float* __restrict__ loss, const float* __restrict__ keys, int L,
                                                           int NP, int len_keep, int len_loss, uint8_t* __restrict__ mask) {
  extern __shared__ unsigned long long sk[];                  // NP sort keys, then NP bytes hard flags
  uint8_t* hard = (uint8_t*)(sk + NP);
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < NP; i += 256) {
    sk[i] = i < L ? (((unsigned long long)f2ord(loss[b * L + i])) << 32) | (unsigned)i : ~0ull;
    hard[i] = 0;
  }
  bitonic_sort(sk, NP);
  for (int i = threadIdx.x; i < L; i += 256)
    if (i >= L - len_loss) hard[(unsigned)(sk[i] & 0xffffffffu)] = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < NP; i += 256) {
    unsigned long long v = ~0ull;
    if (i < L) v = hard[i] ? ((0xfffffffeull << 32) | (unsigned)i) : ((((unsigned long long)f2ord(keys[b * L + i])) << 32) | (unsigned)i);
    sk[i] = v;
  }
  for (int i = threadIdx.x; i < L; i += 256) mask[b * L + i] = 0;
  bitonic_sort(sk, NP);
  for (int i = threadIdx.x; i < len_keep; i += 256) mask[b * L + (unsigned)(sk[i] & 0xffffffffu)] = 1;
}

// ------------------------------------------------------------------ optimizer side
__global__ void sumsq_kernel(const float* __restrict__ g, long n, double* out) {
  __shared__ double shd[4];
  double s = 0.0;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)g)[i];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 << 2; i < n; ++i) s += (double)g[i] * g[i];
  s = warp_sum_d(s);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, shd[0] + shd[1] + shd[2] + shd[3]);
}

// clip_grad_norm_(12) + torch.optim.AdamW + timm ModelEma.update in one pass over the flat parameter buffer
// (P/pretrain_AntoMask.py:437-440).  n % 4 == 0 (the flat buffer is padded).
__global__ void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 float* __restrict__ ema, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                 float bc2_sqrt, const double* __restrict__ sumsq, float max_norm, float ema_decay,
                                 float grad_scale, float* __restrict__ gnorm_out, const float* __restrict__ dyn, int* __restrict__ guard) {
  if (dyn) { lr = dyn[0]; bc1 = dyn[1]; bc2_sqrt = dyn[2]; ema_decay = dyn[3]; }   // hipGraph replays: the per-step scalars live in device memory
  // grad_scale: g holds the SUM of the ranks' gradients; the 1/world of DDP's mean is folded in here (norm and update see g*scale)
  const float total = sumsq ? (float)sqrt(sumsq[0]) * grad_scale : 0.f;
  float coef = sumsq ? max_norm / (total + 1e-6f) : 1.f;
  coef = (coef > 1.f ? 1.f : coef) * grad_scale;
  if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_out[0] = total;
  if (guard) {
    // per-step non-finite stop without a host round trip (P/pretrain_AntoMask.py:441-446 looks at loss.item() every step): a non-finite
    // gradient norm -- a non-finite loss makes it so, am_patch_loss_bwd -- leaves p / m / v / ema untouched and latches guard[0]; once
    // latched every later call is skipped too, so what the driver finds when it looks (once per epoch) is the state BEFORE the bad step.
    // guard[2] counts the calls; guard[1] = the 1-based index of the first bad one.  (guard[0] can only change in a call whose norm is not
    // finite, where every workgroup skips anyway: no ordering between the workgroups is needed.)
    const bool bad = !(total - total == 0.f);
    const bool latched = guard[0] != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int call = guard[2] + 1;
      guard[2] = call;
      if (bad && !latched) { guard[1] = call; guard[0] = 1; }
    }
    if (bad || latched) return;
  }
  const float step_size = lr / bc1, decay_mul = 1.f - lr * wd;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    f32x4 ee = ema ? ((f32x4*)ema)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * coef;
      float w = pp[k] * decay_mul;
      mm[k] = mm[k] + (gk - mm[k]) * (1.f - b1);
      vv[k] = vv[k] * b2 + (1.f - b2) * gk * gk;
      const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      w = w - step_size * (mm[k] / denom);
      pp[k] = w;
      ee[k] = ee[k] * ema_decay + (1.f - ema_decay) * w;
    }
    ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    if (ema) ((f32x4*)ema)[i] = ee;
  }
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float decay, const int* __restrict__ guard) {
  if (guard && guard[0]) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    ema[i] = ema[i] * decay + (1.f - decay) * p[i];
}

// timm's update on an integer state_dict entry (BatchNorm's num_batches_tracked): ema.copy_(ema * decay + (1 - decay) * model) --
// int64 * python float promotes to float32, the sum is float32, copy_ into int64 truncates toward zero
__global__ void ema_i64_kernel(long long* __restrict__ ema, const long long* __restrict__ p, int n, float decay, float one_minus, const int* __restrict__ guard) {
  if (guard && guard[0]) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) ema[i] = (long long)((float)ema[i] * decay + one_minus * (float)p[i]);
}

// guard latched: put the snapshot taken before the step back (BatchNorm running statistics / counters the student's forward updated)
__global__ void guard_restore_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ snap, long nwords, const int* __restrict__ guard) {
  if (!guard[0]) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long)gridDim.x * 256) dst[i] = snap[i];
}

}  // namespace

extern "C" {

int am_patch_loss_fwd(const float* inp, const float* rec, const uint8_t* active, int B, int D, int H, int W, int normalized,
                      float* l2m, float* pmean, float* prstd, float* lossinfo, void* stream) {
  if (D % 16 || H % 16 || W % 16) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int fd = D / 16, fh = H / 16, fw = W / 16;
  AM_LAUNCH(patch_loss_fwd_kernel, dim3(B * fd * fh * fw), dim3(256), 0, st, inp, rec, active, D, H, W, fd, fh, fw,
                     normalized, l2m, pmean, prstd);
  if (lossinfo) AM_LAUNCH(patch_loss_reduce_kernel, dim3(1), dim3(256), 0, st, l2m, active, B * fd * fh * fw, lossinfo);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_patch_loss_bwd(const float* inp, const float* rec, const uint8_t* active, int B, int D, int H, int W, const float* pmean,
                      const float* prstd, const float* lossinfo, const float* gout, float* drec, void* stream) {
  if (D % 16 || H % 16 || W % 16) return -1;
  const int fd = D / 16, fh = H / 16, fw = W / 16;
  AM_LAUNCH(patch_loss_bwd_kernel, dim3(B * fd * fh * fw), dim3(256), 0, (hipStream_t)stream, inp, rec, active, D, H, W,
                     fd, fh, fw, pmean, prstd, lossinfo, gout, drec);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_mask_sampler(const float* loss, const float* keys, int B, int L, int len_keep, int len_loss, uint8_t* mask, void* stream) {
  if (L > 4096 || len_keep > L || len_loss < 0 || len_loss + len_keep > L) return -1;
  int NP = 1; while (NP < L) NP <<= 1;
  AM_LAUNCH(mask_sampler_kernel, dim3(B), dim3(256), NP * 9, (hipStream_t)stream, loss, keys, L, NP, len_keep, len_loss, mask);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_sumsq(const float* g, long n, double* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  AM_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
  int nb = (int)((n / 4 + 255) / 256); if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
  AM_LAUNCH(sumsq_kernel, dim3(nb), dim3(256), 0, st, g, n, out);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, double lr, double beta1, double beta2, double eps,
                 double weight_decay, int step, const double* sumsq, double max_norm, double ema_decay, double grad_scale,
                 float* gnorm_out, const float* dyn_scalars, int* guard, void* stream) {
  if (n % 4) return -1;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);   // as torch: python doubles
  int nb = (int)((n / 4 + 255) / 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  AM_LAUNCH(adamw_ema_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, (float)lr, (float)beta1, (float)beta2,
                     (float)eps, (float)weight_decay, (float)bc1, (float)sqrt(bc2), sumsq, (float)max_norm, (float)ema_decay, (float)grad_scale, gnorm_out, dyn_scalars, guard);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_ema(float* ema, const float* p, long n, double decay, const int* guard, void* stream) {
  int nb = (int)((n + 255) / 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  AM_LAUNCH(ema_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, ema, p, n, (float)decay, guard);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_ema_i64(long* ema, const long* p, int n, double decay, const int* guard, void* stream) {
  if (n <= 0) return 0;
  AM_LAUNCH(ema_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)ema, (const long long*)p, n, (float)decay,
            (float)(1.0 - decay), guard);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_guard_restore(void* dst, const void* snapshot, long nbytes, const int* guard, void* stream) {
  if (nbytes % 4 || !guard) return -1;
  if (nbytes == 0) return 0;
  int nb = (int)((nbytes / 4 + 255) / 256); if (nb > 1024) nb = 1024;
  AM_LAUNCH(guard_restore_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (unsigned*)dst, (const unsigned*)snapshot, nbytes / 4, guard);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_version(void) { return 1; }

}  // extern "C"
