// HBM-bound streaming kernels of the AnatoMask step (gfx950, VALU only): pooled sparse InstanceNorm /
// BatchNorm statistics + fused apply (+LeakyReLU / ReLU6 / residual / mask-token fill), their
// backward passes, the Cin=1 stem convolutions, the 1x1 projection, weight (un)packing.
// Every kernel walks channels-last rows with one 16-byte chunk per lane (coalesced: consecutive
// lanes take consecutive chunks of consecutive voxels) and touches only voxels of active patches.
#include <mutex>
#include <type_traits>
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

// MFMA tile row -> output channel (same permutation as the convolution kernels, conv_plan.h): a lane's registers of tiles (2h, 2h+1)
// are 8 consecutive channels
__device__ __forceinline__ int crow(int R) { return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3); }

constexpr int NREP = 8;      // replicated reduction accumulators (spreads same-address atomic contention)

struct Geo {                 // a channels-last tensor [B][D][H][W][C] and its (optional) patch mask
  int B, D, H, W, C;
  int vpw;                   // voxels per workgroup (host picks it so that every launch has ~1-2k workgroups)
  unsigned mW, mH, mD;       // floor(2^32/d)+1: exact floor(v/d) by __umulhi for v*d < 2^32
  MaskView mask;
  __device__ __forceinline__ bool active(long v) const {
    if (!mask.m) return true;
    const unsigned u = (unsigned)v;
    const unsigned t1 = __umulhi(u, mW); const int w = u - t1 * W;
    const unsigned t2 = __umulhi(t1, mH); const int h = t1 - t2 * H;
    const unsigned b = __umulhi(t2, mD); const int d = t2 - b * D;
    return mask.active((int)b, d, h, w);
  }
  __device__ __forceinline__ void decode(long v, int& b, int& d, int& h, int& w) const {
    const unsigned u = (unsigned)v;
    const unsigned t1 = __umulhi(u, mW); w = u - t1 * W;
    const unsigned t2 = __umulhi(t1, mH); h = t1 - t2 * H;
    const unsigned bb = __umulhi(t2, mD); d = t2 - bb * D; b = (int)bb;
  }
  __device__ __forceinline__ long nvox() const { return (long)B * D * H * W; }
};

// thread -> (voxel lane, chunk lane); CPV chunks per voxel; VPP voxels per pass.  nthr = threads of the workgroup: 256, or 512
// for rows of more than 256 chunks (fp32 storage with C > 1024: STUNet-H's 1536 channels) -- cpv <= nthr is what the kernels need.
template <typename T> struct Walk {
  int cpv, vpp, cl, vl; bool live;
  __device__ __forceinline__ Walk(int C, int nthr = 256) {
    cpv = C / TT<T>::EPC; vpp = nthr / cpv; cl = threadIdx.x % cpv; vl = threadIdx.x / cpv; live = vl < vpp;
  }
};

// ------------------------------------------------------------------ statistics (forward)
// sums[c][0] += sum x, sums[c][1] += sum x^2 over active voxels (double atomics, one per WG per channel)
template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void chan_stats_kernel(const T* __restrict__ x, Geo g, double* __restrict__ sums) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[NT * 2 * 8];
  Walk<T> wk(g.C, NT);
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) s1[i] = s2[i] = 0.f;
  const long v0 = (long)blockIdx.x * g.vpw, v1 = min(v0 + (long)g.vpw, g.nvox());
  if (wk.live)
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      if (!g.active(v)) continue;
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + v * g.C + wk.cl * EPC), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) { s1[i] += f[i]; s2[i] += f[i] * f[i]; }
    }
#pragma unroll
  for (int i = 0; i < EPC; ++i) { red[(threadIdx.x * 2) * 8 + i] = s1[i]; red[(threadIdx.x * 2 + 1) * 8 + i] = s2[i]; }
  __syncthreads();
  if (threadIdx.x < wk.cpv) {                      // one thread per chunk lane folds the voxel lanes
    double a1[EPC], a2[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = a2[i] = 0.0;
    for (int vl = 0; vl < wk.vpp; ++vl) {
      const int t = vl * wk.cpv + threadIdx.x;
#pragma unroll
      for (int i = 0; i < EPC; ++i) { a1[i] += red[(t * 2) * 8 + i]; a2[i] += red[(t * 2 + 1) * 8 + i]; }
    }
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int c = threadIdx.x * EPC + i;
      double* sr = sums + (size_t)(blockIdx.x % NREP) * g.C * 2;
      atomicAdd(&sr[c * 2], a1[i]); atomicAdd(&sr[c * 2 + 1], a2[i]);
    }
  }
}

// count of active voxels implied by the mask (dense: B*D*H*W)
__global__ void mask_count_kernel(const uint8_t* mask, int n, int voxels_per_patch, double* out) {
  __shared__ double part[4];
  double c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += mask[i] ? 1.0 : 0.0;
  c = warp_sum_d(c);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (part[0] + part[1] + part[2] + part[3]) * voxels_per_patch;
}

// mean / rstd / folded scale+shift; optional BatchNorm running-stat update (momentum, unbiased var)
__global__ void norm_finalize_kernel(const double* sums, int nrep, const double* count_ptr, double count_host, int C,
                                     const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                     float* scale, float* shift, float* run_mean, float* run_var, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = count_ptr ? count_ptr[0] : count_host;
  double q1 = 0.0, q2 = 0.0;
  for (int r = 0; r < nrep; ++r) { q1 += sums[((size_t)r * C + c) * 2]; q2 += sums[((size_t)r * C + c) * 2 + 1]; }
  const double m = q1 / n;
  double var = q2 / n - m * m;
  if (var < 0) var = 0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m; rstd[c] = rs;
  const float sc = gamma[c] * rs;
  scale[c] = sc; shift[c] = beta[c] - (float)m * sc;
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (n / (n - 1.0)));
  }
}

// eval-mode BatchNorm: fold running stats
__global__ void norm_fold_running_kernel(int C, const float* gamma, const float* beta, const float* run_mean,
                                         const float* run_var, float eps, float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(run_var[c] + eps);
  scale[c] = sc; shift[c] = beta[c] - run_mean[c] * sc;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == AM_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  if (act == AM_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}
__device__ __forceinline__ float act_grad(float out, int act) {   // derivative from the SAVED OUTPUT
  if (act == AM_ACT_LRELU) return out > 0.f ? 1.f : 0.01f;
  if (act == AM_ACT_RELU6) return (out > 0.f && out < 6.f) ? 1.f : 0.f;
  return 1.f;
}

__device__ __forceinline__ float act_grad_pre(float pre, int act) {   // derivative from the RECOMPUTED pre-activation
  if (act == AM_ACT_LRELU) return pre > 0.f ? 1.f : 0.01f;
  if (act == AM_ACT_RELU6) return (pre > 0.f && pre < 6.f) ? 1.f : 0.f;
  return 1.f;
}

// ------------------------------------------------------------------ apply (forward)
// y = act(x*scale + shift [+ res | + stem 1x1 shortcut]) on active voxels;
// fill != nullptr: inactive voxels get the mask token (densify, P/AnatoMask.py:160-163), output dense.
template <typename T, int NT = 256, int ACT_ = -1>
__global__ __launch_bounds__(NT) void norm_apply_kernel(const T* __restrict__ x, Geo g, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, const T* __restrict__ res,
                                                         const float* __restrict__ stem_x, const float* __restrict__ stem_w,
                                                         const float* __restrict__ stem_b, const float* __restrict__ fill,
                                                         T* __restrict__ y) {
  constexpr int EPC = TT<T>::EPC;
  Walk<T> wk(g.C, NT);
  if (!wk.live) return;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) { sc[i] = scale[wk.cl * EPC + i]; sh[i] = shift[wk.cl * EPC + i]; }
  const long v0 = (long)blockIdx.x * g.vpw, v1 = min(v0 + (long)g.vpw, g.nvox());
  for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
    const size_t off = (size_t)v * g.C + wk.cl * EPC;
    float o[EPC];
    if (g.active(v)) {
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + off), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) o[i] = f[i] * sc[i] + sh[i];
      if (res) {
        float r[EPC];
        chunk_to_f<T>(*(const u32x4*)(res + off), r);
#pragma unroll
        for (int i = 0; i < EPC; ++i) o[i] += r[i];
      }
      if (stem_x) {
        const float xv = stem_x[v];
#pragma unroll
        for (int i = 0; i < EPC; ++i) o[i] += stem_w[wk.cl * EPC + i] * xv + stem_b[wk.cl * EPC + i];
      }
#pragma unroll
      for (int i = 0; i < EPC; ++i) o[i] = act_fwd(o[i], ACT_ >= 0 ? ACT_ : act);
    } else if (fill) {
#pragma unroll
      for (int i = 0; i < EPC; ++i) o[i] = fill[wk.cl * EPC + i];
    } else {
      continue;
    }
    *(u32x4*)(y + off) = f_to_chunk<T>(o);
  }
}


// ---- fused tails of the backward kernels (last-workgroup pattern, as partials_finalize_kernel) --------------------------------
// The tiny finalize / replica-fold kernels that used to follow the backward reduce / apply passes each waited ~130 us for a CU slot
// whenever the weight-gradient kernels of the side stream owned the chip (profiles/r02: 22 + 10 such waits per step on the critical
// chain).  Now the LAST workgroup of the streaming kernel does that work itself; the accumulators are device-scope atomics on a
// workspace that is zero on entry and left zero on exit (no memset launch either).
struct BwdFin {
  unsigned* ticket;                  // behind the accumulators; nullptr: no fused tail
  const double* count_ptr; double count_host;
  const float* gamma; const float* rstd;
  float *k0, *k1, *k2, *dgamma, *dbeta, *dtoken, *dbeta2;
};

__device__ __forceinline__ bool last_workgroup(unsigned* ticket) {
  __shared__ unsigned tk;
  // The accumulators are only ever touched by device-scope atomics, which execute at the memory side (MI355X_MICROARCH.md "Global
  // float atomics"): what the ticket must follow is their ACKNOWLEDGEMENT (vmcnt), not an L2 write-back.  __threadfence() here
  // (buffer_wbl2 + invalidate in all 256 threads of ~2000 workgroups, flushing the kernel's own streaming stores) cost 3.5 ms/step.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) tk = atomicAdd(ticket, 1u);
  __syncthreads();
  return tk == gridDim.x - 1;
}

// bsum [NREP][C][3] -> k0/k1/k2 + parameter gradients (norm_bwd_finalize_kernel's arithmetic), accumulators re-zeroed
__device__ __forceinline__ void bwd_reduce_tail(double* bsum, int C, const BwdFin& f) {
  if (!f.ticket) return;
  if (!last_workgroup(f.ticket)) return;
  const double n = f.count_ptr ? f.count_ptr[0] : f.count_host;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double b1 = 0.0, b2 = 0.0, b3 = 0.0;
    for (int r = 0; r < NREP; ++r) {
      double* q = bsum + ((size_t)r * C + c) * 3;
      b1 += __longlong_as_double(atomicExch((unsigned long long*)&q[0], 0ull));
      b2 += __longlong_as_double(atomicExch((unsigned long long*)&q[1], 0ull));
      b3 += __longlong_as_double(atomicExch((unsigned long long*)&q[2], 0ull));
    }
    const float gr = f.gamma[c] * f.rstd[c];
    f.k0[c] = gr; f.k1[c] = gr * (float)(b1 / n); f.k2[c] = gr * (float)(b2 / n);
    if (f.dgamma) f.dgamma[c] += (float)b2;
    if (f.dbeta) f.dbeta[c] += (float)b1;
    if (f.dbeta2) f.dbeta2[c] += (float)b1;
    if (f.dtoken) f.dtoken[c] += (float)b3;
  }
  if (threadIdx.x == 0) atomicExch(f.ticket, 0u);
}

// dxsum replicas [nrep][C] -> accum[c] += sum, replicas re-zeroed
// The replicas hold FIXED-POINT sums (int64, 2^-36 units): these bias gradients are what is left of a sum that cancels analytically
// (a conv bias under a norm), so even fp64 atomics leave the arrival order visible in the fp32 result (1e-16 of the partial sums
// is 1e-7 of the remainder); integer addition is associative -- the result is the same bits whatever the order.
constexpr double AM_DX_FIX = 68719476736.0;         // 2^36: resolution 1.5e-11, range +-1.3e8
__device__ __forceinline__ void dxsum_tail(double* rep, int nrep, int C, float* accum, unsigned* ticket) {
  if (!ticket) return;
  if (!last_workgroup(ticket)) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    long long s_ = 0;
    for (int r = 0; r < nrep; ++r) s_ += (long long)atomicExch((unsigned long long*)&rep[(size_t)r * C + c], 0ull);
    accum[c] += (float)((double)s_ / AM_DX_FIX);
  }
  if (threadIdx.x == 0) atomicExch(ticket, 0u);
}

// ------------------------------------------------------------------ backward: reduce
// dpre = dout * act'(out);  bsum[c] = {sum dpre, sum dpre*xhat, sum_{inactive} dout (token grad)}
template <typename T, int NT = 256, int ACT_ = -1, int OUT_ = 0>
__global__ __launch_bounds__(NT) void norm_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                              const T* __restrict__ x, Geo g, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int act, int fill,
                                                              double* __restrict__ bsum, const float* __restrict__ psc,
                                                              const float* __restrict__ psh, BwdFin fin) {
  constexpr int EPC = TT<T>::EPC;
  const int actv = ACT_ >= 0 ? ACT_ : act;                     // compile-time activation / derivative source in the specialised instantiations:
  const bool has_out = ACT_ >= 0 ? OUT_ != 0 : out != nullptr;  // the per-element code is then branch-free (the generic form compiled to scalar branches per element)
  __shared__ float red[NT * 3 * 8];
  Walk<T> wk(g.C, NT);
  float s1[EPC], s2[EPC], s3[EPC], mu[EPC], rs[EPC], qs[EPC], qh[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) { s1[i] = s2[i] = s3[i] = 0.f; mu[i] = rs[i] = qs[i] = qh[i] = 0.f; }
  if (wk.live) {
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      mu[i] = mean[wk.cl * EPC + i]; rs[i] = rstd[wk.cl * EPC + i];
      if (!has_out && actv != AM_ACT_NONE) { qs[i] = psc[wk.cl * EPC + i]; qh[i] = psh[wk.cl * EPC + i]; }
    }
    const long v0 = (long)blockIdx.x * g.vpw, v1 = min(v0 + (long)g.vpw, g.nvox());
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      const size_t off = (size_t)v * g.C + wk.cl * EPC;
      const bool a = g.active(v);
      if (!a && !fill) continue;
      float d[EPC];
      chunk_to_f<T>(*(const u32x4*)(dout + off), d);
      if (!a) {
#pragma unroll
        for (int i = 0; i < EPC; ++i) s3[i] += d[i];
        continue;
      }
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + off), f);
      if (actv != AM_ACT_NONE) {
        if (has_out) {
          float o[EPC];
          chunk_to_f<T>(*(const u32x4*)(out + off), o);
#pragma unroll
          for (int i = 0; i < EPC; ++i) d[i] *= act_grad(o[i], actv);
        } else {                                     // same expression as norm_apply_kernel's forward
#pragma unroll
          for (int i = 0; i < EPC; ++i) d[i] *= act_grad_pre(f[i] * qs[i] + qh[i], actv);
        }
      }
#pragma unroll
      for (int i = 0; i < EPC; ++i) { s1[i] += d[i]; s2[i] += d[i] * (f[i] - mu[i]) * rs[i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    red[(threadIdx.x * 3) * 8 + i] = s1[i]; red[(threadIdx.x * 3 + 1) * 8 + i] = s2[i]; red[(threadIdx.x * 3 + 2) * 8 + i] = s3[i];
  }
  __syncthreads();
  if (threadIdx.x < wk.cpv) {
    double a1[EPC], a2[EPC], a3[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = a2[i] = a3[i] = 0.0;
    for (int vl = 0; vl < wk.vpp; ++vl) {
      const int t = vl * wk.cpv + threadIdx.x;
#pragma unroll
      for (int i = 0; i < EPC; ++i) { a1[i] += red[(t * 3) * 8 + i]; a2[i] += red[(t * 3 + 1) * 8 + i]; a3[i] += red[(t * 3 + 2) * 8 + i]; }
    }
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int c = threadIdx.x * EPC + i;
      double* br = bsum + (size_t)(blockIdx.x % NREP) * g.C * 3;
      atomicAdd(&br[c * 3], a1[i]); atomicAdd(&br[c * 3 + 1], a2[i]);
      if (fill) atomicAdd(&br[c * 3 + 2], a3[i]);
    }
  }
  bwd_reduce_tail(bsum, g.C, fin);
}

// per-channel coefficients of the apply pass + parameter gradients (accumulated into fp32 grads)
__global__ void norm_bwd_finalize_kernel(const double* bsum, const double* count_ptr, double count_host, int C,
                                         const float* gamma, const float* rstd, float* k0, float* k1, float* k2,
                                         float* dgamma, float* dbeta, float* dtoken, float* dbeta2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = count_ptr ? count_ptr[0] : count_host;
  double b1 = 0.0, b2 = 0.0, b3 = 0.0;
  for (int r = 0; r < NREP; ++r) { const double* q = bsum + ((size_t)r * C + c) * 3; b1 += q[0]; b2 += q[1]; b3 += q[2]; }
  const float gr = gamma[c] * rstd[c];
  k0[c] = gr;                                  // dx = k0*dpre - k1 - k2*xhat
  k1[c] = gr * (float)(b1 / n);
  k2[c] = gr * (float)(b2 / n);
  if (dgamma) dgamma[c] += (float)b2;
  if (dbeta) dbeta[c] += (float)b1;
  if (dbeta2) dbeta2[c] += (float)b1;             // bias of a conv added AFTER the norm (1x1 shortcut): same sum
  if (dtoken) dtoken[c] += (float)b3;
}

// ------------------------------------------------------------------ backward: apply
// dx = k0*dpre - k1 - k2*xhat on active voxels; optionally store dpre (gradient of the residual branch)
template <typename T, int NT = 256, int ACT_ = -1, int OUT_ = 0>
__global__ __launch_bounds__(NT) void norm_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                             const T* __restrict__ x, Geo g, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ k0,
                                                             const float* __restrict__ k1, const float* __restrict__ k2, int act,
                                                             T* __restrict__ dx, T* __restrict__ dres, double* __restrict__ dxsum,
                                                             int dxrep, const float* __restrict__ psc, const float* __restrict__ psh,
                                                             float* __restrict__ dx_accum, unsigned* dx_ticket) {
  constexpr int EPC = TT<T>::EPC;
  const int actv = ACT_ >= 0 ? ACT_ : act;                     // compile-time activation / derivative source in the specialised instantiations:
  const bool has_out = ACT_ >= 0 ? OUT_ != 0 : out != nullptr;  // the per-element code is then branch-free (the generic form compiled to scalar branches per element)
  __shared__ float red[NT * 8];
  Walk<T> wk(g.C, NT);
  float mu[EPC], rs[EPC], c0[EPC], c1[EPC], c2[EPC], sx[EPC], qs[EPC], qh[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    const int c = wk.cl * EPC + i;
    sx[i] = 0.f; qs[i] = qh[i] = 0.f;
    if (wk.live) {
      mu[i] = mean[c]; rs[i] = rstd[c]; c0[i] = k0[c]; c1[i] = k1[c]; c2[i] = k2[c];
      if (!has_out && actv != AM_ACT_NONE) { qs[i] = psc[c]; qh[i] = psh[c]; }
    }
  }
  const long v0 = (long)blockIdx.x * g.vpw, v1 = min(v0 + (long)g.vpw, g.nvox());
  if (wk.live)
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      if (!g.active(v)) continue;
      const size_t off = (size_t)v * g.C + wk.cl * EPC;
      float d[EPC], f[EPC];
      chunk_to_f<T>(*(const u32x4*)(dout + off), d);
      chunk_to_f<T>(*(const u32x4*)(x + off), f);
      if (actv != AM_ACT_NONE) {
        if (has_out) {
          float o[EPC];
          chunk_to_f<T>(*(const u32x4*)(out + off), o);
#pragma unroll
          for (int i = 0; i < EPC; ++i) d[i] *= act_grad(o[i], actv);
        } else {
#pragma unroll
          for (int i = 0; i < EPC; ++i) d[i] *= act_grad_pre(f[i] * qs[i] + qh[i], actv);
        }
      }
      if (dres) *(u32x4*)(dres + off) = f_to_chunk<T>(d);
      float r[EPC];
#pragma unroll
      for (int i = 0; i < EPC; ++i) r[i] = c0[i] * d[i] - c1[i] - c2[i] * (f[i] - mu[i]) * rs[i];
      const u32x4 pk = f_to_chunk<T>(r);
      *(u32x4*)(dx + off) = pk;
      if (dxsum) {                                 // bias gradient of the conv that feeds this norm = sum of the STORED dx
        float q[EPC];
        chunk_to_f<T>(pk, q);
#pragma unroll
        for (int i = 0; i < EPC; ++i) sx[i] += q[i];
      }
    }
  if (dxsum) {                                     // (uniform) fold voxel lanes, one float atomic per channel per workgroup
#pragma unroll
    for (int i = 0; i < EPC; ++i) red[threadIdx.x * 8 + i] = sx[i];
    __syncthreads();
    if (threadIdx.x < wk.cpv) {
      float a1[EPC];
#pragma unroll
      for (int i = 0; i < EPC; ++i) a1[i] = 0.f;
      for (int vl = 0; vl < wk.vpp; ++vl)
#pragma unroll
        for (int i = 0; i < EPC; ++i) a1[i] += red[(vl * wk.cpv + threadIdx.x) * 8 + i];
#pragma unroll
      for (int i = 0; i < EPC; ++i)
        atomicAdd((unsigned long long*)&dxsum[(size_t)(blockIdx.x % dxrep) * g.C + threadIdx.x * EPC + i], (unsigned long long)__double2ll_rn((double)a1[i] * AM_DX_FIX));
    }
    dxsum_tail(dxsum, dxrep, g.C, dx_accum, dx_ticket);
  }
}


// ================================================================== block-sparse tensors: ACTIVE-PATCH ROW WALK
// A block-sparse tensor [B][D][H][W][C] with patch edge P = 1 << bs voxels is a set of "patch rows": P consecutive voxels along
// W = P*C contiguous elements (for STUNet-B exactly 1 KB at every level: 16 x 32 ch ... 1 x 512 ch, bf16).  The kernels below
// walk ONLY the rows of active patches, taken from the compacted active-patch list (am_mask_compact: one int32 per active
// patch, b << 24 | pd << 16 | ph << 8 | pw): no per-voxel mask lookup, no integer divisions, no loop iterations spent on the
// 60 % of the volume that is masked, and every thread keeps UNR independent 16-byte loads in flight.
// Thread t of a workgroup owns chunk (t % rowchunks) of row lane (t / rowchunks); its channel chunk is t % cpv (rowchunks is a
// multiple of cpv), the same thread -> channel map as the linear kernels, so the reduction epilogues are shared.
struct RowGeo {
  int B, D, H, W, C, bs;
  int rowchunks;             // 16-byte chunks per patch row = P * C / EPC  (<= 256)
  int rpp;                   // rows per pass = blockDim.x / rowchunks
  int rpw;                   // rows per workgroup (multiple of rpp)
  long total_rows;           // n_active * P * P
  const int* plist;
};
constexpr int UNR = 4;

template <typename T> struct RowWalk {
  int cl, vx, r0, cpv; bool live;
  long row, row_end; int rpp;
  __device__ __forceinline__ RowWalk(const RowGeo& g) {
    cpv = g.C / TT<T>::EPC;
    const int c = threadIdx.x % g.rowchunks;
    r0 = threadIdx.x / g.rowchunks; cl = c % cpv; vx = c / cpv; rpp = g.rpp; live = r0 < g.rpp;
    row = (long)blockIdx.x * g.rpw + r0;
    row_end = min((long)(blockIdx.x + 1) * g.rpw, g.total_rows);
  }
  // voxel index of this thread's chunk in patch row `rw`
  __device__ __forceinline__ long voxel(const RowGeo& g, long rw) const {
    const int P = 1 << g.bs;
    const int a = (int)(rw >> (2 * g.bs)), r = (int)(rw & (P * P - 1));
    const int pk = g.plist[a];
    const int b = (pk >> 24) & 255, pd = (pk >> 16) & 255, ph = (pk >> 8) & 255, pw = pk & 255;
    return (((long)b * g.D + pd * P + (r >> g.bs)) * g.H + ph * P + (r & (P - 1))) * g.W + pw * P + vx;
  }
};

// MODE: 0 plain, 1 + residual tensor, 2 + Cin=1 stem shortcut.  Straight-line body (compile-time activation / mode, tail rows
// clamped to the last row -- a pure map may store a row twice), the patch-list entries of the NEXT batch are fetched while the
// current batch is in flight (the row -> address chain otherwise adds an L2 round trip in front of every batch of loads).
template <typename T, int ACT, int MODE>
__global__ __launch_bounds__(256) void norm_apply_rows_kernel(const T* __restrict__ x, RowGeo g, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              const float* __restrict__ stem_x, const float* __restrict__ stem_w,
                                                              const float* __restrict__ stem_b, T* __restrict__ y) {
  constexpr int EPC = TT<T>::EPC;
  RowWalk<T> wk(g);
  float sc[EPC], sh[EPC], sw[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    sc[i] = scale[wk.cl * EPC + i]; sh[i] = shift[wk.cl * EPC + i]; sw[i] = 0.f;
    if (MODE == 2) { sw[i] = stem_w[wk.cl * EPC + i]; sh[i] += stem_b[wk.cl * EPC + i]; }
  }
  const int P = 1 << g.bs, pm = P * P - 1, sh2 = 2 * g.bs;
  const long last = wk.row_end - 1;
  if (wk.row > last) return;
  int pk[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) { const long r = min(wk.row + (long)u * wk.rpp, last); pk[u] = g.plist[r >> sh2]; }
  for (long rw = wk.row; rw <= last; rw += (long)UNR * wk.rpp) {
    size_t off[UNR]; long vox[UNR]; u32x4 xv[UNR], rv[UNR]; float sx[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = min(rw + (long)u * wk.rpp, last);
      const int q = (int)(r & pm);
      const int b = (pk[u] >> 24) & 255, pd = (pk[u] >> 16) & 255, ph = (pk[u] >> 8) & 255, pw = pk[u] & 255;
      vox[u] = (((long)b * g.D + pd * P + (q >> g.bs)) * g.H + ph * P + (q & (P - 1))) * g.W + pw * P + wk.vx;
      off[u] = (size_t)vox[u] * g.C + wk.cl * EPC;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      xv[u] = *(const u32x4*)(x + off[u]);
      if (MODE == 1) rv[u] = *(const u32x4*)(res + off[u]);
      if (MODE == 2) sx[u] = stem_x[vox[u]];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {                                   // next batch's patch-list entries (clamped: always a valid index)
      const long r = min(rw + (long)(UNR + u) * wk.rpp, last);
      pk[u] = g.plist[r >> sh2];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float f[EPC], o[EPC];
      chunk_to_f<T>(xv[u], f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) o[i] = f[i] * sc[i] + sh[i];
      if (MODE == 1) {
        float r[EPC];
        chunk_to_f<T>(rv[u], r);
#pragma unroll
        for (int i = 0; i < EPC; ++i) o[i] += r[i];
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < EPC; ++i) o[i] += sw[i] * sx[u];
      }
#pragma unroll
      for (int i = 0; i < EPC; ++i) o[i] = act_fwd(o[i], ACT);
      *(u32x4*)(y + off[u]) = f_to_chunk<T>(o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void chan_stats_rows_kernel(const T* __restrict__ x, RowGeo g, double* __restrict__ sums) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[256 * 2 * 8];
  RowWalk<T> wk(g);
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) s1[i] = s2[i] = 0.f;
  if (wk.live)
    for (long rw = wk.row; rw < wk.row_end; rw += (long)UNR * wk.rpp) {
      long v[UNR]; u32x4 xv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const long r = rw + (long)u * wk.rpp; v[u] = r < wk.row_end ? wk.voxel(g, r) : -1; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) if (v[u] >= 0) xv[u] = *(const u32x4*)(x + (size_t)v[u] * g.C + wk.cl * EPC);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (v[u] < 0) continue;
        float f[EPC];
        chunk_to_f<T>(xv[u], f);
#pragma unroll
        for (int i = 0; i < EPC; ++i) { s1[i] += f[i]; s2[i] += f[i] * f[i]; }
      }
    }
#pragma unroll
  for (int i = 0; i < EPC; ++i) { red[(threadIdx.x * 2) * 8 + i] = s1[i]; red[(threadIdx.x * 2 + 1) * 8 + i] = s2[i]; }
  __syncthreads();
  const int nlane = blockDim.x / wk.cpv;
  if (threadIdx.x < wk.cpv) {
    double a1[EPC], a2[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = a2[i] = 0.0;
    for (int vl = 0; vl < nlane; ++vl) {
      const int t = vl * wk.cpv + threadIdx.x;
#pragma unroll
      for (int i = 0; i < EPC; ++i) { a1[i] += red[(t * 2) * 8 + i]; a2[i] += red[(t * 2 + 1) * 8 + i]; }
    }
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int c = threadIdx.x * EPC + i;
      double* sr = sums + (size_t)(blockIdx.x % NREP) * g.C * 2;
      atomicAdd(&sr[c * 2], a1[i]); atomicAdd(&sr[c * 2 + 1], a2[i]);
    }
  }
}

// backward reduce over the active rows (no fill: the densify norms, whose inactive voxels carry the token gradient, keep the
// linear kernel)
template <typename T, int ACT_ = -1, int OUT_ = 0>
__global__ __launch_bounds__(256) void norm_bwd_reduce_rows_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                                   const T* __restrict__ x, RowGeo g, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, int act, double* __restrict__ bsum,
                                                                   const float* __restrict__ psc, const float* __restrict__ psh, BwdFin fin) {
  constexpr int EPC = TT<T>::EPC;
  const int actv = ACT_ >= 0 ? ACT_ : act;                     // compile-time activation / derivative source in the specialised instantiations:
  const bool has_out = ACT_ >= 0 ? OUT_ != 0 : out != nullptr;  // the per-element code is then branch-free (the generic form compiled to scalar branches per element)
  __shared__ float red[256 * 2 * 8];
  RowWalk<T> wk(g);
  float s1[EPC], s2[EPC], mu[EPC], rs[EPC], qs[EPC], qh[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) { s1[i] = s2[i] = 0.f; mu[i] = rs[i] = qs[i] = qh[i] = 0.f; }
  if (wk.live) {
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      mu[i] = mean[wk.cl * EPC + i]; rs[i] = rstd[wk.cl * EPC + i];
      if (!has_out && actv != AM_ACT_NONE) { qs[i] = psc[wk.cl * EPC + i]; qh[i] = psh[wk.cl * EPC + i]; }
    }
    for (long rw = wk.row; rw < wk.row_end; rw += (long)UNR * wk.rpp) {
      long v[UNR]; u32x4 dv[UNR], xv[UNR], ov[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const long r = rw + (long)u * wk.rpp; v[u] = r < wk.row_end ? wk.voxel(g, r) : -1; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (v[u] < 0) continue;
        const size_t off = (size_t)v[u] * g.C + wk.cl * EPC;
        dv[u] = *(const u32x4*)(dout + off); xv[u] = *(const u32x4*)(x + off);
        if (has_out && actv != AM_ACT_NONE) ov[u] = *(const u32x4*)(out + off);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (v[u] < 0) continue;
        float d[EPC], f[EPC];
        chunk_to_f<T>(dv[u], d); chunk_to_f<T>(xv[u], f);
        if (actv != AM_ACT_NONE) {
          if (has_out) {
            float o[EPC];
            chunk_to_f<T>(ov[u], o);
#pragma unroll
            for (int i = 0; i < EPC; ++i) d[i] *= act_grad(o[i], actv);
          } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) d[i] *= act_grad_pre(f[i] * qs[i] + qh[i], actv);
          }
        }
#pragma unroll
        for (int i = 0; i < EPC; ++i) { s1[i] += d[i]; s2[i] += d[i] * (f[i] - mu[i]) * rs[i]; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < EPC; ++i) { red[(threadIdx.x * 2) * 8 + i] = s1[i]; red[(threadIdx.x * 2 + 1) * 8 + i] = s2[i]; }
  __syncthreads();
  const int nlane = blockDim.x / wk.cpv;
  if (threadIdx.x < wk.cpv) {
    double a1[EPC], a2[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = a2[i] = 0.0;
    for (int vl = 0; vl < nlane; ++vl) {
      const int t = vl * wk.cpv + threadIdx.x;
#pragma unroll
      for (int i = 0; i < EPC; ++i) { a1[i] += red[(t * 2) * 8 + i]; a2[i] += red[(t * 2 + 1) * 8 + i]; }
    }
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int c = threadIdx.x * EPC + i;
      double* br = bsum + (size_t)(blockIdx.x % NREP) * g.C * 3;
      atomicAdd(&br[c * 3], a1[i]); atomicAdd(&br[c * 3 + 1], a2[i]);
    }
  }
  bwd_reduce_tail(bsum, g.C, fin);
}

template <typename T, int ACT_ = -1, int OUT_ = 0>
__global__ __launch_bounds__(256) void norm_bwd_apply_rows_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                                  const T* __restrict__ x, RowGeo g, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ k0,
                                                                  const float* __restrict__ k1, const float* __restrict__ k2, int act,
                                                                  T* __restrict__ dx, T* __restrict__ dres, double* __restrict__ dxsum,
                                                                  int dxrep, const float* __restrict__ psc, const float* __restrict__ psh,
                                                                  float* __restrict__ dx_accum, unsigned* dx_ticket) {
  constexpr int EPC = TT<T>::EPC;
  const int actv = ACT_ >= 0 ? ACT_ : act;                     // compile-time activation / derivative source in the specialised instantiations:
  const bool has_out = ACT_ >= 0 ? OUT_ != 0 : out != nullptr;  // the per-element code is then branch-free (the generic form compiled to scalar branches per element)
  __shared__ float red[256 * 8];
  RowWalk<T> wk(g);
  float mu[EPC], rs[EPC], c0[EPC], c1[EPC], c2[EPC], sx[EPC], qs[EPC], qh[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    const int c = wk.cl * EPC + i;
    sx[i] = 0.f; qs[i] = qh[i] = 0.f;
    if (wk.live) {
      mu[i] = mean[c]; rs[i] = rstd[c]; c0[i] = k0[c]; c1[i] = k1[c]; c2[i] = k2[c];
      if (!has_out && actv != AM_ACT_NONE) { qs[i] = psc[c]; qh[i] = psh[c]; }
    }
  }
  if (wk.live)
    for (long rw = wk.row; rw < wk.row_end; rw += (long)UNR * wk.rpp) {
      long v[UNR]; u32x4 dv[UNR], xv[UNR], ov[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const long r = rw + (long)u * wk.rpp; v[u] = r < wk.row_end ? wk.voxel(g, r) : -1; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (v[u] < 0) continue;
        const size_t off = (size_t)v[u] * g.C + wk.cl * EPC;
        dv[u] = *(const u32x4*)(dout + off); xv[u] = *(const u32x4*)(x + off);
        if (has_out && actv != AM_ACT_NONE) ov[u] = *(const u32x4*)(out + off);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (v[u] < 0) continue;
        const size_t off = (size_t)v[u] * g.C + wk.cl * EPC;
        float d[EPC], f[EPC];
        chunk_to_f<T>(dv[u], d); chunk_to_f<T>(xv[u], f);
        if (actv != AM_ACT_NONE) {
          if (has_out) {
            float o[EPC];
            chunk_to_f<T>(ov[u], o);
#pragma unroll
            for (int i = 0; i < EPC; ++i) d[i] *= act_grad(o[i], actv);
          } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) d[i] *= act_grad_pre(f[i] * qs[i] + qh[i], actv);
          }
        }
        if (dres) *(u32x4*)(dres + off) = f_to_chunk<T>(d);
        float r[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) r[i] = c0[i] * d[i] - c1[i] - c2[i] * (f[i] - mu[i]) * rs[i];
        const u32x4 pk = f_to_chunk<T>(r);
        *(u32x4*)(dx + off) = pk;
        if (dxsum) {
          float q[EPC];
          chunk_to_f<T>(pk, q);
#pragma unroll
          for (int i = 0; i < EPC; ++i) sx[i] += q[i];
        }
      }
    }
  if (dxsum) {
#pragma unroll
    for (int i = 0; i < EPC; ++i) red[threadIdx.x * 8 + i] = sx[i];
    __syncthreads();
    const int nlane = blockDim.x / wk.cpv;
    if (threadIdx.x < wk.cpv) {
      float a1[EPC];
#pragma unroll
      for (int i = 0; i < EPC; ++i) a1[i] = 0.f;
      for (int vl = 0; vl < nlane; ++vl)
#pragma unroll
        for (int i = 0; i < EPC; ++i) a1[i] += red[(vl * wk.cpv + threadIdx.x) * 8 + i];
#pragma unroll
      for (int i = 0; i < EPC; ++i)
        atomicAdd((unsigned long long*)&dxsum[(size_t)(blockIdx.x % dxrep) * g.C + threadIdx.x * EPC + i], (unsigned long long)__double2ll_rn((double)a1[i] * AM_DX_FIX));
    }
    dxsum_tail(dxsum, dxrep, g.C, dx_accum, dx_ticket);
  }
}

// active-patch list of a patch mask: list[i] = b << 24 | pd << 16 | ph << 8 | pw of the i-th active patch in memory order (ONE
// workgroup: ordered block scan over B*fd*fh*fw <= a few thousand bytes), count[0] = number of active patches.
__global__ __launch_bounds__(256) void mask_compact_kernel(const uint8_t* __restrict__ mask, int B, int fd, int fh, int fw,
                                                           int* __restrict__ list, int* __restrict__ count) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int n = B * fd * fh * fw;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const int a = (i < n && mask[i]) ? 1 : 0;
    const unsigned long long bal = __ballot(a);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (a) {
      int q = i;
      const int pw = q % fw; q /= fw;
      const int ph = q % fh; q /= fh;
      const int pd = q % fd; const int b = q / fd;
      list[off + pre] = (b << 24) | (pd << 16) | (ph << 8) | pw;
    }
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = base;
}

// ------------------------------------------------------------------ generic per-channel sum (bias grads)
template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void chan_sum_kernel(const T* __restrict__ x, Geo g, float* __restrict__ out) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[NT * 8];
  Walk<T> wk(g.C, NT);
  float s1[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) s1[i] = 0.f;
  const long v0 = (long)blockIdx.x * g.vpw, v1 = min(v0 + (long)g.vpw, g.nvox());
  if (wk.live)
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      if (!g.active(v)) continue;
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + v * g.C + wk.cl * EPC), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) s1[i] += f[i];
    }
#pragma unroll
  for (int i = 0; i < EPC; ++i) red[threadIdx.x * 8 + i] = s1[i];
  __syncthreads();
  if (threadIdx.x < wk.cpv) {
    float a1[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = 0.f;
    for (int vl = 0; vl < wk.vpp; ++vl)
#pragma unroll
      for (int i = 0; i < EPC; ++i) a1[i] += red[(vl * wk.cpv + threadIdx.x) * 8 + i];
#pragma unroll
    for (int i = 0; i < EPC; ++i) atomicAdd(&out[threadIdx.x * EPC + i], a1[i]);
  }
}

// ------------------------------------------------------------------ elementwise add  y = a + b (dense)
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* y, size_t nchunk) {
  constexpr int EPC = TT<T>::EPC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (size_t)gridDim.x * 256) {
    float fa[EPC], fb[EPC];
    chunk_to_f<T>(((const u32x4*)a)[i], fa); chunk_to_f<T>(((const u32x4*)b)[i], fb);
#pragma unroll
    for (int k = 0; k < EPC; ++k) fa[k] += fb[k];
    ((u32x4*)y)[i] = f_to_chunk<T>(fa);
  }
}

// ------------------------------------------------------------------ stem: Cin = 1 convolution (k = 1 or 3)
// y[v][c] = b[c] + sum_t w[c][t] * xm[v + t - pad],  xm = x on active patches, 0 elsewhere / outside.
// One workgroup per 4x8x16 brick of an active patch: the haloed input brick is staged once into LDS with the patch mask
// and the volume bounds already applied (so the 27-tap loop is pure LDS reads + FMAs, the wavefront re-reads its
// neighbours' values from LDS instead of re-fetching them), one 16-byte chunk of output channels per lane.
constexpr int SBD = 4, SBH = 8, SBW = 16;                   // divides the 16^3 patch: a brick never straddles patches

// Tap-outer form: a thread owns one 16-byte chunk of channels for its VPT voxels of the brick; per tap it reads the chunk's
// weights once and one x value per voxel (LDS traffic 16 reads per 64 FMAs; the voxel-outer form needed 72).
// part != nullptr: per-workgroup per-channel (sum, sum of squares) of the STORED values -> [brick][C][2] (zeros for inactive
// bricks), the statistics of the norm that follows without another pass over y.
template <typename T, int K>
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ x, Geo g,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ y, float* __restrict__ part) {
  constexpr int EPC = TT<T>::EPC;
  constexpr int NTP = K * K * K, PAD = K / 2;
  constexpr int ED = SBD + 2 * PAD, EH = SBH + 2 * PAD, EW = SBW + 2 * PAD;
  constexpr int MV = SBD * SBH * SBW;
  extern __shared__ float sm[];                              // [k^3][C] weights (tap-major), then the haloed x brick
  float* wl = sm;
  float* xb = sm + g.C * NTP;
  int bid = blockIdx.x;
  const int nbw = g.W / SBW, nbh = g.H / SBH, nbd = g.D / SBD;
  const int bw_ = bid % nbw; bid /= nbw;
  const int bh_ = bid % nbh; bid /= nbh;
  const int bd_ = bid % nbd; const int b = bid / nbd;
  const int d0 = bd_ * SBD, h0 = bh_ * SBH, w0 = bw_ * SBW;
  float* prow = part ? part + (size_t)blockIdx.x * g.C * 2 : nullptr;
  if (!g.mask.active(b, d0, h0, w0)) {                        // whole brick inactive (wave-uniform)
    if (prow) for (int i = threadIdx.x; i < g.C * 2; i += 256) prow[i] = 0.f;
    return;
  }
  for (int i = threadIdx.x; i < g.C * NTP; i += 256) wl[(i % NTP) * g.C + i / NTP] = w[i];
  for (int e = threadIdx.x; e < ED * EH * EW; e += 256) {
    const int ex = e % EW, ey = (e / EW) % EH, ez = e / (EW * EH);
    const int id = d0 + ez - PAD, ih = h0 + ey - PAD, iw = w0 + ex - PAD;
    float v = 0.f;
    if (id >= 0 && id < g.D && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && g.mask.active(b, id, ih, iw))
      v = x[((size_t)(b * g.D + id) * g.H + ih) * g.W + iw];
    xb[e] = v;
  }
  __syncthreads();
  Walk<T> wk(g.C);
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) s1[i] = s2[i] = 0.f;
  if (wk.live) {
    float bs[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) bs[i] = bias ? bias[wk.cl * EPC + i] : 0.f;
    constexpr int VPT = 8;                                   // voxels per thread per round
    for (int vb = wk.vl; vb < MV; vb += wk.vpp * VPT) {
      float o[VPT][EPC];
      int xo[VPT];
#pragma unroll
      for (int q = 0; q < VPT; ++q) {
        const int v = vb + q * wk.vpp;
        const int vv = v < MV ? v : 0;
        xo[q] = ((vv / (SBW * SBH)) * EH + (vv / SBW) % SBH) * EW + vv % SBW;
#pragma unroll
        for (int i = 0; i < EPC; ++i) o[q][i] = bs[i];
      }
#pragma unroll 1
      for (int td = 0; td < K; ++td)                          // (rolled: unrolled, hipcc preloads all 27 x EPC weights -> 256 VGPRs, spills)
#pragma unroll 1
        for (int th = 0; th < K; ++th)
#pragma unroll
          for (int tw = 0; tw < K; ++tw) {
            const int ti = (td * K + th) * K + tw;
            const f32x4 wa = *(const f32x4*)(wl + ti * g.C + wk.cl * EPC);
            f32x4 wb = wa;
            if constexpr (EPC == 8) wb = *(const f32x4*)(wl + ti * g.C + wk.cl * EPC + 4);
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
              const float xv = xb[xo[q] + (td * EH + th) * EW + tw];
#pragma unroll
              for (int i = 0; i < 4; ++i) o[q][i] += wa[i] * xv;
              if constexpr (EPC == 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[q][4 + i] += wb[i] * xv;
              }
            }
          }
#pragma unroll
      for (int q = 0; q < VPT; ++q) {
        const int v = vb + q * wk.vpp;
        if (v >= MV) continue;
        const int lw = v % SBW, lh = (v / SBW) % SBH, ld = v / (SBW * SBH);
        const size_t vox = ((size_t)(b * g.D + d0 + ld) * g.H + h0 + lh) * g.W + w0 + lw;
        const u32x4 pk = f_to_chunk<T>(o[q]);
        *(u32x4*)(y + vox * g.C + wk.cl * EPC) = pk;
        if (prow) {
          float r[EPC];
          chunk_to_f<T>(pk, r);
#pragma unroll
          for (int i = 0; i < EPC; ++i) { s1[i] += r[i]; s2[i] += r[i] * r[i]; }
        }
      }
    }
  }
  if (prow) {                                                 // fold the voxel lanes: red[vl][cl][EPC][2] reuses the weight area
    __syncthreads();
    float* red = sm;
    if (wk.live) {
#pragma unroll
      for (int i = 0; i < EPC; ++i) { red[(threadIdx.x * EPC + i) * 2] = s1[i]; red[(threadIdx.x * EPC + i) * 2 + 1] = s2[i]; }
    }
    __syncthreads();
    for (int c2 = threadIdx.x; c2 < g.C * 2; c2 += 256) {    // c2 = channel * 2 + {sum, sumsq}
      const int c = c2 >> 1, cl = c / EPC, i = c % EPC;
      float a = 0.f;
      for (int vl = 0; vl < wk.vpp; ++vl) a += red[((vl * wk.cpv + cl) * EPC + i) * 2 + (c2 & 1)];
      prow[c2] = a;
    }
  }
}


// ------------------------------------------------------------------ stem k3 on the matrix cores (bf16 storage mode)
// The Cin = 1 stem is a [voxels x 27] x [27 x C] product: K = 27 taps padded to the 32 of one v_mfma_f32_16x16x32_bf16.  One
// workgroup owns one ACTIVE 16^3 patch (active-patch list): the haloed 18^3 input patch is staged once into LDS (fp32) with the
// patch mask and the volume bounds applied, the weights live in registers as MFMA A-fragments for the whole kernel, and per
// 16-voxel w-row a lane gathers the 8 taps of its k-group from LDS (8 four-byte reads), issues 3 C/16 MFMAs, adds the bias and
// stores 8 consecutive channels (16 bytes; a wave writes 1 KB runs).
// The input VOLUME and the stem weights are not bf16-storage tensors (C = 1 input, fp32 master weights; oracle._qw): both enter the
// matrix cores as hi + lo bf16 parts (w_lo x_hi + w_hi x_lo + w_hi x_hi: 16 significant bits per operand).  A plain bf16 rounding of
// the volume (round 3-4) perturbed a smooth CT-like input by as much as its local differences, and the stem weight's gradient sat at
// 1.7 x the ideal emulation's distance from fp32 (profiles/r05_experiments.md section 7).  The VALU form (stem_conv_fwd_kernel) spends 27 LDS reads and
// 216 FMAs per voxel-chunk and runs at 24 TFLOP/s (450 us for 430 MB of output); this one is bound by its output stream.
// One partials row per patch (the VALU kernel leaves one per 512-voxel brick, inactive ones included).
template <int NS>
__global__ __launch_bounds__(256) void stem_conv_mfma_kernel(const float* __restrict__ x, int D, int H, int W, int C, MaskView mask,
                                                             const int* __restrict__ plist, const float* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                             float* __restrict__ part) {
  constexpr int E = 18;                                          // haloed patch edge
  __shared__ float xl[E * E * E + 8];
  __shared__ float red[4 * 16 * NS * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int pk = plist[blockIdx.x];
  const int b = (pk >> 24) & 255, pd = (pk >> 16) & 255, ph = (pk >> 8) & 255, pw = pk & 255;
  const int d0 = pd * 16, h0 = ph * 16, w0 = pw * 16;
  // ---- haloed patch -> LDS (fp32), zero outside the volume and in inactive neighbour patches
  for (int e = tid; e < E * E * E; e += 256) {
    const int ex = e % E, ey = (e / E) % E, ez = e / (E * E);
    const int id = d0 + ez - 1, ih = h0 + ey - 1, iw = w0 + ex - 1;
    float v = 0.f;
    if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && mask.active(b, id, ih, iw))
      v = x[((size_t)(b * D + id) * H + ih) * W + iw];
    xl[e] = v;
  }
  // ---- weights -> A fragments (row R of tile i <-> channel crow(16 i + R): a lane ends up with 8 consecutive channels)
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  bfx8 af[NS], afl[NS];                                         // hi / lo parts of the weights
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = 8 * g + j;
    toff[j] = t < 27 ? ((t / 9) * E + (t / 3) % 3) * E + t % 3 : 0;
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int ch = crow(i * 16 + r16);
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int t = 8 * g + j; wv[j] = t < 27 ? w[ch * 27 + t] : 0.f; }
    u32x4 h, l;
    split8_bf16(wv, h, l);
    af[i] = __builtin_bit_cast(bfx8, h); afl[i] = __builtin_bit_cast(bfx8, l);
  }
  f32x4 bia[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) bia[i] = bias ? *(const f32x4*)(bias + (i >> 1) * 32 + g * 8 + (i & 1) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[NS][4], s2[NS][4];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) s1[i][q] = s2[i][q] = 0.f;
  __syncthreads();
  // ---- 256 w-rows of 16 voxels: wave `wave` takes rows wave, wave + 4, ...
  for (int row = wave; row < 256; row += 4) {
    const int dz = row >> 4, dy = row & 15;
    const int base = (dz * E + dy) * E + r16;                    // (d - 1 + td, h - 1 + th, w - 1 + tw) with the halo offset folded in
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = xl[base + toff[j]];
    u32x4 xh, xo;
    split8_bf16(xv, xh, xo);
    const bfx8 bf = __builtin_bit_cast(bfx8, xh), bfl = __builtin_bit_cast(bfx8, xo);
    f32x4 o[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {                               // small terms first
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[i], bf, bia[i], 0, 0, 0);
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfl, o[i], 0, 0, 0);
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, o[i], 0, 0, 0);
    }
    const size_t vox = ((size_t)(b * D + d0 + dz) * H + h0 + dy) * W + w0 + r16;
    bf16_t* dst = y + vox * C + g * 8;
#pragma unroll
    for (int h = 0; h < NS / 2; ++h) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
      const bfx4 p0 = __builtin_convertvector(o[2 * h], bfx4), p1 = __builtin_convertvector(o[2 * h + 1], bfx4);
      *(bfx8*)(dst + h * 32) = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
      const f32x4 q0 = __builtin_convertvector(p0, f32x4), q1 = __builtin_convertvector(p1, f32x4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s1[2 * h][c] += q0[c]; s2[2 * h][c] += q0[c] * q0[c];
        s1[2 * h + 1][c] += q1[c]; s2[2 * h + 1][c] += q1[c] * q1[c];
      }
    }
  }
  if (part) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a1 = s1[i][c], a2 = s2[i][c];
        a1 = row16_sum(a1); a2 = row16_sum(a2);
        if (r16 == 0) {
          const int ch = (i >> 1) * 32 + g * 8 + (i & 1) * 4 + c;
          red[(wave * 16 * NS + ch) * 2] = a1; red[(wave * 16 * NS + ch) * 2 + 1] = a2;
        }
      }
    __syncthreads();
    if (tid < 16 * NS) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) { a1 += red[(wv * 16 * NS + tid) * 2]; a2 += red[(wv * 16 * NS + tid) * 2 + 1]; }
      float* pr = part + ((size_t)blockIdx.x * C + tid) * 2;
      pr[0] = a1; pr[1] = a2;
    }
  }
}

// ------------------------------------------------------------------ stem weight gradient on the matrix cores (bf16 storage mode)
// dW[c][t] = sum_v dy[v][c] * xm[v + t - pad] is a [C x voxels] x [voxels x taps] product whose contraction index is the voxel:
// M = channel, N = tap (27 taps + one column of ones that yields db, padded to 32; k1: centre tap + ones), K = 32 voxels (two
// 16-voxel w-rows) per v_mfma_f32_16x16x32_bf16.  Persistent workgroups walk the active-patch list; per patch the haloed 18^3 input
// patch sits in LDS as fp32 and enters the matrix cores as hi + lo bf16 parts (the values the forward kernel multiplied: 16
// significant bits), dy is staged one 16x16 d-plane at a time in padded rows and
// fetched as A-fragments with the transposing LDS read (ds_read_b64_tr_b16, as conv_wgrad.hip), the B-fragments (x at the lane's
// tap, 8 voxels) are 8 four-byte LDS reads split in registers.  Accumulators live in registers over the whole walk; one flush of C x 28 atomics per
// workgroup.  The VALU form below re-stages x three times and runs at 5 TFLOP/s (1.2 ms per call at 128^3, B=8).
__device__ __forceinline__ s16x4 tr_read16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <int NS, int K>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, int D, int H, int W,
                                                              MaskView mask, const int* __restrict__ plist, int n_active,
                                                              float* __restrict__ dw, float* __restrict__ db, float* __restrict__ det_ws) {
  constexpr int E = 18, C = 16 * NS, RS = 2 * C + 32;           // dy rows: C bf16 + 32 B pad (conflict-free transposing reads)
  constexpr int NTT = K == 3 ? 2 : 1, NTAP = K * K * K;
  constexpr int XB = (E * E * E + 8) * 4;                        // bytes of the x patch (fp32; multiple of 16)
  constexpr int CPV = C / 8, NLD = CPV;                          // 16-byte chunks per voxel; loads per thread and d-plane (256 voxels)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* xl = (float*)lds;
  unsigned char* yl = lds + XB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int q = (lane >> 2) & 3, p = lane & 3;
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  f32x4 acc[NS][NTT];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int toff[NTT], kind[NTT];                                      // kind: 0 real tap, 1 ones column (db), 2 padding
#pragma unroll
  for (int n = 0; n < NTT; ++n) {
    const int t = 16 * n + r16;
    kind[n] = t < NTAP ? 0 : (t == NTAP ? 1 : 2);
    toff[n] = t < NTAP ? (K == 3 ? ((t / 9) * E + (t / 3) % 3) * E + t % 3 : (E + 1) * E + 1) : 0;
  }
  for (int pi = blockIdx.x; pi < n_active; pi += gridDim.x) {
    const int pk = plist[pi];
    const int b = (pk >> 24) & 255, d0 = ((pk >> 16) & 255) * 16, h0 = ((pk >> 8) & 255) * 16, w0 = (pk & 255) * 16;
    __syncthreads();                                             // the previous patch's reads of xl / yl are done
    for (int e = tid; e < E * E * E; e += 256) {
      const int ex = e % E, ey = (e / E) % E, ez = e / (E * E);
      const int id = d0 + ez - 1, ih = h0 + ey - 1, iw = w0 + ex - 1;
      float v = 0.f;
      if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && mask.active(b, id, ih, iw))
        v = x[((size_t)(b * D + id) * H + ih) * W + iw];
      xl[e] = v;
    }
    u32x4 ld[NLD];
    auto load_plane = [&](int dz) {
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int idx = tid + it * 256, vox = idx / CPV, ch = idx % CPV;
        const size_t gv = ((size_t)(b * D + d0 + dz) * H + h0 + (vox >> 4)) * W + w0 + (vox & 15);
        ld[it] = *(const u32x4*)(dy + gv * C + ch * 8);
      }
    };
    load_plane(0);
    for (int dz = 0; dz < 16; ++dz) {
      __syncthreads();                                           // previous plane's fragment reads are done (first plane: xl is published below)
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int idx = tid + it * 256, vox = idx / CPV, ch = idx % CPV;
        *(u32x4*)(yl + vox * RS + ch * 16) = ld[it];
      }
      load_plane(dz < 15 ? dz + 1 : 15);                         // the next plane's loads fly during this plane's MFMAs (the last one re-loads: branch-free)
      __syncthreads();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int s = wave + 4 * s2;                             // k-step: h-rows 2s, 2s+1 of this d-plane
        const int v1 = s * 32 + g * 4 + q, v2 = v1 + 16;
        bfx8 af[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const s16x4 lo = tr_read16(yl + v1 * RS + (16 * i + 4 * p) * 2), hi = tr_read16(yl + v2 * RS + (16 * i + 4 * p) * 2);
          af[i] = __builtin_bit_cast(bfx8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        }
        const int xbase = (dz * E + 2 * s) * E + 4 * g;           // halo offset folded into the tap offsets
#pragma unroll
        for (int n = 0; n < NTT; ++n) {
          float qv[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { qv[j] = xl[xbase + toff[n] + j]; qv[4 + j] = xl[xbase + toff[n] + E + j]; }
          if (kind[n]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qv[j] = kind[n] == 1 ? 1.f : 0.f;
          }
          u32x4 xh, xo;
          split8_bf16(qv, xh, xo);
          const bfx8 bf = __builtin_bit_cast(bfx8, xh), bfl = __builtin_bit_cast(bfx8, xo);
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfl, acc[i][n], 0, 0, 0);
            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][n], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- flush: D row 4g+r of tile i = channel 16i + 4g + r, column r16 of tile n = tap 16n + r16.  Fold the 4 waves in LDS first.
  __syncthreads();
  float* red = (float*)lds;                                      // [4 waves][C][16 * NTT]
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int n = 0; n < NTT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * C + 16 * i + 4 * g + r) * (16 * NTT) + 16 * n + r16] = acc[i][n][r];
  __syncthreads();
  for (int i = tid; i < C * (NTAP + 1); i += 256) {
    const int c = i / (NTAP + 1), t = i % (NTAP + 1);
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) v += red[(wv * C + c) * (16 * NTT) + t];
    if (det_ws) det_ws[(size_t)blockIdx.x * (C * (NTAP + 1)) + i] = v;      // deterministic mode: one row per workgroup, folded in order afterwards
    else if (t < NTAP) atomicAdd(&dw[c * NTAP + t], v);
    else if (db) atomicAdd(&db[c], v);
  }
}

// deterministic mode of the stem weight gradient: rows [nwg][C * (NTAP + 1)] of per-workgroup sums -> dw / db, in workgroup order
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ ws, int nwg, int C, int ntap, float* __restrict__ dw,
                                                              float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * (ntap + 1)) return;
  float s = 0.f;
  for (int r = 0; r < nwg; ++r) s += ws[(size_t)r * C * (ntap + 1) + i];
  const int c = i / (ntap + 1), t = i % (ntap + 1);
  if (t < ntap) dw[c * ntap + t] += s;
  else if (db) db[c] += s;
}

// dW[c][t] += sum_v dy[v][c] * xm[v+t-pad];  db[c] += sum_v dy[v][c]     (same brick staging; 9 taps at a time in registers).
// Persistent: a workgroup walks a strided set of bricks and keeps its [C][k^3+1] partial sums in LDS, so the global
// atomics happen once per workgroup, not once per brick.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy, Geo g, int k,
                                                              float* __restrict__ dw, float* __restrict__ db) {
  constexpr int EPC = TT<T>::EPC;
  extern __shared__ float sm[];                              // [C][k^3 + 1] accumulators, then the haloed x brick
  const int nt = k * k * k, pad = k / 2;
  const int ED = SBD + 2 * pad, EH = SBH + 2 * pad, EW = SBW + 2 * pad;
  float* acc_l = sm;
  float* xb = sm + g.C * (nt + 1);
  const int nbw = g.W / SBW, nbh = g.H / SBH, nbd = g.D / SBD;
  const int nbrick = g.B * nbd * nbh * nbw;
  for (int i = threadIdx.x; i < g.C * (nt + 1); i += 256) acc_l[i] = 0.f;
  Walk<T> wk(g.C);
  for (int tg = 0; tg < nt; tg += 9) {                        // 9 taps x EPC channels of partial sums live in registers
    float a[9][EPC], sb[EPC];                                 // across ALL bricks of this workgroup
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int i = 0; i < EPC; ++i) a[q][i] = 0.f;
#pragma unroll
    for (int i = 0; i < EPC; ++i) sb[i] = 0.f;
    for (int brick = blockIdx.x; brick < nbrick; brick += gridDim.x) {
      int bid = brick;
      const int bw_ = bid % nbw; bid /= nbw;
      const int bh_ = bid % nbh; bid /= nbh;
      const int bd_ = bid % nbd; const int b = bid / nbd;
      const int d0 = bd_ * SBD, h0 = bh_ * SBH, w0 = bw_ * SBW;
      if (!g.mask.active(b, d0, h0, w0)) continue;            // wave-uniform
      __syncthreads();                                        // previous brick's reads of xb are done
      for (int e = threadIdx.x; e < ED * EH * EW; e += 256) {
        const int ex = e % EW, ey = (e / EW) % EH, ez = e / (EW * EH);
        const int id = d0 + ez - pad, ih = h0 + ey - pad, iw = w0 + ex - pad;
        float v = 0.f;
        if (id >= 0 && id < g.D && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && g.mask.active(b, id, ih, iw))
          v = x[((size_t)(b * g.D + id) * g.H + ih) * g.W + iw];
        xb[e] = v;
      }
      __syncthreads();
      if (!wk.live) continue;
      for (int v = wk.vl; v < SBD * SBH * SBW; v += wk.vpp) {
        const int lw = v % SBW, lh = (v / SBW) % SBH, ld = v / (SBW * SBH);
        const size_t vox = ((size_t)(b * g.D + d0 + ld) * g.H + h0 + lh) * g.W + w0 + lw;
        float d[EPC];
        chunk_to_f<T>(*(const u32x4*)(dy + vox * g.C + wk.cl * EPC), d);
        if (tg == 0) {
#pragma unroll
          for (int i = 0; i < EPC; ++i) sb[i] += d[i];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int ti = tg + q;
          if (ti < nt) {
            const int td = ti / (k * k), th = (ti / k) % k, tw = ti % k;
            const float xv = xb[((ld + td) * EH + lh + th) * EW + lw + tw];
#pragma unroll
            for (int i = 0; i < EPC; ++i) a[q][i] += d[i] * xv;
          }
        }
      }
    }
    if (wk.live) {
#pragma unroll
      for (int q = 0; q < 9; ++q)
        if (tg + q < nt)
#pragma unroll
          for (int i = 0; i < EPC; ++i) atomicAdd(&acc_l[(wk.cl * EPC + i) * (nt + 1) + tg + q], a[q][i]);
      if (tg == 0)
#pragma unroll
        for (int i = 0; i < EPC; ++i) atomicAdd(&acc_l[(wk.cl * EPC + i) * (nt + 1) + nt], sb[i]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.C * (nt + 1); i += 256) {
    const int c = i / (nt + 1), t = i % (nt + 1);
    const float v = acc_l[i];
    if (v == 0.f) continue;
    if (t < nt) atomicAdd(&dw[c * nt + t], v);
    else if (db) atomicAdd(&db[c], v);
  }
}

// ------------------------------------------------------------------ 1x1 projection C -> 1 (P/decoder3D.py:51,61)
// psc / psh != nullptr: x is the INPUT of a per-channel affine map (a train-mode BatchNorm whose output is never materialised):
//   rec = b + sum_c w[c] * (x[c] * psc[c] + psh[c])  =  (b + sum_c w[c] psh[c]) + sum_c (w[c] psc[c]) x[c]
template <typename T>
__global__ __launch_bounds__(256) void proj_fwd_kernel(const T* __restrict__ x, long nvox, int C, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ psc,
                                                       const float* __restrict__ psh, float* __restrict__ rec) {
  constexpr int EPC = TT<T>::EPC;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvox) return;
  float s = b[0];
  if (psc) {
    for (int c = 0; c < C; c += EPC) {
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + (size_t)v * C + c), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) s += (f[i] * psc[c + i] + psh[c + i]) * w[c + i];       // (uniform operands: scalar loads)
    }
  } else {
    for (int c = 0; c < C; c += EPC) {
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + (size_t)v * C + c), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) s += f[i] * w[c + i];
    }
  }
  rec[v] = s;
}

// ---- projection head + the BatchNorm in front of it, backward (P/decoder3D.py:22 -> :51,61) -------------------------------------
// rec = proj(o), o = BN_train(x) = x * scale + shift (no activation, no skip: the LAST decoder block).  The gradient wrt o is the
// rank-1 tensor g[v][c] = drec[v] * w[c], so neither o nor g has to exist in memory:
//   S0 = sum_v drec[v],  S1[c] = sum_v drec[v] * (x[v][c] - mean[c])                         (ONE pass over x and drec)
//   proj:  db += S0,  dw[c] += sum_v drec * o = scale[c] * S1[c] + beta[c] * S0              (shift + mean * scale = beta)
//   BN:    sum g = w S0,  sum g xhat = rstd w S1  ->  dbeta += w S0, dgamma += rstd w S1,
//          dx[v][c] = k0 w drec[v] - k1 - k2 xhat[v][c],  k0 = gamma rstd, k1 = k0 w S0 / n, k2 = k0 rstd w S1 / n   (second pass)
// Against proj_bwd + norm_bwd_reduce + norm_bwd_apply this drops the write and two reads of g and the read of o (forward: o's
// write + read), 6 of the 13 tensor-sized transfers of the head.
struct ProjNormFin {
  unsigned* ticket; double n;
  const float *gamma, *beta, *rstd, *scale, *w;
  float *k0w, *k1, *k2, *dgamma, *dbeta, *dw, *db;
};

template <typename T>
__global__ __launch_bounds__(256) void proj_norm_bwd_reduce_kernel(const T* __restrict__ x, const float* __restrict__ drec, long nvox, int C,
                                                                   int vpw, const float* __restrict__ mean, double* __restrict__ acc,
                                                                   ProjNormFin fin) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[256 * 8];
  __shared__ float redb[4];
  __shared__ double s0sh;
  Walk<T> wk(C);
  float s1[EPC], mu[EPC], sb = 0.f;
#pragma unroll
  for (int i = 0; i < EPC; ++i) { s1[i] = 0.f; mu[i] = wk.live ? mean[wk.cl * EPC + i] : 0.f; }
  const long v0 = (long)blockIdx.x * vpw, v1 = min(v0 + (long)vpw, nvox);
  if (wk.live)
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      const float d = drec[v];
      float f[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + (size_t)v * C + wk.cl * EPC), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) s1[i] += d * (f[i] - mu[i]);
      if (wk.cl == 0) sb += d;
    }
#pragma unroll
  for (int i = 0; i < EPC; ++i) red[threadIdx.x * 8 + i] = s1[i];
  sb = warp_sum(sb);
  if ((threadIdx.x & 63) == 0) redb[threadIdx.x >> 6] = sb;
  __syncthreads();
  double* ar = acc + (size_t)(blockIdx.x % NREP) * (C + 1);
  if (threadIdx.x < wk.cpv) {
    double a1[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = 0.0;
    for (int vl = 0; vl < wk.vpp; ++vl)
#pragma unroll
      for (int i = 0; i < EPC; ++i) a1[i] += red[(vl * wk.cpv + threadIdx.x) * 8 + i];
#pragma unroll
    for (int i = 0; i < EPC; ++i) atomicAdd(&ar[threadIdx.x * EPC + i], a1[i]);
  }
  if (threadIdx.x == 0) atomicAdd(&ar[C], (double)redb[0] + (double)redb[1] + (double)redb[2] + (double)redb[3]);
  // ---- last workgroup: coefficients of the apply pass + the four parameter gradients; accumulators re-zeroed
  if (!last_workgroup(fin.ticket)) return;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int r = 0; r < NREP; ++r) t += __longlong_as_double(atomicExch((unsigned long long*)&acc[(size_t)r * (C + 1) + C], 0ull));
    s0sh = t;
    fin.db[0] += (float)t;
  }
  __syncthreads();
  const double S0 = s0sh;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double S1 = 0.0;
    for (int r = 0; r < NREP; ++r) S1 += __longlong_as_double(atomicExch((unsigned long long*)&acc[(size_t)r * (C + 1) + c], 0ull));
    const double wc = fin.w[c], rs = fin.rstd[c];
    const double b1 = wc * S0, b2 = rs * wc * S1;
    const float gr = fin.gamma[c] * fin.rstd[c];
    fin.k0w[c] = gr * fin.w[c]; fin.k1[c] = gr * (float)(b1 / fin.n); fin.k2[c] = gr * (float)(b2 / fin.n);
    fin.dgamma[c] += (float)b2;
    fin.dbeta[c] += (float)b1;
    fin.dw[c] += (float)((double)fin.scale[c] * S1 + (double)fin.beta[c] * S0);
  }
  if (threadIdx.x == 0) atomicExch(fin.ticket, 0u);
}

template <typename T>
__global__ __launch_bounds__(256) void proj_norm_bwd_apply_kernel(const T* __restrict__ x, const float* __restrict__ drec, long nvox, int C,
                                                                  int vpw, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ k0w, const float* __restrict__ k1,
                                                                  const float* __restrict__ k2, T* __restrict__ dx) {
  constexpr int EPC = TT<T>::EPC;
  Walk<T> wk(C);
  if (!wk.live) return;
  float mu[EPC], c0[EPC], c1[EPC], c2[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    const int c = wk.cl * EPC + i;
    mu[i] = mean[c]; c0[i] = k0w[c]; c1[i] = k1[c]; c2[i] = k2[c] * rstd[c];
  }
  const long v0 = (long)blockIdx.x * vpw, v1 = min(v0 + (long)vpw, nvox);
  for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
    const float d = drec[v];
    const size_t off = (size_t)v * C + wk.cl * EPC;
    float f[EPC], r[EPC];
    chunk_to_f<T>(*(const u32x4*)(x + off), f);
#pragma unroll
    for (int i = 0; i < EPC; ++i) r[i] = c0[i] * d - c1[i] - c2[i] * (f[i] - mu[i]);
    *(u32x4*)(dx + off) = f_to_chunk<T>(r);
  }
}

// dx[v][c] = drec[v]*w[c];  dw[c] += sum_v drec[v]*x[v][c];  db += sum_v drec[v]
template <typename T>
__global__ __launch_bounds__(256) void proj_bwd_kernel(const T* __restrict__ x, const float* __restrict__ drec, long nvox, int C, int vpw,
                                                       const float* __restrict__ w, T* __restrict__ dx, float* __restrict__ dw,
                                                       float* __restrict__ db, float* __restrict__ det_ws) {
  constexpr int EPC = TT<T>::EPC;
  __shared__ float red[256 * 8];
  __shared__ float redb[4];
  Walk<T> wk(C);
  float sw[EPC], wv[EPC], sb = 0.f;
#pragma unroll
  for (int i = 0; i < EPC; ++i) { sw[i] = 0.f; wv[i] = wk.live ? w[wk.cl * EPC + i] : 0.f; }
  const long v0 = (long)blockIdx.x * vpw, v1 = min(v0 + (long)vpw, nvox);
  if (wk.live)
    for (long v = v0 + wk.vl; v < v1; v += wk.vpp) {
      const float d = drec[v];
      const size_t off = (size_t)v * C + wk.cl * EPC;
      float f[EPC], o[EPC];
      chunk_to_f<T>(*(const u32x4*)(x + off), f);
#pragma unroll
      for (int i = 0; i < EPC; ++i) { sw[i] += d * f[i]; o[i] = d * wv[i]; }
      *(u32x4*)(dx + off) = f_to_chunk<T>(o);
      if (wk.cl == 0) sb += d;
    }
#pragma unroll
  for (int i = 0; i < EPC; ++i) red[threadIdx.x * 8 + i] = sw[i];
  sb = warp_sum(sb);
  if ((threadIdx.x & 63) == 0) redb[threadIdx.x >> 6] = sb;
  __syncthreads();
  if (threadIdx.x < wk.cpv) {
    float a1[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) a1[i] = 0.f;
    for (int vl = 0; vl < wk.vpp; ++vl)
#pragma unroll
      for (int i = 0; i < EPC; ++i) a1[i] += red[(vl * wk.cpv + threadIdx.x) * 8 + i];
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      if (det_ws) det_ws[(size_t)blockIdx.x * (C + 1) + threadIdx.x * EPC + i] = a1[i];    // deterministic mode: a row per workgroup, folded in order
      else atomicAdd(&dw[threadIdx.x * EPC + i], a1[i]);
    }
  }
  if (threadIdx.x == 0) {
    const float sbt = redb[0] + redb[1] + redb[2] + redb[3];
    if (det_ws) det_ws[(size_t)blockIdx.x * (C + 1) + C] = sbt; else atomicAdd(db, sbt);
  }
}

// rows [nrow][n] of per-workgroup sums -> dst0[0..n0) += column sums (in row order), dst1[0..n-n0) likewise
__global__ __launch_bounds__(256) void rows_fold_kernel(const float* __restrict__ ws, int nrow, int n, int n0, float* __restrict__ dst0,
                                                        float* __restrict__ dst1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < nrow; ++r) s += ws[(size_t)r * n + i];
  if (i < n0) dst0[i] += s; else if (dst1) dst1[i - n0] += s;
}

// ------------------------------------------------------------------ fp32 -> (hi, lo) bf16 planes (AM_DT_F32S weight gradients)
// x = hi + lo + r, hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|: the weight gradient of the split mode is three bf16 contractions
// (hi hi + hi lo + lo hi) of these planes accumulated into the fp32 gradient (ops.conv3d_wgrad); one streaming pass, 16 bytes per lane in
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n4) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const u32x4 c = *(const u32x4*)(x + 4 * i);
    unsigned h2[2], l2[2];
    split4_bf16(c, h2, l2);
    *(u32x2_*)(hi + 4 * i) = u32x2_{h2[0], h2[1]};
    *(u32x2_*)(lo + 4 * i) = u32x2_{l2[0], l2[1]};
  }
}

// ------------------------------------------------------------------ weight (un)packing
// dst[t][r][k] = src[r*sr + k*sk + t] for r < R, k < K, zero in the padding (dst is [taps][Rp][Kp])
// (fp32 master -> compute dtype, MFMA row-fragment layout, whole tiles so the conv inner loop needs no bounds).
// One workgroup per (row r, 64 consecutive k): when taps are innermost in the source (sk == taps, the forward
// pack) the 64*taps source floats are one contiguous run -> coalesced read, LDS transpose, coalesced 128-byte row writes.
// packed element (row base `rowp`, k): plain for float / bf16; AM_DT_F32S: every 16-channel group of a row is 64 bytes
// [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] in bf16 (hi = bf16(w), lo = bf16(w - hi)) -- the LDS image conv_igemm's split mode contracts
template <typename T> __device__ __forceinline__ void pack_store(T* rowp, int k, float v) {
  if constexpr (std::is_same<T, f32s_t>::value) {
    bf16_t* g = (bf16_t*)rowp + (k >> 4) * 32 + (k & 15);
    const bf16_t h = f2bf(v);
    g[0] = h; g[16] = f2bf(v - bf2f(h));
  } else {
    TT<T>::st(rowp + k, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int K, int taps,
                                                          long sr, long sk, int Rp, int Kp) {
  extern __shared__ float tile[];                          // [64][taps]
  const int r = blockIdx.x, k0 = blockIdx.y * 64;
  const int n = 64 * taps;
  if (r < R) {
    if (sk == taps) {
      const float* base = src + r * sr + (long)k0 * taps;
      const int lim = (K - k0 < 64 ? (K - k0 > 0 ? K - k0 : 0) : 64) * taps;
      for (int i = threadIdx.x; i < n; i += 256) tile[i] = i < lim ? base[i] : 0.f;
    } else {
      for (int i = threadIdx.x; i < n; i += 256) {         // taps are innermost in every torch conv weight: runs of `taps` floats
        const int kk = i / taps, tp = i - kk * taps;
        tile[i] = (k0 + kk < K) ? src[r * sr + (long)(k0 + kk) * sk + tp] : 0.f;
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) tile[i] = 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kk = i % 64, tp = i / 64;
    if (k0 + kk < Kp) pack_store<T>(dst + ((size_t)tp * Rp + r) * Kp, k0 + kk, tile[kk * taps + tp]);
  }
}
// the same repack for a table of weights: block -> (descriptor by binary search on first_block, row, 64-wide k block)
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const am_pack_desc* __restrict__ descs, int nd) {
  __shared__ float tile[64 * 64];                          // taps <= 64
  int lo = 0, hi = nd;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const am_pack_desc d = descs[lo];
  const int local = blockIdx.x - d.first_block, nkb = (d.Kp + 63) / 64;
  const int r = local / nkb, k0 = (local % nkb) * 64;
  const int taps = d.taps, n = 64 * taps;
  const float* __restrict__ src = d.src;
  T* __restrict__ dst = (T*)d.dst;
  if (r < d.R) {
    if (d.stride_k == taps) {
      const float* base = src + r * d.stride_r + (long)k0 * taps;
      const int lim = (d.K - k0 < 64 ? (d.K - k0 > 0 ? d.K - k0 : 0) : 64) * taps;
      for (int i = threadIdx.x; i < n; i += 256) tile[i] = i < lim ? base[i] : 0.f;
    } else {
      for (int i = threadIdx.x; i < n; i += 256) {
        const int kk = i / taps, tp = i - kk * taps;
        tile[i] = (k0 + kk < d.K) ? src[r * d.stride_r + (long)(k0 + kk) * d.stride_k + tp] : 0.f;
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) tile[i] = 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kk = i % 64, tp = i / 64;
    if (k0 + kk < d.Kp) pack_store<T>(dst + ((size_t)tp * d.Rp + r) * d.Kp, k0 + kk, tile[kk * taps + tp]);
  }
}
// dst[r*sr + k*sk + t] (+)= src[t][r][k]
__global__ void unpack_grad_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int K, int taps, long sr, long sk, int accumulate) {
  const long n = (long)taps * R * K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {   // i walks dst order: coalesced writes
    const int tp = i % taps; long t = i / taps;
    long r, k;
    if (sr > sk) { k = t % K; r = t / K; } else { r = t % R; k = t / R; }
    const float v = src[((long)tp * R + r) * K + k];
    float* d = dst + r * sr + k * sk + tp;
    *d = accumulate ? *d + v : v;
  }
}

// dst[c] += sum over the replicas rep[r][c]
__global__ __launch_bounds__(256) void rep_reduce_kernel(const double* __restrict__ rep, int nrep, int C, float* __restrict__ dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  long long s = 0;
  for (int r = 0; r < nrep; ++r) s += ((const long long*)rep)[(size_t)r * C + c];
  dst[c] += (float)((double)s / AM_DX_FIX);
}

// per-workgroup conv partials [rows][C][2] -> sums[C][2] (double) and/or sum_accum[C] += sum
// v1 (any C): one workgroup per channel, strided 8-byte reads.
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int rows, int C, double* __restrict__ sums,
                                                               float* __restrict__ sum_accum) {
  __shared__ double sh[8];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float2 v = *(const float2*)(part + ((size_t)r * C + c) * 2);
    s1 += v.x; s2 += v.y;
  }
  s1 = warp_sum_d(s1); s2 = warp_sum_d(s2);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[4 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = sh[0] + sh[1] + sh[2] + sh[3]; s2 = sh[4] + sh[5] + sh[6] + sh[7];
    if (sums) { sums[2 * c] = s1; sums[2 * c + 1] = s2; }
    if (sum_accum) sum_accum[c] += (float)s1;
  }
}
// v2 (256 % C == 0 or C % 256 == 0): each workgroup owns a contiguous run of rows and reads them as whole coalesced rows
// (the 128^3 launches leave 32768 x 64 float2 = 17 MB: v1 touches a 64-byte sector per 8 useful bytes); a thread keeps its
// channel(s), rows are folded through LDS, one double atomic per channel per workgroup into the zeroed sums.
__global__ __launch_bounds__(256) void partials_reduce2_kernel(const float* __restrict__ part, int rows, int C, int rows_per_block,
                                                                double* __restrict__ sums, float* __restrict__ sum_accum) {
  __shared__ double sh1[256], sh2[256];
  const int t = threadIdx.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const float2* __restrict__ p2 = (const float2*)part;
  if (C <= 256) {
    const int c = t % C, rstep = 256 / C;
    double s1 = 0.0, s2 = 0.0;
    for (int r = r0 + t / C; r < r1; r += rstep) { const float2 v = p2[(size_t)r * C + c]; s1 += v.x; s2 += v.y; }
    sh1[t] = s1; sh2[t] = s2;
    __syncthreads();
    if (t < C) {
      for (int k = 1; k < rstep; ++k) { s1 += sh1[k * C + t]; s2 += sh2[k * C + t]; }
      if (sums) { atomicAdd(&sums[2 * t], s1); atomicAdd(&sums[2 * t + 1], s2); }
      if (sum_accum) atomicAdd(&sum_accum[t], (float)s1);
    }
  } else {
    for (int c = t; c < C; c += 256) {
      double s1 = 0.0, s2 = 0.0;
      for (int r = r0; r < r1; ++r) { const float2 v = p2[(size_t)r * C + c]; s1 += v.x; s2 += v.y; }
      if (sums) { atomicAdd(&sums[2 * c], s1); atomicAdd(&sums[2 * c + 1], s2); }
      if (sum_accum) atomicAdd(&sum_accum[c], (float)s1);
    }
  }
}


// Statistics plumbing of one norm in ONE launch: per-workgroup conv partials [rows][C][2] -> per-channel sums (double atomics
// into a zeroed workspace) -> the LAST workgroup to finish (ticket counter) folds them into mean / rstd / scale / shift
// (+ BatchNorm running statistics and num_batches_tracked), then re-zeroes the workspace for the next user.  Replaces
// memset + partials_reduce + norm_finalize (3 launches, ~20 us of an encoder level that computes for 100 us).
// Cross-workgroup visibility: the sums are only ever touched by device-scope atomics (performed at the memory side), the ticket
// increment follows a __threadfence(), and the finalizing workgroup reads the sums back with atomics as well.
__global__ __launch_bounds__(256) void partials_finalize_kernel(const float* __restrict__ part, int rows, int C, int rows_per_block,
                                                                double* __restrict__ ws, const double* count_ptr, double count_host,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                float* mean, float* rstd, float* scale, float* shift, float* run_mean,
                                                                float* run_var, float momentum, long* nbt, float* sum_accum,
                                                                const float* bw_mean, const float* bw_rstd, float* k0, float* k1, float* k2,
                                                                float* dgamma, float* dbeta, float* dbeta2) {
  // (k0 != nullptr: the rows are (sum g, sum g*x) of a fused norm-backward reduce, am_conv3d_nbred: the tail computes the backward
  // coefficients and the affine gradients instead of forward statistics)
  // The rows are a real stream (a 128^3 conv at batch 16 leaves 131 072 rows x 64 channels x 8 B = 67 MB): up to 2048 workgroups, a
  // thread reads 16 bytes = (sum, sumsq) of TWO channels per row, four rows in flight, and the per-workgroup sums go to one of
  // AM_FIN_REP replicas (two thousand workgroups adding into the same cache lines serialise in L2).  The first version (256
  // workgroups, one 8-byte load in flight per thread, one accumulator row) ran at 0.3 TB/s: 133 us per call, 4.9 ms per step.
  __shared__ double sh[256][4];
  __shared__ unsigned ticket_s;
  const int t = threadIdx.x, CP = C >> 1;                      // channel pairs
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const f32x4* __restrict__ p4 = (const f32x4*)part;
  double* sums = ws + (size_t)(blockIdx.x % AM_FIN_REP) * 2 * C;   // this workgroup's replica [C][2]
  unsigned* ticket = (unsigned*)(ws + (size_t)AM_FIN_REP * 2 * C);
  const int rstep = CP <= 256 ? 256 / CP : 1;
  for (int cp0 = 0; cp0 < CP; cp0 += 256) {                    // (one pass unless C > 512)
    const int cp = cp0 + (CP <= 256 ? t % CP : t), lane_r = CP <= 256 ? t / CP : 0;
    const bool live = cp < CP && lane_r < rstep;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live) {
      int r = r0 + lane_r;
      for (; r + 3 * rstep < r1; r += 4 * rstep) {
        const f32x4 v0 = p4[(size_t)r * CP + cp], v1 = p4[(size_t)(r + rstep) * CP + cp], v2 = p4[(size_t)(r + 2 * rstep) * CP + cp],
                    v3 = p4[(size_t)(r + 3 * rstep) * CP + cp];
        a0 += (double)v0[0] + (double)v1[0] + (double)v2[0] + (double)v3[0];
        a1 += (double)v0[1] + (double)v1[1] + (double)v2[1] + (double)v3[1];
        a2 += (double)v0[2] + (double)v1[2] + (double)v2[2] + (double)v3[2];
        a3 += (double)v0[3] + (double)v1[3] + (double)v2[3] + (double)v3[3];
      }
      for (; r < r1; r += rstep) { const f32x4 v = p4[(size_t)r * CP + cp]; a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3]; }
    }
    __syncthreads();
    sh[t][0] = a0; sh[t][1] = a1; sh[t][2] = a2; sh[t][3] = a3;
    __syncthreads();
    if (live && lane_r == 0) {
      for (int k = 1; k < rstep; ++k) { a0 += sh[k * CP + t][0]; a1 += sh[k * CP + t][1]; a2 += sh[k * CP + t][2]; a3 += sh[k * CP + t][3]; }
      atomicAdd(&sums[4 * cp], a0); atomicAdd(&sums[4 * cp + 1], a1); atomicAdd(&sums[4 * cp + 2], a2); atomicAdd(&sums[4 * cp + 3], a3);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the sums are memory-side atomics: acknowledged is visible
  __syncthreads();
  if (t == 0) ticket_s = atomicAdd(ticket, 1u);
  __syncthreads();
  if (ticket_s != gridDim.x - 1) return;
  // ---- last workgroup: finalize
  const double n = count_ptr ? count_ptr[0] : count_host;
  for (int c = t; c < C; c += 256) {
    double q1 = 0.0, q2 = 0.0;
    for (int r = 0; r < AM_FIN_REP; ++r) {                     // read + re-zero at the memory side, fixed order
      q1 += __longlong_as_double(atomicExch((unsigned long long*)&ws[(size_t)r * 2 * C + 2 * c], 0ull));
      q2 += __longlong_as_double(atomicExch((unsigned long long*)&ws[(size_t)r * 2 * C + 2 * c + 1], 0ull));
    }
    if (sum_accum) sum_accum[c] += (float)q1;
    if (k0) {                                                  // bsum of norm_bwd_reduce: b1 = sum g, b2 = sum g * xhat = rstd * (sum g*x - mean * sum g)
      const double b1 = q1, b2 = (double)bw_rstd[c] * (q2 - (double)bw_mean[c] * q1);
      const float gr = gamma[c] * bw_rstd[c];
      k0[c] = gr; k1[c] = gr * (float)(b1 / n); k2[c] = gr * (float)(b2 / n);
      if (dgamma) dgamma[c] += (float)b2;
      if (dbeta) dbeta[c] += (float)b1;
      if (dbeta2) dbeta2[c] += (float)b1;
      continue;
    }
    if (!gamma) continue;
    const double m = q1 / n;
    double var = q2 / n - m * m;
    if (var < 0) var = 0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m; rstd[c] = rs;
    const float scv = gamma[c] * rs;
    scale[c] = scv; shift[c] = beta[c] - (float)m * scv;
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (n / (n - 1.0)));
    }
  }
  if (t == 0) { atomicExch(ticket, 0u); if (nbt) nbt[0] += 1; }
}

// voxels per workgroup: whole passes (256/(C/EPC) voxels each), ~2048 workgroups per launch (<= 1024 for reductions)
// threads per workgroup of the linear (Walk) kernels: a voxel row of more than 256 16-byte chunks (fp32, C > 1024) takes 512
inline int walk_threads(int C, int dtype) { return C / (dtype == AM_DT_BF16 ? 8 : 4) > 256 ? 512 : 256; }
inline int pick_vpw(long nvox, int C, int dtype, bool reduction) {
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  int vpp = walk_threads(C, dtype) / (C / epc); if (vpp < 1) vpp = 1;
  long target = reduction ? 1024 : 2048;
#ifdef AM_ABLATE
  if (const char* e = getenv(reduction ? "AM_RED_TARGET" : "AM_MAP_TARGET")) target = atol(e);
#endif
  long v = (nvox + target - 1) / target;
  v = (v + vpp - 1) / vpp * vpp;
  // small tensors: at least 16 passes (64 KB of a tensor) per workgroup, but never fewer than ~256 workgroups -- thousands of
  // 4 KB workgroups are launch- and atomic-bound (their per-channel sums all land on the same accumulators)
  const long vmin = 16L * vpp, vcap = ((nvox + 255) / 256 + vpp - 1) / vpp * vpp;
  if (v < vmin) v = vmin < vcap ? vmin : vcap;
  if (v < vpp) v = vpp;
  if (v > 65536) v = 65536;
  return (int)v;
}
inline int nblk(long nvox, int vpw) { return (int)((nvox + vpw - 1) / vpw); }
inline Geo mkgeo(int dtype, bool reduction, int B, int D, int H, int W, int C, const uint8_t* mask, int bs, int fd, int fh, int fw) {
  Geo g; g.B = B; g.D = D; g.H = H; g.W = W; g.C = C; g.mask = MaskView{mask, fd, fh, fw, bs};
  g.vpw = pick_vpw((long)B * D * H * W, C, dtype, reduction);
  g.mW = (unsigned)(0x100000000ull / (unsigned)W) + 1; g.mH = (unsigned)(0x100000000ull / (unsigned)H) + 1;
  g.mD = (unsigned)(0x100000000ull / (unsigned)D) + 1;
  return g;
}
// the reciprocal-multiply decode is exact while v*d < 2^32
inline bool geo_ok(int B, int D, int H, int W) {
  const unsigned long long n = (unsigned long long)B * D * H * W;
  int m = W > H ? W : H; if (D > m) m = D;
  return n * (unsigned long long)m < 0xffffffffull;
}


// row-walk geometry of a block-sparse tensor, or false when the tensor does not qualify (rows wider than one workgroup, >255 patches
// per axis): the caller then uses the linear kernel
inline bool mkrows(RowGeo& r, int dtype, bool reduction, int B, int D, int H, int W, int C, int bs, const int* plist, int n_active) {
  if (!plist || n_active <= 0 || bs < 0 || bs > 6) return false;
  const int epc = dtype == AM_DT_BF16 ? 8 : 4;
  const int P = 1 << bs;
  const long rc = (long)P * C / epc;
  if (rc > 256 || rc < 1 || (D >> bs) > 255 || (H >> bs) > 255 || (W >> bs) > 255 || B > 255) return false;
  if (D % P || H % P || W % P) return false;
  r.B = B; r.D = D; r.H = H; r.W = W; r.C = C; r.bs = bs; r.plist = plist;
  r.rowchunks = (int)rc; r.rpp = 256 / r.rowchunks;
  r.total_rows = (long)n_active * P * P;
  // rows per workgroup: >= UNR passes, ~32 KB of the tensor, but keep >= ~1024 workgroups on big tensors / >= 1 pass on tiny ones
  const long target = reduction ? 1024 : 2048;
  long rpw = (r.total_rows + target - 1) / target;
  const long unit = (long)UNR * r.rpp;
  rpw = (rpw + unit - 1) / unit * unit;
  if (rpw < unit) rpw = unit;
  if (rpw > 64 * unit) rpw = 64 * unit;
  r.rpw = (int)rpw;
  return true;
}
inline int rows_blocks(const RowGeo& r) { return (int)((r.total_rows + r.rpw - 1) / r.rpw); }
inline int rows_threads(const RowGeo& r) { return r.rpp * r.rowchunks; }

}  // namespace

// fp32 launch of a linear (Walk) kernel: 512-thread workgroups when a voxel row has more than 256 chunks (C > 1024)
#define AM_LAUNCH_F32W(C_, KERN_, GRID_, ST_, ...) do { if ((C_) / 4 > 256) AM_LAUNCH((KERN_<float, 512>), GRID_, dim3(512), 0, ST_, __VA_ARGS__); \
                                                        else AM_LAUNCH((KERN_<float, 256>), GRID_, dim3(256), 0, ST_, __VA_ARGS__); } while (0)
#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) do { if ((dtype) == AM_DT_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)
#define CHK_C(C) do { if ((C) % 8 || (C) > 2048 || (C) <= 0) return -1; } while (0)
// entry points without a 512-thread instantiation (stem convs, projection head: Walk<T> kernels launched with 256 threads) refuse fp32 rows
// wider than 1024 channels, as layer_ops.hip does -- with 256 threads such a row leaves every thread idle and the output untouched
#define CHK_C_NARROW(C) do { CHK_C(C); if (dtype != AM_DT_BF16 && (C) > 1024) return -1; } while (0)

// backward norm kernels: activation x derivative source (saved output / recomputed pre-activation) -> the specialised instantiation
#define AM_BWD_SPEC(act_, has_out_, M_) do {                                                                        \
    if ((act_) == AM_ACT_NONE) M_(0, 0);                                                                            \
    else if ((act_) == AM_ACT_LRELU && (has_out_)) M_(1, 1);                                                        \
    else if ((act_) == AM_ACT_LRELU) M_(1, 0);                                                                      \
    else if ((act_) == AM_ACT_RELU6 && (has_out_)) M_(2, 1);                                                        \
    else if ((act_) == AM_ACT_RELU6) M_(2, 0);                                                                      \
    else M_(-1, 0); } while (0)

extern "C" {

int am_chan_stats(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                  int fw, double* sums, const int32_t* active_list, int n_active, void* stream) {
  CHK_C(C);
  Geo g = mkgeo(dtype, true, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  AM_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * NREP, st));
  RowGeo rg;
  if (mask && mkrows(rg, dtype, true, B, D, H, W, C, bshift, active_list, n_active)) {
    DISPATCH_T(dtype, AM_LAUNCH(chan_stats_rows_kernel<float>, dim3(rows_blocks(rg)), dim3(rows_threads(rg)), 0, st, (const float*)x, rg, sums),
               AM_LAUNCH(chan_stats_rows_kernel<bf16_t>, dim3(rows_blocks(rg)), dim3(rows_threads(rg)), 0, st, (const bf16_t*)x, rg, sums));
    AM_CHECK_LAUNCH();
    return 0;
  }
  if (geo_bad) return -4;
  const int nb = nblk((long)B * D * H * W, g.vpw);
  DISPATCH_T(dtype, AM_LAUNCH_F32W(C, chan_stats_kernel, dim3(nb), st, (const float*)x, g, sums),
             AM_LAUNCH(chan_stats_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, g, sums));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_mask_count(const uint8_t* mask, int n, int voxels_per_patch, double* out, void* stream) {
  AM_LAUNCH(mask_count_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mask, n, voxels_per_patch, out);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_mask_compact(const uint8_t* mask, int B, int fd, int fh, int fw, int32_t* list, int32_t* count, void* stream) {
  if (B > 255 || fd > 255 || fh > 255 || fw > 255) return -1;
  AM_LAUNCH(mask_compact_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mask, B, fd, fh, fw, list, count);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_finalize(const double* sums, int nrep, const double* count_ptr, double count_host, int C, const float* gamma,
                     const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift, float* run_mean,
                     float* run_var, float momentum, void* stream) {
  AM_LAUNCH(norm_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, nrep, count_ptr, count_host,
                     C, gamma, beta, eps, mean, rstd, scale, shift, run_mean, run_var, momentum);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_fold_running(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var, float eps,
                         float* scale, float* shift, void* stream) {
  AM_LAUNCH(norm_fold_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, run_mean,
                     run_var, eps, scale, shift);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_apply(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                  int fw, const float* scale, const float* shift, int act, const void* res, const float* stem_x,
                  const float* stem_w, const float* stem_b, const float* fill, void* y, const int32_t* active_list, int n_active,
                  void* stream) {
  CHK_C(C);
  Geo g = mkgeo(dtype, false, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  RowGeo rg;
  if (mask && !fill && !(res && stem_x) && mkrows(rg, dtype, false, B, D, H, W, C, bshift, active_list, n_active)) {
    const dim3 gr(rows_blocks(rg)), bl(rows_threads(rg));
    const int mode = res ? 1 : (stem_x ? 2 : 0);
#define AM_ROWS_APPLY(TT_, ACT_, MODE_)                                                                                        \
    AM_LAUNCH((norm_apply_rows_kernel<TT_, ACT_, MODE_>), gr, bl, 0, st, (const TT_*)x, rg, scale, shift, (const TT_*)res, stem_x, stem_w, stem_b, (TT_*)y)
#define AM_ROWS_APPLY_M(TT_, ACT_)                                                                                             \
    do { if (mode == 0) AM_ROWS_APPLY(TT_, ACT_, 0); else if (mode == 1) AM_ROWS_APPLY(TT_, ACT_, 1); else AM_ROWS_APPLY(TT_, ACT_, 2); } while (0)
#define AM_ROWS_APPLY_A(TT_)                                                                                                   \
    do { if (act == AM_ACT_LRELU) AM_ROWS_APPLY_M(TT_, AM_ACT_LRELU); else if (act == AM_ACT_RELU6) AM_ROWS_APPLY_M(TT_, AM_ACT_RELU6); \
         else AM_ROWS_APPLY_M(TT_, AM_ACT_NONE); } while (0)
    DISPATCH_T(dtype, AM_ROWS_APPLY_A(float), AM_ROWS_APPLY_A(bf16_t));
#undef AM_ROWS_APPLY_A
#undef AM_ROWS_APPLY_M
#undef AM_ROWS_APPLY
    AM_CHECK_LAUNCH();
    return 0;
  }
  if (geo_bad) return -4;
  const int nb = nblk((long)B * D * H * W, g.vpw);
#define AM_NA_(A_, O_) do {                                                                                                             \
    if (dtype == AM_DT_BF16) AM_LAUNCH((norm_apply_kernel<bf16_t, 256, A_>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, g, scale, shift, act, \
                                (const bf16_t*)res, stem_x, stem_w, stem_b, fill, (bf16_t*)y);                                           \
    else if (C / 4 > 256) AM_LAUNCH((norm_apply_kernel<float, 512>), dim3(nb), dim3(512), 0, st, (const float*)x, g, scale, shift, act,   \
                                (const float*)res, stem_x, stem_w, stem_b, fill, (float*)y);                                             \
    else AM_LAUNCH((norm_apply_kernel<float, 256, A_>), dim3(nb), dim3(256), 0, st, (const float*)x, g, scale, shift, act,                \
                                (const float*)res, stem_x, stem_w, stem_b, fill, (float*)y); } while (0)
  AM_BWD_SPEC(act, false, AM_NA_);
#undef AM_NA_
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_bwd_reduce(int dtype, const void* dout, const void* out, const void* x, int B, int D, int H, int W, int C,
                       const uint8_t* mask, int bshift, int fd, int fh, int fw, const float* mean, const float* rstd, int act,
                       int fill, double* bsum, const float* pre_scale, const float* pre_shift, const int32_t* active_list,
                       int n_active, const double* count_ptr, double count_host, const float* gamma, float* k0, float* k1, float* k2,
                       float* dgamma, float* dbeta, float* dtoken, float* dbeta2, void* stream) {
  CHK_C(C);
  if (!out && act != AM_ACT_NONE && (!pre_scale || !pre_shift)) return -1;
  Geo g = mkgeo(dtype, true, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  BwdFin fin{};
  if (k0) {                                  // fused finalize: bsum is a zero workspace [AM_NREP][C][3] doubles + one ticket word, left zero
    if (!gamma || !k1 || !k2) return -1;
    fin.ticket = (unsigned*)(bsum + (size_t)3 * C * NREP);
    fin.count_ptr = count_ptr; fin.count_host = count_host; fin.gamma = gamma; fin.rstd = rstd;
    fin.k0 = k0; fin.k1 = k1; fin.k2 = k2; fin.dgamma = dgamma; fin.dbeta = dbeta; fin.dtoken = dtoken; fin.dbeta2 = dbeta2;
  } else {
    AM_HIP(hipMemsetAsync(bsum, 0, sizeof(double) * 3 * C * NREP, st));
  }
  RowGeo rg;
  if (mask && !fill && mkrows(rg, dtype, true, B, D, H, W, C, bshift, active_list, n_active)) {
#define AM_RR_(A_, O_)                                                                                                                  \
    DISPATCH_T(dtype,                                                                                                                    \
               AM_LAUNCH((norm_bwd_reduce_rows_kernel<float, A_, O_>), dim3(rows_blocks(rg)), dim3(rows_threads(rg)), 0, st, (const float*)dout, \
                         (const float*)out, (const float*)x, rg, mean, rstd, act, bsum, pre_scale, pre_shift, fin),                     \
               AM_LAUNCH((norm_bwd_reduce_rows_kernel<bf16_t, A_, O_>), dim3(rows_blocks(rg)), dim3(rows_threads(rg)), 0, st, (const bf16_t*)dout, \
                         (const bf16_t*)out, (const bf16_t*)x, rg, mean, rstd, act, bsum, pre_scale, pre_shift, fin))
    AM_BWD_SPEC(act, out != nullptr, AM_RR_);
#undef AM_RR_
    AM_CHECK_LAUNCH();
    return 0;
  }
  if (geo_bad) return -4;
  const int nb = nblk((long)B * D * H * W, g.vpw);
#define AM_RL_(A_, O_) do {                                                                                                             \
    if (dtype == AM_DT_BF16) AM_LAUNCH((norm_bwd_reduce_kernel<bf16_t, 256, A_, O_>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dout,   \
                                (const bf16_t*)out, (const bf16_t*)x, g, mean, rstd, act, fill, bsum, pre_scale, pre_shift, fin);       \
    else if (C / 4 > 256) AM_LAUNCH((norm_bwd_reduce_kernel<float, 512>), dim3(nb), dim3(512), 0, st, (const float*)dout,                \
                                (const float*)out, (const float*)x, g, mean, rstd, act, fill, bsum, pre_scale, pre_shift, fin);         \
    else AM_LAUNCH((norm_bwd_reduce_kernel<float, 256, A_, O_>), dim3(nb), dim3(256), 0, st, (const float*)dout,                         \
                                (const float*)out, (const float*)x, g, mean, rstd, act, fill, bsum, pre_scale, pre_shift, fin); } while (0)
  AM_BWD_SPEC(act, out != nullptr, AM_RL_);
#undef AM_RL_
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_bwd_finalize(const double* bsum, const double* count_ptr, double count_host, int C, const float* gamma,
                         const float* rstd, float* k0, float* k1, float* k2, float* dgamma, float* dbeta, float* dtoken,
                         float* dbeta2, void* stream) {
  AM_LAUNCH(norm_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, bsum, count_ptr,
                     count_host, C, gamma, rstd, k0, k1, k2, dgamma, dbeta, dtoken, dbeta2);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_norm_bwd_apply(int dtype, const void* dout, const void* out, const void* x, int B, int D, int H, int W, int C,
                      const uint8_t* mask, int bshift, int fd, int fh, int fw, const float* mean, const float* rstd,
                      const float* k0, const float* k1, const float* k2, int act, void* dx, void* dres, float* dxsum_accum,
                      double* dxsum_scratch, const float* pre_scale, const float* pre_shift, const int32_t* active_list, int n_active,
                      int scratch_is_zero_workspace, void* stream) {
  CHK_C(C);
  if (!out && act != AM_ACT_NONE && (!pre_scale || !pre_shift)) return -1;
  Geo g = mkgeo(dtype, false, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  RowGeo rg;
  const bool rows = mask && mkrows(rg, dtype, false, B, D, H, W, C, bshift, active_list, n_active);
  if (!rows && geo_bad) return -4;
  const int nb = rows ? rows_blocks(rg) : nblk((long)B * D * H * W, g.vpw);
  // bias-gradient sums: every workgroup adds C floats; on ONE accumulator 2048 workgroups serialise (measured 400 us on an
  // 8 MB tensor), so they go to AM_DXREP replicas that a 1-block kernel folds afterwards
  if (dxsum_accum && !dxsum_scratch) return -1;              // the bias-gradient sums always go through the fp64 replicas
  const bool rep = dxsum_accum != nullptr;
  double* dxs = rep ? dxsum_scratch : nullptr;
  const int nrep = AM_DXREP;
  // scratch_is_zero_workspace: [AM_DXREP][C] floats + one ticket word, zero on entry and left zero: the last workgroup folds the
  // replicas into dxsum_accum itself (no memset launch before, no fold launch after)
  const bool fused = rep && scratch_is_zero_workspace;
  float* dx_accum = fused ? dxsum_accum : nullptr;
  unsigned* dx_ticket = fused ? (unsigned*)(dxsum_scratch + (size_t)AM_DXREP * C) : nullptr;
  if (rep && !fused) AM_HIP(hipMemsetAsync(dxsum_scratch, 0, sizeof(double) * AM_DXREP * C, st));
  if (rows) {
#define AM_AR_(A_, O_)                                                                                                                  \
    DISPATCH_T(dtype,                                                                                                                    \
               AM_LAUNCH((norm_bwd_apply_rows_kernel<float, A_, O_>), dim3(nb), dim3(rows_threads(rg)), 0, st, (const float*)dout, (const float*)out, \
                         (const float*)x, rg, mean, rstd, k0, k1, k2, act, (float*)dx, (float*)dres, dxs, nrep, pre_scale, pre_shift, dx_accum, dx_ticket), \
               AM_LAUNCH((norm_bwd_apply_rows_kernel<bf16_t, A_, O_>), dim3(nb), dim3(rows_threads(rg)), 0, st, (const bf16_t*)dout, (const bf16_t*)out, \
                         (const bf16_t*)x, rg, mean, rstd, k0, k1, k2, act, (bf16_t*)dx, (bf16_t*)dres, dxs, nrep, pre_scale, pre_shift, dx_accum, dx_ticket))
    AM_BWD_SPEC(act, out != nullptr, AM_AR_);
#undef AM_AR_
  } else {
#define AM_AL_(A_, O_) do {                                                                                                             \
    if (dtype == AM_DT_BF16) AM_LAUNCH((norm_bwd_apply_kernel<bf16_t, 256, A_, O_>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, \
                                (const bf16_t*)x, g, mean, rstd, k0, k1, k2, act, (bf16_t*)dx, (bf16_t*)dres, dxs, nrep, pre_scale, pre_shift, dx_accum, dx_ticket); \
    else if (C / 4 > 256) AM_LAUNCH((norm_bwd_apply_kernel<float, 512>), dim3(nb), dim3(512), 0, st, (const float*)dout, (const float*)out, \
                                (const float*)x, g, mean, rstd, k0, k1, k2, act, (float*)dx, (float*)dres, dxs, nrep, pre_scale, pre_shift, dx_accum, dx_ticket); \
    else AM_LAUNCH((norm_bwd_apply_kernel<float, 256, A_, O_>), dim3(nb), dim3(256), 0, st, (const float*)dout, (const float*)out,        \
                                (const float*)x, g, mean, rstd, k0, k1, k2, act, (float*)dx, (float*)dres, dxs, nrep, pre_scale, pre_shift, dx_accum, dx_ticket); } while (0)
    AM_BWD_SPEC(act, out != nullptr, AM_AL_);
#undef AM_AL_
  }
  AM_CHECK_LAUNCH();
  if (rep && !fused) { AM_LAUNCH(rep_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dxsum_scratch, AM_DXREP, C, dxsum_accum); AM_CHECK_LAUNCH(); }
  return 0;
}

int am_chan_sum(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                int fw, float* out_accum, void* stream) {
  CHK_C(C);
  Geo g = mkgeo(dtype, true, B, D, H, W, C, mask, bshift, fd, fh, fw);
  if (mask && !geo_ok(B, D, H, W)) return -4;
  hipStream_t st = (hipStream_t)stream;
  // small tensors (the densify projection of the coarsest level: its bias gradient): ONE workgroup, so the fp32 sum has a fixed order
  // (with several workgroups the per-workgroup sums arrive as fp32 atomics in any order)
  if ((long)B * D * H * W * C <= (4L << 20)) { g.vpw = (int)((long)B * D * H * W); const int vpp = walk_threads(C, dtype) / (C / (dtype == AM_DT_BF16 ? 8 : 4)) > 0 ? walk_threads(C, dtype) / (C / (dtype == AM_DT_BF16 ? 8 : 4)) : 1; g.vpw = (g.vpw + vpp - 1) / vpp * vpp; }
  const int nb = nblk((long)B * D * H * W, g.vpw);
  DISPATCH_T(dtype, AM_LAUNCH_F32W(C, chan_sum_kernel, dim3(nb), st, (const float*)x, g, out_accum),
             AM_LAUNCH(chan_sum_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, g, out_accum));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_add(int dtype, const void* a, const void* b, void* y, long n_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t nchunk = (size_t)n_elems / (dtype == AM_DT_BF16 ? 8 : 4);
  int nb = (int)((nchunk + 255) / 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  DISPATCH_T(dtype, AM_LAUNCH(add_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)y, nchunk),
             AM_LAUNCH(add_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, nchunk));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_stem_conv_fwd(int dtype, const float* x, int B, int D, int H, int W, int C, int ksize, const uint8_t* mask, int bshift,
                     int fd, int fh, int fw, const float* w, const float* bias, void* y, float* partials,
                     const int32_t* active_list, int n_active, int* partial_rows_written, void* stream) {
  CHK_C_NARROW(C);
  if (ksize != 1 && ksize != 3) return -2;
  Geo g = mkgeo(dtype, false, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  if (!mask || bshift != 4 || D % 16 || H % 16 || W % 16) return -2;      // stage-0 tensor: 16^3 patches
  if (partial_rows_written) *partial_rows_written = B * (D / SBD) * (H / SBH) * (W / SBW);
  if (dtype == AM_DT_BF16 && ksize == 3 && active_list && n_active > 0 && (C == 32 || C == 64 || C == 96) && B <= 255 && fd <= 255 && fh <= 255 && fw <= 255) {
    const MaskView mv{mask, fd, fh, fw, bshift};
    if (C == 32) AM_LAUNCH(stem_conv_mfma_kernel<2>, dim3(n_active), dim3(256), 0, st, x, D, H, W, C, mv, active_list, w, bias, (bf16_t*)y, partials);
    else if (C == 64) AM_LAUNCH(stem_conv_mfma_kernel<4>, dim3(n_active), dim3(256), 0, st, x, D, H, W, C, mv, active_list, w, bias, (bf16_t*)y, partials);
    else AM_LAUNCH(stem_conv_mfma_kernel<6>, dim3(n_active), dim3(256), 0, st, x, D, H, W, C, mv, active_list, w, bias, (bf16_t*)y, partials);
    AM_CHECK_LAUNCH();
    if (partial_rows_written) *partial_rows_written = n_active;
    return 0;
  }
  if (geo_bad) return -4;
  const int nb = B * (D / SBD) * (H / SBH) * (W / SBW);
  const int pad_ = ksize / 2;
  size_t fl = (size_t)C * ksize * ksize * ksize;
  if (partials && fl < (size_t)256 * 8 * 2) fl = (size_t)256 * 8 * 2;    // the statistics fold reuses the weight area
  const size_t sm = sizeof(float) * (fl + (size_t)(SBD + 2 * pad_) * (SBH + 2 * pad_) * (SBW + 2 * pad_));
  if (ksize == 3) {
    DISPATCH_T(dtype, AM_LAUNCH((stem_conv_fwd_kernel<float, 3>), dim3(nb), dim3(256), sm, st, x, g, w, bias, (float*)y, partials),
               AM_LAUNCH((stem_conv_fwd_kernel<bf16_t, 3>), dim3(nb), dim3(256), sm, st, x, g, w, bias, (bf16_t*)y, partials));
  } else {
    DISPATCH_T(dtype, AM_LAUNCH((stem_conv_fwd_kernel<float, 1>), dim3(nb), dim3(256), sm, st, x, g, w, bias, (float*)y, partials),
               AM_LAUNCH((stem_conv_fwd_kernel<bf16_t, 1>), dim3(nb), dim3(256), sm, st, x, g, w, bias, (bf16_t*)y, partials));
  }
  AM_CHECK_LAUNCH();
  return 0;
}

int am_stem_conv_wgrad(int dtype, const float* x, const void* dy, int B, int D, int H, int W, int C, int ksize,
                       const uint8_t* mask, int bshift, int fd, int fh, int fw, float* dw_accum, float* db_accum,
                       const int32_t* active_list, int n_active, float* det_workspace, long det_workspace_floats, void* stream) {
  CHK_C_NARROW(C);
  if (ksize != 1 && ksize != 3) return -2;
  Geo g = mkgeo(dtype, true, B, D, H, W, C, mask, bshift, fd, fh, fw);
  const bool geo_bad = mask && !geo_ok(B, D, H, W);          // only the LINEAR kernels decode voxel indices by reciprocal multiply
  hipStream_t st = (hipStream_t)stream;
  if (!mask || bshift != 4 || D % 16 || H % 16 || W % 16) return -2;
  if (dtype == AM_DT_BF16 && active_list && n_active > 0 && (C == 32 || C == 64 || C == 96) && B <= 255 && fd <= 255 && fh <= 255 && fw <= 255) {
    const MaskView mv{mask, fd, fh, fw, bshift};
    int nwg = n_active < 1024 ? n_active : 1024;                // persistent: four workgroups per CU, one atomic flush each
    const long det_row = (long)C * (ksize * ksize * ksize + 1);
    if (det_workspace) {                                        // deterministic mode: as many workgroups as the workspace has rows for
        if (det_workspace_floats < det_row) return -6;
        if (nwg > det_workspace_floats / det_row) nwg = (int)(det_workspace_floats / det_row);
    }
    const size_t sm = (size_t)(18 * 18 * 18 + 8) * 4 + (size_t)256 * (2 * C + 32);
#define AM_STEM_WG(NS_, K_)                                                                                                           \
    {                                                                                                                                 \
      auto kern = stem_wgrad_mfma_kernel<NS_, K_>;                                                                                    \
      static PerDeviceOnce cap;                                                                                                       \
      cap.run([&](int) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); }); \
      AM_LAUNCH(kern, dim3(nwg), dim3(256), sm, st, x, (const bf16_t*)dy, D, H, W, mv, active_list, n_active, dw_accum, db_accum, det_workspace);     \
    }
    if (ksize == 3) { if (C == 32) AM_STEM_WG(2, 3) else if (C == 64) AM_STEM_WG(4, 3) else AM_STEM_WG(6, 3) }
    else { if (C == 32) AM_STEM_WG(2, 1) else if (C == 64) AM_STEM_WG(4, 1) else AM_STEM_WG(6, 1) }
#undef AM_STEM_WG
    AM_CHECK_LAUNCH();
    if (det_workspace) {
      AM_LAUNCH(stem_wgrad_fold_kernel, dim3((unsigned)((det_row + 255) / 256)), dim3(256), 0, st, det_workspace, nwg, C, ksize * ksize * ksize, dw_accum, db_accum);
      AM_CHECK_LAUNCH();
    }
    return 0;
  }
  if (geo_bad) return -4;
  int nb = B * (D / SBD) * (H / SBH) * (W / SBW);
  if (nb > 1024) nb = 1024;                                   // persistent workgroups: one atomic flush each
  const int pad_ = ksize / 2;
  const size_t sm = sizeof(float) * ((size_t)C * (ksize * ksize * ksize + 1) + (size_t)(SBD + 2 * pad_) * (SBH + 2 * pad_) * (SBW + 2 * pad_));
  DISPATCH_T(dtype,
             AM_LAUNCH(stem_conv_wgrad_kernel<float>, dim3(nb), dim3(256), sm, st, x, (const float*)dy, g, ksize, dw_accum, db_accum),
             AM_LAUNCH(stem_conv_wgrad_kernel<bf16_t>, dim3(nb), dim3(256), sm, st, x, (const bf16_t*)dy, g, ksize, dw_accum, db_accum));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_proj_fwd(int dtype, const void* x, long nvox, int C, const float* w, const float* b, const float* pre_scale, const float* pre_shift,
                float* rec, void* stream) {
  CHK_C_NARROW(C);
  if ((pre_scale == nullptr) != (pre_shift == nullptr)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((nvox + 255) / 256);
  DISPATCH_T(dtype, AM_LAUNCH(proj_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, nvox, C, w, b, pre_scale, pre_shift, rec),
             AM_LAUNCH(proj_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, nvox, C, w, b, pre_scale, pre_shift, rec));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_proj_norm_bwd(int dtype, const void* x, const float* drec, long nvox, int C, const float* w, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, const float* scale, double* workspace, float* coef, void* dx,
                     float* dgamma_accum, float* dbeta_accum, float* dw_accum, float* db_accum, void* stream) {
  CHK_C_NARROW(C);
  if (!x || !drec || !w || !gamma || !beta || !mean || !rstd || !scale || !workspace || !coef || !dx || !dgamma_accum || !dbeta_accum ||
      !dw_accum || !db_accum || nvox <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  ProjNormFin fin{(unsigned*)(workspace + (size_t)NREP * (C + 1)), (double)nvox, gamma, beta, rstd, scale, w,
                  coef, coef + C, coef + 2 * C, dgamma_accum, dbeta_accum, dw_accum, db_accum};
  const int vr = pick_vpw(nvox, C, dtype, true), va = pick_vpw(nvox, C, dtype, false);
  DISPATCH_T(dtype, AM_LAUNCH(proj_norm_bwd_reduce_kernel<float>, dim3(nblk(nvox, vr)), dim3(256), 0, st, (const float*)x, drec, nvox, C, vr, mean, workspace, fin),
             AM_LAUNCH(proj_norm_bwd_reduce_kernel<bf16_t>, dim3(nblk(nvox, vr)), dim3(256), 0, st, (const bf16_t*)x, drec, nvox, C, vr, mean, workspace, fin));
  AM_CHECK_LAUNCH();
  DISPATCH_T(dtype, AM_LAUNCH(proj_norm_bwd_apply_kernel<float>, dim3(nblk(nvox, va)), dim3(256), 0, st, (const float*)x, drec, nvox, C, va, mean, rstd, coef, coef + C, coef + 2 * C, (float*)dx),
             AM_LAUNCH(proj_norm_bwd_apply_kernel<bf16_t>, dim3(nblk(nvox, va)), dim3(256), 0, st, (const bf16_t*)x, drec, nvox, C, va, mean, rstd, coef, coef + C, coef + 2 * C, (bf16_t*)dx));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_proj_bwd(int dtype, const void* x, const float* drec, long nvox, int C, const float* w, void* dx, float* dw_accum,
                float* db_accum, float* det_workspace, long det_workspace_floats, void* stream) {
  CHK_C_NARROW(C);
  hipStream_t st = (hipStream_t)stream;
  const int vpw = pick_vpw(nvox, C, dtype, true);
  const int nb = nblk(nvox, vpw);
  if (det_workspace && det_workspace_floats < (long)nb * (C + 1)) return -6;
  DISPATCH_T(dtype, AM_LAUNCH(proj_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, drec, nvox, C, vpw, w, (float*)dx, dw_accum, db_accum, det_workspace),
             AM_LAUNCH(proj_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, drec, nvox, C, vpw, w, (bf16_t*)dx, dw_accum, db_accum, det_workspace));
  AM_CHECK_LAUNCH();
  if (det_workspace) {
    AM_LAUNCH(rows_fold_kernel, dim3((C + 1 + 255) / 256), dim3(256), 0, st, det_workspace, nb, C + 1, C, dw_accum, db_accum);
    AM_CHECK_LAUNCH();
  }
  return 0;
}

int am_partials_reduce(const float* partials, int rows, int C, double* sums, float* sum_accum, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((256 % C == 0 || C % 256 == 0) && rows >= 64) {
    const int rstep = C <= 256 ? 256 / C : 1;
    int nb = rows / (rstep * 16); nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
    const int rpb = ((rows + nb - 1) / nb + rstep - 1) / rstep * rstep;
    nb = (rows + rpb - 1) / rpb;
    if (sums) AM_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st));
    AM_LAUNCH(partials_reduce2_kernel, dim3(nb), dim3(256), 0, st, partials, rows, C, rpb, sums, sum_accum);
    AM_CHECK_LAUNCH();
    return 0;
  }
  AM_LAUNCH(partials_reduce_kernel, dim3(C), dim3(256), 0, st, partials, rows, C, sums, sum_accum);
  AM_CHECK_LAUNCH();
  return 0;
}


int am_partials_finalize(const float* partials, int rows, int C, double* workspace, const double* count_ptr, double count_host,
                         const float* gamma, const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                         float* run_mean, float* run_var, float momentum, long* num_batches_tracked, float* sum_accum, void* stream) {
  if (C <= 0 || C > 4096 || rows <= 0 || !workspace) return -1;
  if (gamma && (!beta || !mean || !rstd || !scale || !shift)) return -1;
  if (C % 2) return -1;
  const int cp = C / 2, rstep = cp <= 256 ? 256 / cp : 1;
  int nb = rows / (rstep * 8); nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  const int rpb = ((rows + nb - 1) / nb + rstep - 1) / rstep * rstep;
  nb = (rows + rpb - 1) / rpb;
  AM_LAUNCH(partials_finalize_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, partials, rows, C, rpb, workspace, count_ptr, count_host,
            gamma, beta, eps, mean, rstd, scale, shift, run_mean, run_var, momentum, num_batches_tracked, sum_accum,
            (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  AM_CHECK_LAUNCH();
  return 0;
}

// rows of am_conv3d_nbred (sum g, sum g*x per workgroup and channel) -> k0 / k1 / k2 of am_norm_bwd_apply + the affine gradients:
// replaces am_norm_bwd_reduce (+ finalize) for a norm whose output gradient came out of that launch.  Same workspace contract as
// am_partials_finalize (zero on entry, left zero).
int am_norm_bwd_from_partials(const float* partials, int rows, int C, double* workspace, const double* count_ptr, double count_host,
                              const float* gamma, const float* mean, const float* rstd, float* k0, float* k1, float* k2,
                              float* dgamma, float* dbeta, float* dbeta2, void* stream) {
  if (C <= 0 || C > 4096 || C % 2 || rows <= 0 || !workspace || !gamma || !mean || !rstd || !k0 || !k1 || !k2) return -1;
  const int cp = C / 2, rstep = cp <= 256 ? 256 / cp : 1;
  int nb = rows / (rstep * 8); nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  const int rpb = ((rows + nb - 1) / nb + rstep - 1) / rstep * rstep;
  nb = (rows + rpb - 1) / rpb;
  AM_LAUNCH(partials_finalize_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, partials, rows, C, rpb, workspace, count_ptr, count_host,
            gamma, (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
            0.f, (long*)nullptr, (float*)nullptr, mean, rstd, k0, k1, k2, dgamma, dbeta, dbeta2);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_pack_weight(int dtype, const float* src, void* dst, int R, int K, int taps, long stride_r, long stride_k, int Rp, int Kp,
                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (Rp < R || Kp < K) return -1;
  dim3 grid(Rp, (Kp + 63) / 64);
  const size_t sm = sizeof(float) * 64 * taps;
  if (dtype == AM_DT_F32S) {
    if (Kp % 16) return -1;
    AM_LAUNCH(pack_weight_kernel<f32s_t>, grid, dim3(256), sm, st, src, (f32s_t*)dst, R, K, taps, stride_r, stride_k, Rp, Kp);
  } else
  DISPATCH_T(dtype, AM_LAUNCH(pack_weight_kernel<float>, grid, dim3(256), sm, st, src, (float*)dst, R, K, taps, stride_r, stride_k, Rp, Kp),
             AM_LAUNCH(pack_weight_kernel<bf16_t>, grid, dim3(256), sm, st, src, (bf16_t*)dst, R, K, taps, stride_r, stride_k, Rp, Kp));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_pack_weights_batched(int dtype, const am_pack_desc* descs, int ndesc, int total_blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (ndesc <= 0 || total_blocks <= 0) return 0;
  if (dtype == AM_DT_F32S) AM_LAUNCH(pack_weights_batched_kernel<f32s_t>, dim3(total_blocks), dim3(256), 0, st, descs, ndesc);
  else
  DISPATCH_T(dtype, AM_LAUNCH(pack_weights_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, st, descs, ndesc),
             AM_LAUNCH(pack_weights_batched_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, descs, ndesc));
  AM_CHECK_LAUNCH();
  return 0;
}

int am_split_bf16(const float* x, void* hi, void* lo, long n, void* stream) {
  if (n % 4) return -1;
  long nb = (n / 4 + 255) / 256; if (nb > 16384) nb = 16384;
  AM_LAUNCH(split_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)hi, (bf16_t*)lo, n / 4);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_unpack_grad(const float* src_packed, float* dst, int R, int K, int taps, long stride_r, long stride_k, int accumulate,
                   void* stream) {
  const long n = (long)taps * R * K;
  int nb = (int)((n + 255) / 256); if (nb > 8192) nb = 8192;
  AM_LAUNCH(unpack_grad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src_packed, dst, R, K, taps, stride_r, stride_k, accumulate);
  AM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
