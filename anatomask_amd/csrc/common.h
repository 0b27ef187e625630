// Shared device helpers for the AnatoMask gfx950 kernels.
// Layout convention everywhere: activations are channels-last [B][D][H][W][C]
// (C contiguous), dtype T = float or bf16 (stored as uint16), C % 8 == 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AM_DT_F32 0
#define AM_DT_BF16 1
#define AM_DT_F32S 2   // fp32 storage, matrix-core products from bf16 hi / lo splits of both operands (see f32s_t)

// fp32 storage with SPLIT products (AM_DT_F32S): tensors are fp32 in memory, exactly as AM_DT_F32; the channel contraction runs on the bf16
// matrix cores on x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits, |x - hi - lo| <= 2^-17 |x|) with all four partial
// products (hi hi + hi lo + lo hi + lo lo) accumulated in fp32.  The exact mode's v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate;
// this one at 1/4 (two 16x16x32 bf16 instructions per 16 fp32 channels).  Same size and layout in memory as float.
struct f32s_t { float v; };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN (quiet)
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct TT<f32s_t> {
  static constexpr int EPC = 4;
  static __device__ __forceinline__ float ld(const f32s_t* p) { return p->v; }
  static __device__ __forceinline__ void st(f32s_t* p, float v) { p->v = v; }
};
template <> struct TT<bf16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte chunk <-> floats
template <typename T> __device__ __forceinline__ void chunk_to_f(const u32x4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f<float>(const u32x4& c, float* f) {
  f[0] = __uint_as_float(c[0]); f[1] = __uint_as_float(c[1]); f[2] = __uint_as_float(c[2]); f[3] = __uint_as_float(c[3]);
}
template <> __device__ __forceinline__ void chunk_to_f<bf16_t>(const u32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(c[i] << 16); f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ u32x4 f_to_chunk(const float* f);
template <> __device__ __forceinline__ u32x4 f_to_chunk<float>(const float* f) {
  u32x4 c; c[0] = __float_as_uint(f[0]); c[1] = __float_as_uint(f[1]); c[2] = __float_as_uint(f[2]); c[3] = __float_as_uint(f[3]); return c;
}
template <> __device__ __forceinline__ u32x4 f_to_chunk<bf16_t>(const float* f) {
  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN -- the same values as f2bf() at 1/10 of its instruction count
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bfx2_;
  u32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_{f[2 * i], f[2 * i + 1]}), bfx2_));
  return c;
}

// One "chunk MMA": D(16 cout x 16 voxel) += A(16 x KC) * B(KC x 16) where each lane supplies one
// 16-byte chunk of A (row = lane&15) and of B (col = lane&15), chunk index g = lane>>4.
//   bf16: KC = 32, one v_mfma_f32_16x16x32_bf16 (lane's 8 elements are k = 8g..8g+7)
//   f32 : KC = 16, four v_mfma_f32_16x16x4_f32 (step s uses element s of every lane: k = 4g+s);
//         exact f32 fma chain, same rate as the f32 VALU peak (MI355X_MICROARCH.md "FP32-input MFMA").
template <typename T> __device__ __forceinline__ f32x4 mma_chunk(const u32x4& a, const u32x4& b, f32x4 acc);
template <> __device__ __forceinline__ f32x4 mma_chunk<bf16_t>(const u32x4& a, const u32x4& b, f32x4 acc) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma_chunk<float>(const u32x4& a, const u32x4& b, f32x4 acc) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]), acc, 0, 0, 0);
  return acc;
}

// f32s_t: four floats -> their bf16 hi parts (2 dwords) and lo parts (2 dwords); v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ void split4_bf16(const u32x4& c, unsigned (&hi)[2], unsigned (&lo)[2]) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bfx2_;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x0 = __uint_as_float(c[2 * i]), x1 = __uint_as_float(c[2 * i + 1]);
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_{x0, x1}), bfx2_));
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    hi[i] = h;
    lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_{r0, r1}), bfx2_));
  }
}
// eight floats -> one 16-byte MFMA fragment of their bf16 hi parts and one of their lo parts (element j in half-word j)
__device__ __forceinline__ void split8_bf16(const float (&v)[8], u32x4& hi, u32x4& lo) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bfx2_;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_{v[2 * i], v[2 * i + 1]}), bfx2_));
    const float r0 = v[2 * i] - __uint_as_float(h << 16), r1 = v[2 * i + 1] - __uint_as_float(h & 0xffff0000u);
    hi[i] = h;
    lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_{r0, r1}), bfx2_));
  }
}
// One chunk MMA of the split mode: the 64-byte LDS row of 16 fp32 channels is [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] (bf16); lane group g
// supplies chunk g of the A row, chunk (g & 1) of the B row for b1 (hi) and chunk 2 + (g & 1) for b2 (lo):
//   sum_g A_g . B1_g = (hiA + loA) . hiB,   sum_g A_g . B2_g = (hiA + loA) . loB
__device__ __forceinline__ f32x4 mma_split(const u32x4& a, const u32x4& b1, const u32x4& b2, f32x4 acc) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b1), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b2), acc, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane of the row: 4 v_add_f32 with DPP operands
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS instruction per step;
// the statistics epilogue of conv_igemm issued 128 of them per lane (12 % of the dominant launch).
__device__ __forceinline__ float row16_sum(float v) {
#define AM_DPP_ADD(CTRL) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true))
  AM_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  AM_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  AM_DPP_ADD(0x141);   // row_half_mirror
  AM_DPP_ADD(0x140);   // row_mirror
#undef AM_DPP_ADD
  return v;
}

__device__ __forceinline__ float warp_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double warp_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-activity lookup: mask is uint8 [B][fd][fh][fw]; a voxel (d,h,w) at a resolution whose
// blocks are (1<<bs) voxels wide is active iff mask[b][d>>bs][h>>bs][w>>bs] != 0.
struct MaskView {
  const uint8_t* m;  // nullptr => everything active
  int fd, fh, fw, bs;
  __device__ __forceinline__ bool active(int b, int d, int h, int w) const {
    if (!m) return true;
    return m[((b * fd + (d >> bs)) * fh + (h >> bs)) * fw + (w >> bs)] != 0;
  }
  // byte of the patch that holds (d,h,w), or of patch 0 when !inrange: an UNCONDITIONAL load, so a staging plan can issue the
  // lookups of all its rows back to back and wait once (the branchy form serialises one global round trip per row)
  __device__ __forceinline__ uint8_t peek(int b, int d, int h, int w, bool inrange) const {
    const int i = inrange ? ((b * fd + (d >> bs)) * fh + (h >> bs)) * fw + (w >> bs) : 0;
    return m[i];
  }
};

// Run `f` once per DEVICE of this process (not once per process): kernel attributes (hipFuncSetAttribute) and device properties are
// per device, and one process may drive several (one process per GPU is the deployment, but nothing here may assume it).
// Idempotent work only: two threads may both run f for the same device.
struct PerDeviceOnce {
  unsigned long long done = 0;
  template <typename F> __host__ void run(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return;
    f(dev);
    __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
  }
};

#define AM_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
// launch with a clean error slate: hipGetLastError() is per-thread sticky and torch's own probing calls
// can leave a benign error behind that would otherwise be mis-attributed to our launch
// an API call inside an entry point that promises "0 or a hipError_t": its failure is the entry point's return value
#define AM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)
#define AM_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
// The kernel-selection stages (conv_rw / conv_gather / conv_k3 / conv_k3t / conv_wgk3 _launch) answer 1 = served, 0 = the shape does not
// qualify.  hipErrorInvalidValue IS 1 (what a rejected launch configuration returns), so a failure inside a stage is -(1000 + hipError_t):
// never mistaken for "served", and distinct from the entry points' small negative argument errors.
#define AM_STAGE_ERR(e) (-(1000 + (int)(e)))
#define AM_CHECK_LAUNCH_STAGE() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return AM_STAGE_ERR(e_); } while (0)
// opt a kernel in to 160 KB of dynamic LDS once per device; a refused opt-in is the stage's (sticky) failure, not a silent fall-through
#define AM_LDS_OPTIN_STAGE(kern) do { \
    static PerDeviceOnce once_; static int err_ = 0; \
    once_.run([&](int) { hipError_t e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                         if (e_ != hipSuccess) err_ = (int)e_; (void)hipGetLastError(); }); \
    if (err_) return AM_STAGE_ERR(err_); } while (0)
