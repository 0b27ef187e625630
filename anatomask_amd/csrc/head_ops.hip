// The decoder's tail in EVAL mode, collapsed (gfx950): the last UNetBlock ends conv3x3x3(C -> C/2, no bias) -> BatchNorm3d on running
// statistics -> Conv3d(C/2 -> 1, k1, bias) (P/decoder3D.py:20-22,51,61).  Without batch statistics that chain is LINEAR in the block's
// ReLU6 output r, so the EMA teacher's pass (P/pretrain_AntoMask.py:421-425: only the per-patch l2 of the MASKED patches is used) and the
// plain-SparK validation pass (P/pretrain.py:426-441) evaluate it as ONE C -> 1 stencil
//     rec[q] = b_eff + sum_t sum_ci W_eff[t][ci] * r[q + t][ci],   W_eff[t][ci] = sum_c w_proj[c] * scale[c] * W2[c][ci][t],
//     b_eff = b_proj + sum_c w_proj[c] * shift[c]                  (scale / shift: the folded eval-mode BatchNorm)
// on the needed 16^3 patches only: 3.5 kflop per voxel instead of 110, bound by reading r once (+ halo) from HBM -- the LDS-tiled stencil
// `north_star` describes.  am_head_fold makes W_eff / b_eff (fp32, every step: the teacher's weights move with the EMA);
// am_head_stencil evaluates it.
//
// Kernel: one 256-thread workgroup per needed patch, walking its 18 haloed input planes z.  Per plane the 18 x 18 voxels x C channels are
// contracted with the 27 taps on the matrix cores as a 1x1 convolution, P_z[voxel][tap] = sum_ci r_z[voxel][ci] W_eff[tap][ci] (A = the taps
// as two 16-row tiles, B = 16 voxels: the B fragment of a lane is ONE 16-byte global load of 8 (bf16) / 4 (fp32) channels of its voxel --
// no LDS staging of r; out-of-volume voxels are the buffer load's zero fill), P_z goes to LDS, and thread (h, w) of the patch adds the 9
// (th, tw) neighbours of each d-tap into its three rolling output planes: rec[z - 1] is complete after plane z.  fp32 accumulation
// throughout.  bf16 storage: W_eff enters as hi + lo bf16 parts (two MFMAs: r is exact in bf16, the folded weights keep 16 significant
// bits); fp32 storage (both product modes): the exact v_mfma_f32_16x16x4_f32.  Loads run two tiles ahead of the MFMAs in a register ring;
// with 3 workgroups per CU that keeps > 50 KB per CU in flight.  The raw per-patch l2 against the input volume (the teacher's ranking
// signal) is reduced in the same kernel when asked for.
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

constexpr int HP = 16;                       // patch edge (the decoder's output grid is the input grid: 16^3 patches)
constexpr int HE = HP + 2;                   // haloed edge
constexpr int HNV = HE * HE;                 // 324 voxels per haloed plane
constexpr int HNT = (HNV + 15) / 16;         // 21 tiles of 16 voxels
constexpr int HSLOT = 6;                     // tiles per wave and plane (4 waves x 6 >= 21; slots beyond tile 20 are empty)
constexpr int HPS = 33;                      // floats per voxel row of P (27 taps + padding: conflict-free stencil reads)

struct HsArgs {
  const void* r; const float* weff; const float* beff;
  const int* plist; int npatch;
  int B, D, H, W, C;
  float* rec; const float* inp; float* l2; int fd, fh, fw;
};

template <typename T> struct HK;             // channels per 16-byte chunk MMA
template <> struct HK<bf16_t> { static constexpr int KC = 32; static constexpr int NW = 2; };   // W fragments per (chunk, row tile): hi, lo
template <> struct HK<float> { static constexpr int KC = 16; static constexpr int NW = 1; };

template <typename T, int NCH>
__global__ __launch_bounds__(256) void head_stencil_kernel(HsArgs a) {
  constexpr int KC = HK<T>::KC, NW = HK<T>::NW;
  constexpr int NSET = NCH <= 6 ? 3 : 2;     // register ring: loads run NSET - 1 tiles ahead
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* P = (float*)lds;                                        // [HNT * 16][HPS]
  unsigned char* Wl = lds + HNT * 16 * HPS * 4;                  // [NCH][2 row tiles][NW][64 lanes][16 B]
  float* red = (float*)(Wl + NCH * 2 * NW * 1024);               // 4 floats (no static LDS in front of the dynamic carve)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n16 = lane & 15;
  const int C = a.C;

  // ---- the folded weights as MFMA A fragments in LDS: lane (row = tap 16 rt + n16, chunk part g)
  for (int f = tid; f < NCH * 2 * 64; f += 256) {
    const int ln = f & 63, rt = (f >> 6) & 1, ch = f >> 7;
    const int tap = 16 * rt + (ln & 15), gg = ln >> 4;
    if constexpr (sizeof(T) == 2) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = tap < 27 ? a.weff[tap * C + ch * 32 + gg * 8 + e] : 0.f;
      float lo[8];
      const u32x4 hi = f_to_chunk<bf16_t>(w);
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[2 * e] = w[2 * e] - __uint_as_float(hi[e] << 16); lo[2 * e + 1] = w[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u); }
      *(u32x4*)(Wl + (((ch * 2 + rt) * 2 + 0) * 64 + ln) * 16) = hi;
      *(u32x4*)(Wl + (((ch * 2 + rt) * 2 + 1) * 64 + ln) * 16) = f_to_chunk<bf16_t>(lo);
    } else {
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = tap < 27 ? a.weff[tap * C + ch * 16 + gg * 4 + e] : 0.f;
      *(u32x4*)(Wl + ((ch * 2 + rt) * 64 + ln) * 16) = f_to_chunk<float>(w);
    }
  }

  const int pk = __builtin_amdgcn_readfirstlane(a.plist[blockIdx.x]);
  const int b = (pk >> 24) & 255, d0 = ((pk >> 16) & 255) * HP, h0 = ((pk >> 8) & 255) * HP, w0 = (pk & 255) * HP;
  const float beff = a.beff[0];

  // ---- per-lane load offsets of the wave's 6 tile slots, relative to row 0 of an input plane (lane constants over the 18 planes)
  constexpr unsigned OOB = 0x80000000u;
  unsigned off[HSLOT];
#pragma unroll
  for (int s = 0; s < HSLOT; ++s) {
    const int v = (wave + 4 * s) * 16 + n16;
    const int y = v / HE, x = v - y * HE;
    const int hh = h0 - 1 + y, ww = w0 - 1 + x;
    const bool ok = v < HNV && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
    off[s] = ok ? (unsigned)((((hh * a.W + ww) * C) + g * (KC / 4)) * (int)sizeof(T)) : OOB;
  }
  const T* rg = (const T*)a.r;
  const size_t plane = (size_t)a.H * a.W * C;
  const int plane_bytes = (int)(plane * sizeof(T));

  u32x4 ring[NSET][NCH];
  // loads of (plane z, slot s) into ring set `set`: a plane outside the volume reads zeros
  auto issue = [&](const int z, const int s, const int set) __attribute__((always_inline)) {
    const bool zin = z >= 0 && z < a.D;
    const T* base = rg + ((size_t)b * a.D + (zin ? z : 0)) * plane;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, plane_bytes, 0x00020000);
    const unsigned o = zin ? off[s] : OOB;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      ring[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, c * KC * (int)sizeof(T), 0));
  };

  // ---- prologue: the first NSET - 1 slots of the first plane
  const int zf = d0 - 1;
#pragma unroll
  for (int s = 0; s < NSET - 1; ++s) issue(zf, s, s);
  __syncthreads();                                               // the weight fragments are in LDS

  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;                      // output planes z - 1, z, z + 1 of the current input plane z
  float err = 0.f;
  const int oh = tid >> 4, ow = tid & 15;
  for (int zi = 0; zi < HE; ++zi) {
    const int z = d0 - 1 + zi;
    const bool zin = z >= 0 && z < a.D;
#pragma unroll
    for (int s = 0; s < HSLOT; ++s) {
      // keep the ring full: slot s + NSET - 1 of this plane, or the first slots of the next one
      {
        const int s2 = s + NSET - 1;
        if (s2 < HSLOT) issue(z, s2, s2 % NSET);
        else if (zi + 1 < HE) issue(z + 1, s2 - HSLOT, s2 % NSET);
      }
      // (the compiler's own counted `s_waitcnt vmcnt` in front of the first use of ring[s % NSET] leaves the younger sets in flight)
      const int tile = wave + 4 * s;
      if (zin && tile < HNT) {
        f32x4 dacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int p = 0; p < NW; ++p) {
              const u32x4 wf = *(const u32x4*)(Wl + (((c * 2 + rt) * NW + p) * 64 + lane) * 16);
              dacc[rt] = mma_chunk<T>(wf, ring[s % NSET][c], dacc[rt]);
            }
        // D: lane = voxel n16 of the tile, taps 16 rt + 4 g + i
        float* prow = P + (tile * 16 + n16) * HPS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          prow[4 * g + i] = dacc[0][i];
          if (16 + 4 * g + i < 27) prow[16 + 4 * g + i] = dacc[1][i];
        }
      }
    }
    __syncthreads();                                             // P_z complete
    if (zin) {
      // tap (td, th, tw) at index (td * 3 + th) * 3 + tw reads r[q + (td - 1, th - 1, tw - 1)]: plane z feeds output plane z - td + 1
      const float* p0 = P + (oh * HE + ow) * HPS;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) {
          const float* pp = p0 + (th * HE + tw) * HPS + th * 3 + tw;
          s0 += pp[18]; s1 += pp[9]; s2 += pp[0];
        }
      acc0 += s0; acc1 += s1; acc2 += s2;
    }
    // output plane z - 1 has seen its three input planes
    const int zo = z - 1;
    if (zo >= d0 && zo < d0 + HP) {
      const float v = acc0 + beff;
      const size_t o = (((size_t)b * a.D + zo) * a.H + h0 + oh) * a.W + w0 + ow;
      if (a.rec) a.rec[o] = v;
      if (a.l2) { const float e = v - a.inp[o]; err += e * e; }
    }
    acc0 = acc1; acc1 = acc2; acc2 = 0.f;
    __syncthreads();                                             // the stencil reads are done before the next plane overwrites P
  }
  if (a.l2) {
    err = warp_sum(err);
    if (lane == 0) red[wave] = err;
    __syncthreads();
    if (tid == 0) a.l2[(size_t)b * a.fd * a.fh * a.fw + ((d0 / HP) * a.fh + h0 / HP) * a.fw + w0 / HP] = (red[0] + red[1] + red[2] + red[3]) * (1.f / 4096.f);
  }
}

// W_eff[t][ci] = sum_c wproj[c] * scale[c] * W2[c][ci][t];  b_eff = bproj + sum_c wproj[c] * shift[c].  One thread per (t, ci), c in order.
__global__ void head_fold_kernel(const float* __restrict__ w2, int cmid, int cin, const float* __restrict__ scale, const float* __restrict__ shift,
                                 const float* __restrict__ wproj, const float* __restrict__ bproj, float* __restrict__ weff, float* __restrict__ beff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 27 * cin) {
    const int t = i / cin, ci = i - t * cin;
    float s = 0.f;
    for (int c = 0; c < cmid; ++c) s += wproj[c] * scale[c] * w2[((size_t)c * cin + ci) * 27 + t];
    weff[i] = s;
  }
  if (i == 0) {
    float s = bproj[0];
    for (int c = 0; c < cmid; ++c) s += wproj[c] * shift[c];
    beff[0] = s;
  }
}

template <typename T, int NCH> int hs_launch(const HsArgs& a, hipStream_t st) {
  constexpr int NW = HK<T>::NW;
  const size_t lds = (size_t)HNT * 16 * HPS * 4 + (size_t)NCH * 2 * NW * 1024 + 16;
  auto kern = head_stencil_kernel<T, NCH>;
  static PerDeviceOnce once;
  once.run([&](int) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); });
  AM_LAUNCH(kern, dim3((unsigned)a.npatch), dim3(256), lds, st, a);
  AM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" {

int am_head_fold(const float* w2, int cmid, int cin, const float* scale, const float* shift, const float* wproj, const float* bproj,
                 float* weff, float* beff, void* stream) {
  if (cmid < 1 || cin < 1) return -1;
  AM_LAUNCH(head_fold_kernel, dim3((27 * cin + 255) / 256), dim3(256), 0, (hipStream_t)stream, w2, cmid, cin, scale, shift, wproj, bproj, weff, beff);
  AM_CHECK_LAUNCH();
  return 0;
}

int am_head_stencil_supported(int dtype, int C) {
  if (dtype == AM_DT_BF16) return (C == 32 || C == 64 || C == 128 || C == 192) ? 1 : 0;
  if (dtype == AM_DT_F32 || dtype == AM_DT_F32S) return (C == 32 || C == 64 || C == 128 || C == 192) ? 1 : 0;
  return 0;
}

int am_head_stencil(int dtype, const void* r, int B, int D, int H, int W, int C, const float* weff, const float* beff,
                    const int32_t* patch_list, int n_patches, float* rec, const float* inp, float* l2, void* stream) {
  if (!am_head_stencil_supported(dtype, C) || D % HP || H % HP || W % HP || B > 255 || D / HP > 255 || H / HP > 255 || W / HP > 255) return -1;
  if (l2 && !inp) return -1;
  if ((size_t)H * W * C * (dtype == AM_DT_BF16 ? 2 : 4) >= 0x7fffff00ull) return -1;      // one plane per buffer descriptor
  if (n_patches <= 0) return 0;
  HsArgs a;
  a.r = r; a.weff = weff; a.beff = beff; a.plist = patch_list; a.npatch = n_patches;
  a.B = B; a.D = D; a.H = H; a.W = W; a.C = C; a.rec = rec; a.inp = inp; a.l2 = l2; a.fd = D / HP; a.fh = H / HP; a.fw = W / HP;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AM_DT_BF16) {
    switch (C) {
      case 32: return hs_launch<bf16_t, 1>(a, st);
      case 64: return hs_launch<bf16_t, 2>(a, st);
      case 128: return hs_launch<bf16_t, 4>(a, st);
      default: return hs_launch<bf16_t, 6>(a, st);
    }
  }
  switch (C) {
    case 32: return hs_launch<float, 2>(a, st);
    case 64: return hs_launch<float, 4>(a, st);
    case 128: return hs_launch<float, 8>(a, st);
    default: return hs_launch<float, 12>(a, st);
  }
}

}  // extern "C"
