// Weight gradient of the gather convolution (conv k1/k3 stride 1/2 and ConvTranspose3d k4 s2 p1)
// for gfx950 matrix cores, channels-last.  Autograd counterpart of conv_igemm.hip.
//
//   dW[widx_t][cy][cx] = sum_q  dY[q*OS + p_g][cy] * X[q*IS + shift_t][cx]        (t in tap-group g)
//
// i.e. a GEMM with M = cy (dY channels), N = cx (X channels), K = voxels.  Taps are processed in
// groups that share the dY operand: conv k3 -> 3 groups of 9 taps (same d-shift: the X brick needs
// no d-halo), ConvT -> the 8 output-parity classes of 8 taps.  A workgroup owns (group, 64 cy, 64 cx),
// walks a strided set of q-bricks, stages dY-brick and haloed X-brick channels-last in LDS and
// contracts over voxels.  The contraction index (voxel) is NOT the contiguous one in memory, so bf16
// fragments are fetched with ds_read_b64_tr_b16 (hardware transpose read: 4 voxels x 16 channels per
// 16-lane group); f32 fragments are plain 4-byte LDS reads for v_mfma_f32_16x16x4_f32.
// Partial sums leave the workgroup as f32 atomics into the packed [tap][cy][cx] gradient
// (64-byte runs per 16 lanes; order-dependent in the last bits, like any split-K atomic reduce).
#include "common.h"
#include "../../include/anatomask_hip.h"

namespace {

struct WgArgs {
  const void* x; const void* dy; float* dw;
  int B, Dx, Hx, Wx, Cx, Dy, Hy, Wy, Cy;
  int OS, IS, ngroup;
  int nbd, nbh, nbw;
  int tap_begin[9];
  int taps[64];
  int mind[8], minh[8], minw[8];
  int ed[8], eh[8], ew[8];
  int pofs[8];                // parity of dY voxels per group: pd<<2|ph<<1|pw
  MaskView x_mask, y_mask;
};

constexpr int CT = 64, KT = 64, TG = 9;

template <typename T> struct Frag;   // A/B operand of one k-step for a 16-channel subtile
template <> struct Frag<bf16_t> { typedef s16x8 type; static constexpr int KSTEP = 32; };
template <> struct Frag<float> { typedef float type; static constexpr int KSTEP = 4; };

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <typename T, int BD, int BH, int BW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  constexpr int EPC = TT<T>::EPC;
  constexpr int MV = BD * BH * BW;
  constexpr int KSTEP = Frag<T>::KSTEP;
  constexpr int RSY = CT * sizeof(T) + 16;               // LDS row strides (bytes)
  constexpr int RSX = KT * sizeof(T) + 16;
  constexpr int CPR = CT / EPC;                          // 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ldsY = lds;
  unsigned char* ldsX = lds + MV * RSY;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int grp = blockIdx.z;
  const int ncxt = (a.Cx + KT - 1) / KT;
  const int cy0 = (blockIdx.y / ncxt) * CT, cx0 = (blockIdx.y % ncxt) * KT;
  const int pd = (a.pofs[grp] >> 2) & 1, ph = (a.pofs[grp] >> 1) & 1, pw = a.pofs[grp] & 1;
  const int ED = a.ed[grp], EH = a.eh[grp], EW = a.ew[grp];
  const int nvox = ED * EH * EW;
  const int tb = a.tap_begin[grp], ntap = a.tap_begin[grp + 1] - tb;
  if (ntap == 0) return;

  f32x4 acc[TG][4];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int tapoff[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    tapoff[t] = 0;
    if (t < ntap) {
      const int tp = a.taps[tb + t];
      const int sd = (tp & 15) - 8, sh = ((tp >> 4) & 15) - 8, sw = ((tp >> 8) & 15) - 8;
      tapoff[t] = ((sd - a.mind[grp]) * EH + (sh - a.minh[grp])) * EW + (sw - a.minw[grp]);
    }
  }

  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ yg = (const T*)a.dy;
  const int nbrick = a.B * a.nbd * a.nbh * a.nbw;

  for (int brick = blockIdx.x; brick < nbrick; brick += gridDim.x) {
    int bid = brick;
    const int bw_ = bid % a.nbw; bid /= a.nbw;
    const int bh_ = bid % a.nbh; bid /= a.nbh;
    const int bd_ = bid % a.nbd; const int b = bid / a.nbd;
    const int q0d = bd_ * BD, q0h = bh_ * BH, q0w = bw_ * BW;

    // ---- stage dY brick (zero where out of range / inactive); skip the brick if it is all zero ----
    __syncthreads();
    int any = 0;
    for (int idx = tid; idx < MV * CPR; idx += 256) {
      const int v = idx / CPR, c = idx % CPR;
      const int od = (q0d + v / (BW * BH)) * a.OS + pd, oh = (q0h + (v / BW) % BH) * a.OS + ph, ow = (q0w + v % BW) * a.OS + pw;
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      const int cy = cy0 + c * EPC;
      if (od < a.Dy && oh < a.Hy && ow < a.Wy && a.y_mask.active(b, od, oh, ow)) {
        any = 1;
        if (cy < a.Cy) val = *(const u32x4*)(yg + (((size_t)(b * a.Dy + od) * a.Hy + oh) * a.Wy + ow) * a.Cy + cy);
      }
      *(u32x4*)(ldsY + v * RSY + c * 16) = val;
    }
    if (!__syncthreads_or(any)) continue;
    // ---- stage haloed X brick ----
    const int i0d = q0d * a.IS + a.mind[grp], i0h = q0h * a.IS + a.minh[grp], i0w = q0w * a.IS + a.minw[grp];
    for (int idx = tid; idx < nvox * CPR; idx += 256) {
      const int e = idx / CPR, c = idx % CPR;
      const int id = i0d + e / (EW * EH), ih = i0h + (e / EW) % EH, iw = i0w + e % EW;
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      const int cx = cx0 + c * EPC;
      if (cx < a.Cx && id >= 0 && id < a.Dx && ih >= 0 && ih < a.Hx && iw >= 0 && iw < a.Wx && a.x_mask.active(b, id, ih, iw))
        val = *(const u32x4*)(xg + (((size_t)(b * a.Dx + id) * a.Hx + ih) * a.Wx + iw) * a.Cx + cx);
      *(u32x4*)(ldsX + e * RSX + c * 16) = val;
    }
    __syncthreads();

    // ---- contract over the brick's voxels ----
    for (int ks = 0; ks < MV / KSTEP; ++ks) {
      if constexpr (sizeof(T) == 2) {
        const int q = (lane >> 2) & 3, p = lane & 3;
        const int v1 = ks * 32 + g * 8 + q, v2 = v1 + 4;
        const int xv1 = ((v1 / (BW * BH)) * a.IS * EH + ((v1 / BW) % BH) * a.IS) * EW + (v1 % BW) * a.IS;
        const int xv2 = ((v2 / (BW * BH)) * a.IS * EH + ((v2 / BW) % BH) * a.IS) * EW + (v2 % BW) * a.IS;
        s16x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const s16x4 lo = tr_read(ldsY + v1 * RSY + (16 * i + 4 * p) * 2);
          const s16x4 hi = tr_read(ldsY + v2 * RSY + (16 * i + 4 * p) * 2);
          af[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          if (t < ntap) {
            const s16x4 lo = tr_read(ldsX + (xv1 + tapoff[t]) * RSX + (16 * wave + 4 * p) * 2);
            const s16x4 hi = tr_read(ldsX + (xv2 + tapoff[t]) * RSX + (16 * wave + 4 * p) * 2);
            const s16x8 bf = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, af[i]), __builtin_bit_cast(bfx8, bf), acc[t][i], 0, 0, 0);
          }
        }
      } else {
        const int v = ks * 4 + g;
        const int xv = ((v / (BW * BH)) * a.IS * EH + ((v / BW) % BH) * a.IS) * EW + (v % BW) * a.IS;
        float af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const float*)(ldsY + v * RSY + (16 * i + r16) * 4);
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          if (t < ntap) {
            const float bf = *(const float*)(ldsX + (xv + tapoff[t]) * RSX + (16 * wave + r16) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[t][i], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- flush: D row = cy 4g+r, col = cx r16 ----
  const int cx = cx0 + 16 * wave + r16;
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    if (t < ntap && cx < a.Cx) {
      const int widx = a.taps[tb + t] >> 12;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cy = cy0 + 16 * i + 4 * g + r;
          if (cy < a.Cy) atomicAdd(a.dw + ((size_t)widx * a.Cy + cy) * a.Cx + cx, acc[t][i][r]);
        }
    }
  }
}

template <typename T, int BD, int BH, int BW>
int launch(WgArgs& a, size_t maxvox, int split, hipStream_t st) {
  auto kern = conv_wgrad_kernel<T, BD, BH, BW>;
  const size_t lds = (size_t)BD * BH * BW * (CT * sizeof(T) + 16) + maxvox * (KT * sizeof(T) + 16);
  if (lds > 160 * 1024) return -3;
  static size_t attr_lds = 48 * 1024;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) attr_lds = lds;
    (void)hipGetLastError();
  }
  dim3 grid(split, ((a.Cy + CT - 1) / CT) * ((a.Cx + KT - 1) / KT), a.ngroup);
  AM_LAUNCH(kern, grid, dim3(256), lds, st, a);
  AM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int am_conv3d_wgrad(int mode, int dtype, int ksize, int stride, const void* x, const void* dy, float* dw_packed,
                               int B, int Dx, int Hx, int Wx, int Cx, int Dy, int Hy, int Wy, int Cy,
                               const uint8_t* x_mask, int x_bshift, const uint8_t* y_mask, int y_bshift,
                               int fd, int fh, int fw, void* stream) {
  if (Cx % 8 || Cy % 8) return -1;
  WgArgs a;
  a.x = x; a.dy = dy; a.dw = dw_packed;
  a.B = B; a.Dx = Dx; a.Hx = Hx; a.Wx = Wx; a.Cx = Cx; a.Dy = Dy; a.Hy = Hy; a.Wy = Wy; a.Cy = Cy;
  a.x_mask = MaskView{x_mask, fd, fh, fw, x_bshift};
  a.y_mask = MaskView{y_mask, fd, fh, fw, y_bshift};
  const int k = ksize;
  int bd, bh, bw;
  const bool bf = dtype == AM_DT_BF16;
  if (mode == AM_CONV_FWD) {
    a.OS = 1; a.IS = stride;
    if (k != 1 && k != 3) return -2;
    a.ngroup = k;                                          // one group per d-tap
  } else if (mode == AM_CONVT_FWD) {
    if (k != 4 || stride != 2) return -2;
    a.OS = 2; a.IS = 1; a.ngroup = 8;
  } else return -2;
  if (a.IS == 2) { bd = bf ? 4 : 2; bh = 4; bw = 4; } else { bd = bf ? 4 : 2; bh = 8; bw = 8; }
  const int pad = (mode == AM_CONVT_FWD) ? 1 : k / 2;
  int n = 0; size_t maxvox = 0;
  for (int gI = 0; gI < a.ngroup; ++gI) {
    a.tap_begin[gI] = n;
    const int p[3] = {(gI >> 2) & 1, (gI >> 1) & 1, gI & 1};
    a.pofs[gI] = (mode == AM_CONVT_FWD) ? gI : 0;
    int mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; bool first = true;
    for (int td = 0; td < k; ++td) for (int th = 0; th < k; ++th) for (int tw = 0; tw < k; ++tw) {
      const int t[3] = {td, th, tw};
      int s[3]; bool ok = true;
      if (mode == AM_CONV_FWD) { if (td != gI) continue; for (int d = 0; d < 3; ++d) s[d] = t[d] - pad; }
      else for (int d = 0; d < 3; ++d) { const int num = p[d] + pad - t[d]; if (num & 1) { ok = false; break; } s[d] = num / 2; }
      if (!ok) continue;
      a.taps[n++] = (s[0] + 8) | ((s[1] + 8) << 4) | ((s[2] + 8) << 8) | ((td * k * k + th * k + tw) << 12);
      for (int d = 0; d < 3; ++d) { if (first || s[d] < mn[d]) mn[d] = s[d]; if (first || s[d] > mx[d]) mx[d] = s[d]; }
      first = false;
    }
    if (n - a.tap_begin[gI] > TG) return -2;
    a.mind[gI] = mn[0]; a.minh[gI] = mn[1]; a.minw[gI] = mn[2];
    a.ed[gI] = (bd - 1) * a.IS + (mx[0] - mn[0]) + 1;
    a.eh[gI] = (bh - 1) * a.IS + (mx[1] - mn[1]) + 1;
    a.ew[gI] = (bw - 1) * a.IS + (mx[2] - mn[2]) + 1;
    const size_t v = (size_t)a.ed[gI] * a.eh[gI] * a.ew[gI];
    if (v > maxvox) maxvox = v;
  }
  for (int gI = a.ngroup; gI <= 8; ++gI) a.tap_begin[gI] = n;
  const int Qd = (Dy + a.OS - 1) / a.OS, Qh = (Hy + a.OS - 1) / a.OS, Qw = (Wy + a.OS - 1) / a.OS;
  a.nbd = (Qd + bd - 1) / bd; a.nbh = (Qh + bh - 1) / bh; a.nbw = (Qw + bw - 1) / bw;
  const int nbrick = B * a.nbd * a.nbh * a.nbw;
  // enough workgroups to fill 256 CUs, few enough that the atomic flush stays small
  const int tiles = ((Cy + CT - 1) / CT) * ((Cx + KT - 1) / KT) * a.ngroup;
  int split = (1024 + tiles - 1) / tiles;
  if (split > nbrick) split = nbrick;
  if (split < 1) split = 1;
  hipStream_t st = (hipStream_t)stream;
  if (bf) return a.IS == 2 ? launch<bf16_t, 4, 4, 4>(a, maxvox, split, st) : launch<bf16_t, 4, 8, 8>(a, maxvox, split, st);
  return a.IS == 2 ? launch<float, 2, 4, 4>(a, maxvox, split, st) : launch<float, 2, 8, 8>(a, maxvox, split, st);
}
