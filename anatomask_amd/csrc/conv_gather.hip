// k3 convolution (forward stride 1 / 2, data gradient stride 1) of a block-sparse tensor whose patches are SMALLER than a brick, bf16, gfx950.
// Ref: P/encoder3D.py:12-15 (sp_conv_forward: dense conv of the zero-filled tensor, output masked) at the deep encoder levels, where
// a patch of the mask is 4^3 or 2^3 voxels or ONE voxel (STUNet levels 2-4: 128-1536 channels on 8^3-48^3 grids).
//
// conv_igemm.hip tiles the q grid with 256-voxel bricks: at these levels a brick spans 8-256 patches, is live for a fraction of its
// voxels and computes all of them (2.5x the active work at mask 0.6, 3.3x at 0.7; STUNet-L 1024->1024 on a 10^3 grid ran at 130
// TFLOP/s).  Here the rows of the implicit GEMM are the ACTIVE voxels, enumerated from the active-patch list (am_mask_compact):
//   * a workgroup (4 waves) owns 128 consecutive active voxels x 64 or 128 output channels; wave w owns two 16-voxel subtiles,
//   * the source operand needs no LDS: the B fragment of v_mfma_f32_16x16x32_bf16 is, per lane, 16 bytes = 8 consecutive channels of
//     ONE voxel -- a gather load straight into the operand registers, from the voxel's neighbour under the tap (offset = voxel base +
//     a scalar per tap), or out of range (hardware zero fill) when that neighbour is outside the volume or in an inactive patch.
//     Which of its 27 neighbours exist is one bit each per voxel, looked up ONCE at workgroup start,
//   * weights go through LDS as in conv_igemm (every wave needs all of them): a stage = one tap x 256 input channels for 64 output
//     channels or x 128 for 128 (32-34 KB, double buffered, 528- / 272-byte rows: conflict-free 16-byte fragment reads), 64 MFMAs per
//     wave between barriers,
//   * software pipeline: the next stage's weights and source fragments are in flight while this stage's MFMAs issue,
//   * epilogue as conv_igemm: bias, bf16 rounding, 16-byte stores (crow() channel order), optional per-workgroup (sum, sum of
//     squares) row for the norm that follows.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/anatomask_hip.h"
#include "conv_plan.h"

using namespace amconv;

namespace {

struct GaArgs {
  const int* plist;      // active-patch list: b << 24 | pd << 16 | ph << 8 | pw; nullptr = a dense tensor (rows = all voxels of the enumeration grid)
  int M;                 // rows per class: active voxels = n_active << (3 * bs), or B * D * H * W of the enumeration grid
  int bs;                // patch edge = 1 << bs voxels of the enumeration grid
  int ntile, ny;         // voxel tiles, output-channel tiles
  int nkg;               // stages per tap = Cin / KG
  int S;                 // source stride: 1, or 2 (strided forward conv: the source grid is twice the enumeration grid, its patches twice as wide)
  int OS, ncls;          // output stride 2 (transposed conv, strided data gradient): the enumeration grid is the COARSE (source) grid, every
                         //   coarse voxel q yields the 8 output voxels 2q + parity; a workgroup computes ONE parity class (its own taps)
  int Ed, Eh, Ew;        // enumeration grid
  int wide;              // dense tensors with a tap window wider than 3^3 (transposed conv's data gradient: 4^3, shifts -1..2): range tests instead of the 27 bits
  int cbeg[9];           // taps of class c: [cbeg[c], cbeg[c + 1])
  int shift[64];         // per tap: (ud + 1) | (uh + 1) << 2 | (uw + 1) << 4 | widx << 8   (u = shift of the source voxel on the source grid)
};

typedef bf16_t T;
constexpr unsigned OOB = 0x80000000u;

template <int NS, int VS, int KSL>
__global__ __launch_bounds__(256, 2) void conv_gather_kernel(ConvArgs a, GaArgs r) {
  constexpr int NT = 16 * NS, KG = 32 * KSL, MT = 64 * VS;
  constexpr int WROW = KG * 2 + 16;                      // LDS row stride of a weight row: 16 rows x 16 B of one fragment read hit 16 distinct bank groups
  constexpr int WBUF = NT * WROW;
  constexpr int WIT = NT * KSL * 4 / 256;                // 16-byte weight chunks per thread and stage
  static_assert(NT * KSL * 4 % 256 == 0, "every thread stages exactly WIT chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  // channel tiles of a voxel tile are neighbours in blockIdx.x on one XCD (they gather the same source rows): see conv_igemm.hip
  const int t8 = blockIdx.x >> 3, nsib = r.ny * r.ncls;
  const int sib = t8 % nsib, cls = sib % r.ncls, ytile = sib / r.ncls, tile = (t8 / nsib) * 8 + (blockIdx.x & 7);
  if (tile >= r.ntile) return;
  const int co0 = ytile * NT;
  const int t0 = r.cbeg[cls], ntap = r.cbeg[cls + 1] - t0;    // this class's taps
  const int OS = r.OS, pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  const int D = a.Di, H = a.Hi, W = a.Wi;                 // source grid
  const int bs = r.bs, pm = (1 << bs) - 1, S = r.S;
  const int cinB = a.Cin * 2;

  // ---- this lane's voxels (one per subtile): linear index, and which of the 27 neighbours exist (in range AND in an active patch)
  int vlin[VS], vout[VS];                                  // linear index of the tap-centre voxel in the source grid / of the output voxel
  int cdhw[VS];                                            // its coordinates (d | h << 10 | w << 20): the range test of taps outside the 3^3 bit mask (r.wide)
  bool valid[VS];
  unsigned nb[VS];
  {
    uint8_t mb[VS][27];
    unsigned inr[VS];
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      const int i = tile * MT + wave * (16 * VS) + j * 16 + r16;
      valid[j] = i < r.M;
      int b, d, h, w;
      if (r.plist) {
        const int pk = r.plist[valid[j] ? i >> (3 * bs) : 0];
        const int loc = i & ((1 << (3 * bs)) - 1);
        b = (pk >> 24) & 255;
        d = (((pk >> 16) & 255) << bs) | (loc >> (2 * bs)); h = (((pk >> 8) & 255) << bs) | ((loc >> bs) & pm); w = ((pk & 255) << bs) | (loc & pm);
      } else {                                             // dense: row i = voxel i of the enumeration grid
        const int ii = valid[j] ? i : 0;
        w = ii % r.Ew; h = (ii / r.Ew) % r.Eh; d = (ii / (r.Ew * r.Eh)) % r.Ed; b = ii / (r.Ew * r.Eh * r.Ed);
      }
      vout[j] = ((b * a.Do + OS * d + pd) * a.Ho + OS * h + ph) * a.Wo + OS * w + pw;
      vlin[j] = ((b * D + S * d) * H + S * h) * W + S * w;
      cdhw[j] = valid[j] ? (S * d) | ((S * h) << 10) | ((S * w) << 20) : 0x3fffffff;   // (an invalid row fails every range test)
      inr[j] = 0;
#pragma unroll
      for (int c = 0; c < 27; ++c) {                       // unconditional mask-byte loads, all in flight at once
        const int nd = S * d + c / 9 - 1, nh = S * h + (c / 3) % 3 - 1, nw = S * w + c % 3 - 1;
        const bool ok = valid[j] && (unsigned)nd < (unsigned)D && (unsigned)nh < (unsigned)H && (unsigned)nw < (unsigned)W;
        mb[j][c] = a.in_mask.m ? a.in_mask.peek(b, nd, nh, nw, ok) : (uint8_t)1;
        inr[j] |= ok ? 1u << c : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      nb[j] = 0;
#pragma unroll
      for (int c = 0; c < 27; ++c) nb[j] |= mb[j][c] ? 1u << c : 0u;
      nb[j] &= inr[j];
    }
  }

  const size_t xbytes = (size_t)a.B * D * H * W * a.Cin * 2;     // (host: < 2 GB)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const int wtapB = a.Coutp * a.Cinp * 2;                 // bytes per weight tap slice

  // weight staging plan: chunk -> (cout row, 16-byte chunk of the stage's KG channels)
  unsigned wsrc[WIT];
  int wdst[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int idx = tid + it * 256;
    const int row = idx / (KSL * 4), ch = idx % (KSL * 4);
    wsrc[it] = (unsigned)(((co0 + crow(row)) * a.Cinp) * 2 + ch * 16);
    wdst[it] = row * WROW + ch * 16;
  }

  f32x4 acc[NS][VS];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int j = 0; j < VS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 fa[KSL][VS], fb[KSL][VS], wreg[WIT];
  const int nstage = ntap * r.nkg;

  // loads of stage (tap t, channel group kg): weights -> wreg, source fragments -> F.  ok false: nothing (zeros, no traffic)
  auto issue = [&](u32x4 (&F)[KSL][VS], int t, int kg, bool ok) __attribute__((always_inline)) {
    const int sh = r.shift[t0 + (ok ? t : 0)];
    const int ud = (sh & 3) - 1, uh = ((sh >> 2) & 3) - 1, uw = ((sh >> 4) & 3) - 1, widx = sh >> 8;
    const int bit = (ud + 1) * 9 + (uh + 1) * 3 + (uw + 1);   // (taps with a shift of +2 exist only in the range-checked form)
    const int dlinB = ((ud * H + uh) * W + uw) * cinB;    // (uniform) byte shift of the tap's neighbour row
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      wreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (ok && !AM_DBG(a, 4)) ? wsrc[it] : OOB, widx * wtapB + kg * (KG * 2), 0));
#pragma unroll
    for (int j = 0; j < VS; ++j) {
      bool here;
      if (r.wide) {                                        // (uniform) dense tensor, window wider than 3^3: the neighbour exists iff it is in range
        const int nd = (cdhw[j] & 1023) + ud, nh = ((cdhw[j] >> 10) & 1023) + uh, nw = (cdhw[j] >> 20) + uw;
        here = (unsigned)nd < (unsigned)D && (unsigned)nh < (unsigned)H && (unsigned)nw < (unsigned)W;
      } else here = ((nb[j] >> bit) & 1u) != 0;
      const unsigned off = (ok && here && !AM_DBG(a, 2)) ? (unsigned)(vlin[j] * cinB + dlinB + g * 16) : OOB;
#pragma unroll
      for (int sl = 0; sl < KSL; ++sl)
        F[sl][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, kg * (KG * 2) + sl * 64, 0));
    }
  };
  auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) *(u32x4*)(lds + buf * WBUF + wdst[it]) = wreg[it];
  };
  auto compute = [&](u32x4 (&F)[KSL][VS], int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sl = 0; sl < KSL; ++sl) {
      u32x4 af[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) af[i] = *(const u32x4*)(lds + buf * WBUF + (i * 16 + r16) * WROW + sl * 64 + g * 16);
#pragma unroll
      for (int j = 0; j < VS; ++j)
#pragma unroll
        for (int i = 0; i < NS; ++i) acc[i][j] = mma_chunk<T>(af[i], F[sl][j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- pipeline: stage s computes from (F_s, LDS buffer s & 1) while stage s + 1's loads fly
  issue(fa, 0, 0, true);
  wstore(0);
  __syncthreads();
  int t = 0, kg = 0;                                       // tap / channel group of the NEXT stage to issue
  auto advance = [&]() { if (++kg == r.nkg) { kg = 0; ++t; } };
  advance();
  for (int s = 0; s < nstage; s += 2) {
    issue(fb, t, kg, s + 1 < nstage);
    advance();
    compute(fa, 0);
    wstore(1);
    __syncthreads();
    if (s + 1 < nstage) {
      issue(fa, t, kg, s + 2 < nstage);
      advance();
      compute(fb, 1);
      wstore(0);
      __syncthreads();
    }
  }

  // ---- epilogue: D row 4g+r of tile i = channel crow(16i+4g+r), col = voxel r16 (as conv_igemm.hip)
  T* __restrict__ yg = (T*)a.y;
  constexpr int NH = NS / 2;
  float* part = a.partials ? a.partials + ((size_t)cls * r.ntile + tile) * a.Cout * 2 : nullptr;
#pragma unroll
  for (int j = 0; j < VS; ++j) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int co = co0 + h * 32 + g * 8;
      const bool wr = valid[j] && co < a.Cout;
      f32x4 o0 = acc[2 * h][j], o1 = acc[2 * h + 1][j];
      if (a.bias && co < a.Cout) { o0 += *(const f32x4*)(a.bias + co); o1 += *(const f32x4*)(a.bias + co + 4); }
      typedef __attribute__((ext_vector_type(4))) __bf16 bfx4;
      typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
      if (a.accumulate && wr) {                            // y += conv (the data gradient that lands on a tensor which already holds one)
        typedef __attribute__((ext_vector_type(8))) float f32x8_;
        const f32x8_ f = __builtin_convertvector(*(const bfx8*)(yg + (size_t)vout[j] * a.Cout + co), f32x8_);
        o0 += f32x4{f[0], f[1], f[2], f[3]}; o1 += f32x4{f[4], f[5], f[6], f[7]};
      }
      const bfx4 p0 = __builtin_convertvector(o0, bfx4), p1 = __builtin_convertvector(o1, bfx4);   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
      if (wr && !AM_DBG(a, 1)) *(bfx8*)(yg + (size_t)vout[j] * a.Cout + co) = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
      // what was stored, for the statistics below
      acc[2 * h][j] = wr ? __builtin_convertvector(p0, f32x4) : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[2 * h + 1][j] = wr ? __builtin_convertvector(p1, f32x4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (part) {                                              // per-workgroup per-channel (sum, sum of squares) of the stored values
    __syncthreads();                                       // the last stage's fragment reads are done: LDS is free
    float* red = (float*)lds;                              // [4 waves][NT][2]
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VS; ++j) { const float o = acc[i][j][q]; s1 += o; s2 += o * o; }
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        if (r16 == 0) {
          const int c = (i >> 1) * 32 + g * 8 + (i & 1) * 4 + q;
          red[(wave * NT + c) * 2] = s1; red[(wave * NT + c) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < NT && co0 + tid < a.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += red[(w * NT + tid) * 2]; s2 += red[(w * NT + tid) * 2 + 1]; }
      part[(co0 + tid) * 2] = s1; part[(co0 + tid) * 2 + 1] = s2;
    }
  }
}

struct GaGeo { int bs, M, ntile, ny, ns, ksl, S, OS, ncls, Ed, Eh, Ew, wide; };

// Which launches take this kernel (bf16, k3 / transposed k4, 128-channel source groups, 64-channel output tiles, no fused epilogue,
// tensors below 2 GB for the 32-bit row offsets):
//   * block-sparse, the same mask on both sides, output patches of at most 4^3 voxels: forward stride 1 / 2, data gradient stride 1 / 2
//     (the strided data gradient enumerates the COARSE voxels and computes the 8 output parities as 8 classes of workgroups),
//   * dense transposed convs (k4 s2) on grids that the bricks of conv_igemm pad by 1.3x or more (10^3, 12^3, 20^3: STUNet-L / H decoders),
//     dense k3 s1 convs from 1.5x (20-wide grids).
// (Do, Ho, Wo) is the launch's output grid.
bool ga_geometry(GaGeo& G, int mode, int dtype, int k, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, bool sparse, int out_bshift, int n_active) {
  if (dtype != AM_DT_BF16) return false;
  const bool fwd = mode == AM_CONV_FWD && k == 3 && (stride == 1 || stride == 2);
  const bool dg1 = mode == AM_CONV_DGRAD && k == 3 && stride == 1, dg2 = mode == AM_CONV_DGRAD && k == 3 && stride == 2;
  const bool ct = mode == AM_CONVT_FWD && k == 4 && stride == 2;
  const bool ctd = mode == AM_CONVT_DGRAD && k == 4 && stride == 2;     // = a k4 s2 conv of dy: every coarse voxel gathers a 4^3 window of the fine grid
  if (!fwd && !dg1 && !dg2 && !ct && !ctd) return false;
  if (Cin % 64 || Cout % 64) return false;
  if (Cin % 128 && !(fwd && stride == 2)) return false;   // 64-channel sources: the strided forward conv only (level 1 -> 2 of STUNet-B: 321 -> 211 us)
  G.S = (fwd || ctd) ? stride : 1;
  G.OS = (dg2 || ct) ? 2 : 1;
  G.ncls = G.OS == 2 ? 8 : 1;
  if (G.OS == 2 && ((Do | Ho | Wo) & 1)) return false;
  G.Ed = Do / G.OS; G.Eh = Ho / G.OS; G.Ew = Wo / G.OS;    // enumeration grid (= the source grid unless S == 2)
  const size_t src_vox = (size_t)B * G.Ed * G.Eh * G.Ew * G.S * G.S * G.S, out_vox = (size_t)B * Do * Ho * Wo;
  if (src_vox * Cin * 2 >= 0x7fffff00ull || out_vox * Cout * 2 >= 0x7fffff00ull || out_vox >= 0x7fffffffull) return false;
  int max_bs = 2;                                          // patches of 4^3, 2^3 voxels and single voxels (8^3 and 16^3 patches hold whole bricks)
  int wide = -1;                                           // 128-channel tiles: -1 = when they still give every CU a workgroup
#ifdef AM_ABLATE
  { const char* e_ = getenv("AM_GA_MAXBS"); if (e_) max_bs = atoi(e_); }      // tools: -1 disables the kernel
  { const char* e_ = getenv("AM_GA_WIDE"); if (e_) wide = atoi(e_); }
#endif
  if (max_bs < 0) return false;
  if (sparse) {
    if (ct || ctd || n_active <= 0) return false;
    G.bs = out_bshift - (G.OS == 2 ? 1 : 0);               // patch edge on the enumeration grid
    // (the strided data gradient: coarse patches of at most 2^3 -- with 4^3 its one-to-eight-tap class workgroups measured slower than the bricks)
    if (G.bs < 0 || G.bs > (dg2 && max_bs > 1 ? 1 : max_bs)) return false;
    G.M = n_active << (3 * G.bs);
  } else {
    // padding of conv_igemm's bricks on this q grid: the better of 4 x 8 x 8 and 4 x 4 x 16 (k3 s1 plans always take the 16-wide brick)
    auto up = [](int v, int m) { return (v + m - 1) / m * m; };
    const double vol = (double)G.Ed * G.Eh * G.Ew;
    const double p8 = up(G.Ed, 4) * (double)up(G.Eh, 8) * up(G.Ew, 8) / vol, p16 = up(G.Ed, 4) * (double)up(G.Eh, 4) * up(G.Ew, 16) / vol;
    double dense_k3 = 1.5;                                 // padding from which dense k3 s1 convs come here: 20-wide grids (1.6x: +6-10 %); at 1.33x (24-wide) the bricks' h-run reuse still wins on some shapes
#ifdef AM_ABLATE
    { const char* e_ = getenv("AM_GA_DENSEK3"); if (e_) dense_k3 = atof(e_); }
#endif
    if (ct || ctd) { if ((p8 < p16 ? p8 : p16) < 1.3 || G.Ed * G.S > 1023 || G.Eh * G.S > 1023 || G.Ew * G.S > 1023) return false; }
    else if (!((fwd && stride == 1) || dg1) || p16 < dense_k3) return false;
    G.bs = 0;
    G.M = B * G.Ed * G.Eh * G.Ew;
  }
  G.wide = ctd ? 1 : 0;
  G.ntile = (G.M + 127) / 128;
  // 128-channel tiles halve the gather traffic per MFMA (the kernel is bound by the caches' bandwidth: 1 KB gathered per 4 MFMAs with
  // 64-channel tiles) -- 1.4-1.5x on the 4^3- and 2^3-patch levels; with fewer than 256 of them (one-voxel patches: a few thousand
  // active voxels) the 64-channel tiles' second workgroup per voxel tile is worth more (profiles/r03_t_gather_ab.txt)
  const bool w8 = wide < 0 ? (long)G.ntile * G.ncls * (Cout / 128) >= 256 : wide != 0;
  G.ns = (w8 && Cout % 128 == 0 && Cin % 128 == 0) ? 8 : 4;
  G.ny = Cout / (16 * G.ns);
  G.ksl = G.ns == 8 ? 4 : (Cin % 256 == 0 ? 8 : Cin % 128 == 0 ? 4 : 2);     // (64-channel sources: stages of 2 slabs, 32 MFMAs per wave)
  return true;
}

template <int NS, int KSL>
int ga_launch(ConvArgs& a, GaArgs& r, hipStream_t st) {
  auto kern = conv_gather_kernel<NS, 2, KSL>;
  constexpr size_t lds = (size_t)2 * 16 * NS * (KSL * 64 + 16);
  AM_LDS_OPTIN_STAGE(kern);
  r.nkg = a.Cin / (32 * KSL);
  dim3 grid((unsigned)(((r.ntile + 7) / 8) * 8 * r.ny * r.ncls), 1, 1);
  AM_LAUNCH(kern, grid, dim3(256), lds, st, a, r);
  AM_CHECK_LAUNCH_STAGE();
  return 1;
}

}  // namespace

namespace amconv {

int conv_gather_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int out_sparse, int out_bshift,
                     int n_active) {
  GaGeo G;
  return ga_geometry(G, mode, dtype, ksize, stride, B, Do, Ho, Wo, Cin, Cout, out_sparse != 0, out_bshift, n_active) ? G.ntile * G.ncls : 0;
}

// returns 1 when it took the launch, 0 when the shape does not qualify, < 0 on error
int conv_gather_launch(int mode, int dtype, int ksize, int stride, ConvArgs& a0, const int* active_list, int n_active, void* stream) {
  const bool sparse = a0.out_mask.m != nullptr;
  if (sparse != (a0.in_mask.m != nullptr) || (sparse && (a0.in_mask.m != a0.out_mask.m || !active_list))) return 0;
  if (a0.ep_scale || a0.ep_res || a0.ep_act != AM_ACT_NONE || a0.nb_x) return 0;
  if (sparse && (a0.in_mask.fd > 255 || a0.in_mask.fh > 255 || a0.in_mask.fw > 255 || a0.B > 255)) return 0;
  GaGeo G;
  if (!ga_geometry(G, mode, dtype, ksize, stride, a0.B, a0.Do, a0.Ho, a0.Wo, a0.Cin, a0.Cout, sparse, a0.out_mask.bs, n_active)) return 0;
  if (a0.accumulate && a0.partials) return 0;
  // grids and block shifts must be those the geometry assumed: source grid = S x the enumeration grid, output grid = OS x it
  if (a0.Di != G.Ed * G.S || a0.Hi != G.Eh * G.S || a0.Wi != G.Ew * G.S) return 0;
  if (sparse && a0.in_mask.bs != G.bs + (G.S == 2 ? 1 : 0)) return 0;
  ConvArgs& a = a0;
  GaArgs r;
  r.plist = sparse ? active_list : nullptr; r.M = G.M; r.bs = G.bs; r.ntile = G.ntile; r.ny = G.ny; r.S = G.S; r.OS = G.OS; r.ncls = G.ncls;
  r.Ed = G.Ed; r.Eh = G.Eh; r.Ew = G.Ew; r.wide = G.wide;
  if (G.wide) {
    // the transposed conv's data gradient (P/decoder3D.py:16-17 ConvTranspose3d k4 s2 p1): dx[q] = sum_t W_t dy[2q + t - 1], t in 0..3 per dim
    for (int t = 0; t < 64; ++t) r.shift[t] = (t / 16) | (((t / 4) % 4) << 2) | ((t % 4) << 4) | (t << 8);      // (ud + 1 = td, ...)
    r.cbeg[0] = 0;
    for (int c = 1; c <= 8; ++c) r.cbeg[c] = 64;
  } else if (G.OS == 1) {
    // tap t = (td, th, tw) of the 3^3 kernel: the forward conv reads source voxel S * q + t - 1, the data gradient q + 1 - t (P/encoder3D.py
    // :12-15 is F.conv3d with padding 1; its gradient wrt the input correlates dy with the flipped kernel); weight slice = t in both packings
    for (int t = 0; t < 27; ++t) {
      const int td = t / 9, th = (t / 3) % 3, tw = t % 3;
      const int ud = mode == AM_CONV_FWD ? td - 1 : 1 - td, uh = mode == AM_CONV_FWD ? th - 1 : 1 - th, uw = mode == AM_CONV_FWD ? tw - 1 : 1 - tw;
      r.shift[t] = (ud + 1) | ((uh + 1) << 2) | ((uw + 1) << 4) | (t << 8);
    }
    r.cbeg[0] = 0;
    for (int c = 1; c <= 8; ++c) r.cbeg[c] = 27;
  } else {
    // two-class-stride plans: conv_plan.h lists, per output parity class, the taps and their source shifts on the coarse grid
    Plan P;
    P.a = a0;
    P.bd = 4; P.bh = 4; P.bw = 16; P.nt_tile = 64;        // (only the tap table of the plan is used)
    const int rc = build_plan(P, mode, ksize, stride);
    if (rc) return rc;
    if (P.a.OS != 2 || P.a.tap_begin[8] > 64) return 0;
    for (int c = 0; c <= 8; ++c) r.cbeg[c] = P.a.tap_begin[c];
    for (int c = 0; c < 8; ++c) if (r.cbeg[c + 1] == r.cbeg[c]) return 0;
    for (int t = 0; t < P.a.tap_begin[8]; ++t) {
      const int tp = P.a.taps[t];
      const int ud = (tp & 15) - 8, uh = ((tp >> 4) & 15) - 8, uw = ((tp >> 8) & 15) - 8;
      if (ud < -1 || ud > 1 || uh < -1 || uh > 1 || uw < -1 || uw > 1) return 0;
      r.shift[t] = (ud + 1) | ((uh + 1) << 2) | ((uw + 1) << 4) | (((tp >> 12) & 63) << 8);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (G.ns == 8) return ga_launch<8, 4>(a, r, st);
  return G.ksl == 8 ? ga_launch<4, 8>(a, r, st) : G.ksl == 4 ? ga_launch<4, 4>(a, r, st) : ga_launch<4, 2>(a, r, st);
}

}  // namespace amconv
