// Shared between the convolution kernels (conv_igemm.hip: generic gather implicit GEMM; conv_rw.hip: resident-weight variant
// for thin layers): launch arguments, tap plan, LDS layouts.
#pragma once
#include "common.h"

namespace amconv {

struct ConvArgs {
  const void* x; const void* w; const float* bias; void* y; float* partials;
  const float* ep_scale; const float* ep_shift; const void* ep_res; int ep_act;   // fused epilogue: act(conv * scale + shift + res)
  // norm-backward reduce fused into a data-gradient launch (bf16): y is the gradient wrt a = act(nb_x * nb_scale + nb_shift); the
  // partial rows then hold (sum g, sum g * nb_x) with g = y * act'(.) instead of (sum y, sum y^2)  -- am_conv3d_nbred
  const void* nb_x; const float* nb_scale; const float* nb_shift; int nb_act;
  // norm + activation of the INPUT fused into the source staging (conv_rw.hip only, am_conv3d_prenorm): the kernel reads x' = the norm's
  // input and convolves act(x' * in_scale + in_shift) on the rows that exist (inactive / out-of-volume rows stay zero) -- the
  // normalised map is never written.  NULL: x is convolved as it is
  const float* in_scale; const float* in_shift; int in_act;
  int B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, Cinp, Coutp;
  int OS, GS, nclass, nunit;  // output stride (parity classes), global source stride, #classes, #units
  int ny;                     // output-channel tiles per brick (conv_igemm: folded into blockIdx.x with the classes)
  int nbd, nbh, nbw;          // bricks per dim of the q grid
  int tap_begin[9];           // per unit.  A unit = one dense source sub-brick + the taps that read it:
                              //   OS == 2 : unit = output parity class (one per workgroup)
                              //   GS == 2 : unit = source parity sub-lattice (all units looped inside the workgroup)
  int upar[8];                // source parity of the unit (pd<<2 | ph<<1 | pw), 0 unless GS == 2
  int taps[64];               // (ud+8) | (uh+8)<<4 | (uw+8)<<8 | widx<<12 | unit<<18 ; u* = shift in sub-lattice voxels
  int mind[8], minh[8], minw[8];
  int ed[8], eh[8], ew[8];    // LDS source-brick extents per class
  int mdiv_w[8], mdiv_hw[8];  // 2^20-scaled reciprocals of ew and ew*eh (exact floor division for e < 1024)
  int w_bytes;                // size of the packed weight buffer
  int w_lds_off;              // byte offset of the weight-group buffers inside dynamic LDS
  MaskView in_mask, out_mask;
  int accumulate;
  int stats_sum_only;         // am_conv3d's accumulate bit 1: the caller reads only the SUM column of `partials` (conv_k3.hip has a cheaper epilogue for it)
  int brick_in_patch;         // block-sparse output and the q-brick lies inside one patch: one mask lookup decides the whole brick
  const int* plist;           // ... and the active-patch list is at hand: the grid enumerates the LIVE bricks only (patch = plist[i / bpp], brick i % bpp
  int pbd, pbh, pbw, nlive;   //   of its pbd x pbh x pbw bricks); nlive = n_active * bpp.  nullptr: every brick of the q grid, empty ones exit
  int hreuse;                 // taps ordered in h-runs of 3 (see build_plan): the HR kernel variant shares fragment rows across a run
  int nt_store;               // non-temporal output stores (outputs far larger than the 256 MB Infinity Cache)
#ifdef AM_ABLATE
  int dbg;                    // tools-only build (-DAM_ABLATE): AM_CV_DBG ablation bits, 1 no stores, 2 no source loads, 4 no weight loads, 64 a third of the weight-fragment LDS reads skipped, 512 no statistics epilogue, 1024 its barriers off, 2048 its DPP reductions off
#endif
};

// Ablation switches exist only in the tools build (anatomask_amd.build --ablate -> libanatomask_hip_ablate.so); in the product
// library AM_DBG() is the constant false and no environment variable is ever read.
#ifdef AM_ABLATE
#define AM_DBG(a_, bit_) (((a_).dbg & (bit_)) != 0)
#else
#define AM_DBG(a_, bit_) false
#endif

constexpr int ROWB = 64;      // channel-slab bytes staged per voxel / per weight row (unpadded, XOR-swizzled)
constexpr int LROWB = 80;     // LDS row stride of the source brick (16 B pad; B-fragment address = lane const + scalar tap offset)
// taps per weight group staged in LDS (TGS = 3: ~48 MFMAs per wave between barriers).  Multi-unit plans (strided / transposed:
// units of 1, 2, 4 or 8 taps) take TGS = 2: groups of 3 would pad 27 real taps to 39 issued ones (8 to 9 for ConvT), groups of 2 to 28.
constexpr int TG_OF(int ns, int tgs) { return ns == 2 ? 2 * tgs : tgs; }

// LDS image of a [rows][64 B] tile: 16-byte chunk c of row r lives at r*64 + ((c ^ 2*bit2(r)) * 16).
// With this swizzle a ds_read_b128 of 16 consecutive rows (any alignment) x 4 chunks is bank-conflict free
// (brute-forced over the four 16-lane groups of the instruction, MI355X_MICROARCH.md "LDS").
__device__ __forceinline__ int swz(int row, int chunk) { return (row << 6) + ((chunk ^ ((row >> 1) & 2)) << 4); }

// MFMA tile row -> output channel.  D row 4g+r of cout tile i lands in lane group g, register r; rows are assigned so that
// a lane's registers of tiles (2h, 2h+1) are the 8 CONSECUTIVE channels h*32 + g*8 .. +8: the epilogue stores 16 bytes per
// lane and the 4 lane groups of a voxel write one contiguous 64-byte run (instead of 8-byte pieces of four 32-byte runs).
__device__ __forceinline__ int crow(int R) { return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3); }


// host: tap tables ------------------------------------------------------------------------------
struct Plan { ConvArgs a; int bd, bh, bw; size_t lds; int nit, nt_tile, tgs; };

inline int build_plan(Plan& P, int mode, int k, int stride) {
  ConvArgs& a = P.a;
  const int pad = (mode == AM_CONVT_FWD || mode == AM_CONVT_DGRAD) ? 1 : k / 2;
  a.OS = 1; a.GS = 1; a.nclass = 1;
  if (mode == AM_CONV_FWD) a.GS = stride;
  else if (mode == AM_CONV_DGRAD) a.OS = stride;
  else if (mode == AM_CONVT_FWD) { a.OS = 2; if (k != 4 || stride != 2) return -2; }
  else if (mode == AM_CONVT_DGRAD) { a.GS = 2; if (k != 4 || stride != 2) return -2; }
  else return -2;
  if (a.OS == 2) a.nclass = 8;
  a.nunit = (a.OS == 2 || a.GS == 2) ? 8 : 1;
  if (k * k * k > 64) return -2;
  int n = 0;
  for (int c = 0; c < a.nunit; ++c) {
    a.tap_begin[c] = n;
    a.upar[c] = a.GS == 2 ? c : 0;
    const int p[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
    int mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    bool first = true;
    for (int td = 0; td < k; ++td) for (int th = 0; th < k; ++th) for (int tw = 0; tw < k; ++tw) {
      const int t[3] = {td, th, tw};
      int u[3]; bool ok = true;
      for (int d = 0; d < 3; ++d) {
        if (a.GS == 2) {                       // source voxel 2q + (t - pad) = 2(q + u) + r: belongs to unit c iff r == p[d]
          const int sft = t[d] - pad, r = ((sft % 2) + 2) % 2;
          if (r != p[d]) { ok = false; break; }
          u[d] = (sft - r) / 2;
        } else if (mode == AM_CONV_FWD) u[d] = t[d] - pad;
        else if (a.OS == 1) u[d] = pad - t[d];
        else { const int num = p[d] + pad - t[d]; if (num & 1) { ok = false; break; } u[d] = num / 2; }
      }
      if (!ok) continue;
      if (n >= 64) return -2;
      a.taps[n++] = (u[0] + 8) | ((u[1] + 8) << 4) | ((u[2] + 8) << 8) | ((td * k * k + th * k + tw) << 12) | (c << 18);
      for (int d = 0; d < 3; ++d) { if (first || u[d] < mn[d]) mn[d] = u[d]; if (first || u[d] > mx[d]) mx[d] = u[d]; }
      first = false;
    }
    a.mind[c] = mn[0]; a.minh[c] = mn[1]; a.minw[c] = mn[2];
    a.ed[c] = P.bd + (mx[0] - mn[0]);
    a.eh[c] = P.bh + (mx[1] - mn[1]);
    a.ew[c] = P.bw + (mx[2] - mn[2]);
    a.mdiv_w[c] = (1 << 20) / a.ew[c] + 1;
    a.mdiv_hw[c] = (1 << 20) / (a.ew[c] * a.eh[c]) + 1;
  }
  for (int c = a.nunit; c <= 8; ++c) a.tap_begin[c] = n;
  // k3 s1 plans (one unit, 27 taps): order the taps as (ud, uw) runs of uh = min, min+1, min+2.  With the 4x4x16 brick the wave's
  // four voxel subtiles are four consecutive h-rows, so the three taps of a run read 6 distinct fragment rows instead of 12.
  a.hreuse = 0;
  if (a.nunit == 1 && k == 3 && n == 27 && P.bh == 4 && P.bw == 16) {      // (the kernels check the d-extent they need)
    int srt[27], m = 0;
    for (int ud = -1; ud <= 1; ++ud) for (int uw = -1; uw <= 1; ++uw) for (int uh = -1; uh <= 1; ++uh)
      for (int t = 0; t < 27; ++t)
        if ((a.taps[t] & 15) - 8 == ud && ((a.taps[t] >> 4) & 15) - 8 == uh && ((a.taps[t] >> 8) & 15) - 8 == uw) srt[m++] = a.taps[t];
    if (m == 27) { for (int t = 0; t < 27; ++t) a.taps[t] = srt[t]; a.hreuse = 1; }
  }
  size_t mxv = 0;
  for (int c = 0; c < a.nunit; ++c) { size_t v = (size_t)a.ed[c] * a.eh[c] * a.ew[c]; if (v > mxv) mxv = v; }
  P.nit = (int)((mxv * (ROWB / 16) + 255) / 256);
  size_t brick = mxv * LROWB;
  if (brick < 4096) brick = 4096;                        // the stats epilogue reuses the head of the brick
  a.w_lds_off = (int)brick;
  P.tgs = a.nunit > 1 ? 2 : 3;
  P.lds = a.w_lds_off + 2 * TG_OF(P.nt_tile / 16, P.tgs) * P.nt_tile * ROWB;
  return 0;
}


// resident-weight kernel (conv_rw.hip): returns 1 when it took the launch, 0 when the shape does not qualify, < 0 on error
int conv_rw_launch(int mode, int dtype, int ksize, int stride, ConvArgs& a, const int* active_list, int n_active, void* stream);
int conv_rw_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int out_sparse, int out_bshift,
                 int n_active);   // partial-sum rows such a launch writes, or 0 when the shape does not qualify

// voxel-list gather kernel (conv_gather.hip): k3 s1 forward / data gradient of block-sparse tensors whose patches are at most 2^3 voxels
int conv_gather_launch(int mode, int dtype, int ksize, int stride, ConvArgs& a, const int* active_list, int n_active, void* stream);
int conv_gather_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int out_sparse, int out_bshift,
                     int n_active);   // partial-sum rows such a launch writes, or 0 when the shape does not qualify

// dense k3 s1 forward / data gradient, bf16, persistent 8-wave LDS-DMA kernel (conv_k3.hip): 1 when it took the launch, 0 when the shape
// does not qualify; conv_k3_rows = the partial-sum rows such a launch writes (8 per workgroup), or 0
int conv_k3_launch(int mode, int dtype, int ksize, int stride, ConvArgs& a, void* stream);
int conv_k3_rows(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cin, int Cout, int masks);
int conv_k3t_launch(int mode, int dtype, int ksize, int stride, ConvArgs& c, void* stream);   // ConvTranspose3d k4 s2 p1 on the same kernel

}  // namespace amconv
