/* anatomask_hip.h -- C ABI of libanatomask_hip.so (gfx950 / MI355X).
 *
 * The reference (ricklisz/AnatoMask) is 100 % Python on torch/cuDNN: it has NO FFI or plugin layer
 * (SURVEY.md 8b).  The drop-in boundary for its hot path is therefore this library: plain pointers,
 * sizes and a hipStream_t (passed as void*), no torch types.  Each entry point names the reference
 * code it replaces (paths relative to nnunetv2/training/nnUNetTrainer/variants/pretrain/ = P/).
 *
 * Conventions
 *  - activations: channels-last [B][D][H][W][C], dtype AM_DT_F32 (float) or AM_DT_BF16 (uint16 bits),
 *    C % 8 == 0, 16-byte aligned base pointers.  C == 1 tensors (input volume, reconstruction) are fp32.
 *  - patch mask: uint8 [B][fd][fh][fw] (1 = visible/active); a tensor whose voxels are (1<<bshift)
 *    per patch edge is "block-sparse": kernels read inactive voxels as 0 and never rely on what is
 *    stored there.  mask == NULL means dense.
 *  - every function is asynchronous on `stream`, never allocates, never synchronises, returns 0 or
 *    a negative argument error / positive hipError_t; -(1000 + hipError_t) = a launch that failed inside one of
 *    the convolution's kernel-selection stages (never reported as success, never a silent fall-through).
 *  - packed conv weights: [tap][rows_p][K_p] in the compute dtype, rows = channels of the tensor WRITTEN,
 *    K = channels of the tensor READ, tap = (td*k + th)*k + tw of the original kernel; rows/K are zero-padded
 *    to whole MFMA tiles (am_packed_dims) so the inner loop carries no bounds logic.
 */
#ifndef ANATOMASK_HIP_H
#define ANATOMASK_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AM_DT_F32 0
#define AM_DT_BF16 1
/* fp32 tensors, channel contractions on the bf16 matrix cores from hi / lo splits of both operands (hi hi + hi lo + lo hi + lo lo, fp32
 * accumulation): accepted by am_conv3d, am_pack_weight(s), am_packed_dims, am_conv3d_partials_rows wherever AM_DT_F32 is; every other entry
 * point takes AM_DT_F32 for such tensors.  4x the rate of the exact-fp32 mode; 16 significant bits per operand. */
#define AM_DT_F32S 2

/* am_conv3d `accumulate`: bit 0 = add into y (the k1 s2 shortcut's data gradient); bit 1 = with `partials`, the caller reads only their SUM
 * column (a bias gradient = per-channel sum of the output): a kernel may then leave the sum-of-squares column zero. */
#define AM_CONV_PARTIALS_SUM_ONLY 2
#define AM_CONV_FWD 0     /* y[o]  = sum_t x[o*stride + t - k/2] W_t        (k = 1|3, stride 1|2)            */
#define AM_CONV_DGRAD 1   /* dx[i] = sum_t dy[(i + k/2 - t)/stride] W_t^T   (data gradient of AM_CONV_FWD)   */
#define AM_CONVT_FWD 2    /* y[o]  = sum_t x[(o + 1 - t)/2] W_t             (ConvTranspose3d k4 s2 p1)       */
#define AM_CONVT_DGRAD 3  /* dx[i] = sum_t dy[2i - 1 + t] W_t^T                                              */

#define AM_DXREP 64       /* replicas of the bias-gradient accumulator of am_norm_bwd_apply (2048 workgroups on one address serialise) */
#define AM_NREP 8        /* replicated reduction accumulators (spread atomic contention), summed by the finalize kernels */

#define AM_ACT_NONE 0
#define AM_ACT_LRELU 1    /* LeakyReLU(0.01)  P/STUNet_head.py:84,89 */
#define AM_ACT_RELU6 2    /* ReLU6            P/decoder3D.py:21      */

int am_version(void);

/* Convolutions on the matrix cores.
 * Replaces: SparseConv3d / sp_conv_forward P/encoder3D.py:12-15,27-28 (in_mask/out_mask set),
 * densify_proj Conv3d P/AnatoMask.py:63-65, UNetBlock convs + ConvTranspose3d P/decoder3D.py:19-22,
 * and their autograd data gradients (loss.backward() P/pretrain_AntoMask.py:435). */
int am_conv3d(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed, const float* bias, void* y,
              int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
              const uint8_t* in_mask, int in_bshift, const uint8_t* out_mask, int out_bshift, int fd, int fh, int fw,
              int accumulate, float* partials /* NULL or [am_conv3d_partials_rows][Cout][2]: per-workgroup sum / sum-of-squares
              of the written outputs, feeds the following norm (or a bias gradient) without another pass */,
              const float* ep_scale, const float* ep_shift /* NULL or [Cout]: y = conv * scale + shift -- an eval-mode BatchNorm
              (running statistics, the EMA teacher's decoder: P/decoder3D.py:20-22 under model_ema.ema.eval()) folded into the store */,
              const void* ep_res /* NULL or a tensor shaped like y that is added (x = x + to_dec[i], P/decoder3D.py:59) */,
              int ep_act /* AM_ACT_*: applied last */,
              const int32_t* active_list, int n_active /* am_mask_compact of out_mask, or NULL / 0: block-sparse launches whose bricks lie inside
              the patches then enumerate their live bricks instead of launching (and leaving) the empty ones; it also lets thin block-sparse layers
              (Cin <= 32) run on persistent workgroups that walk only the active bricks with all weights resident in LDS, and the
              levels whose patches are at most 4^3 voxels on the voxel-list gather kernel (rows = active voxels; conv_gather.hip) */,
              int* partial_rows_written /* NULL or (host) the number of partials rows this launch wrote (<= am_conv3d_partials_rows) */,
              void* stream);
/* Forward convolution of act(x * in_scale + in_shift): the pooled sparse InstanceNorm + LeakyReLU in front of a SparseConv3d
 * (P/STUNet_head.py:96-103: y = LReLU(IN(conv1 x)); conv2(y), P/encoder3D.py:12-15,138-165) folded into the consumer's source staging --
 * the normalised map is never written.  Inactive / out-of-volume source rows read as zero (dense-conv-then-mask semantics).  For passes
 * that keep no tape (the EMA teacher, validation, stand-alone encoder forwards).  One patch mask for input and output (in_bshift =
 * out_bshift + 1 for stride 2).  am_conv3d_prenorm_supported -> 1 / 0 (an answer, not an error code): bf16, Cin <= 32, k3, block-sparse
 * with an active-patch list (the resident-weight kernel of conv_rw.hip). */
int am_conv3d_prenorm_supported(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout, int sparse,
                                int in_bshift, int out_bshift, int n_active);
int am_conv3d_prenorm(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed, const float* bias, void* y, int B, int Di,
                      int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, const uint8_t* mask, int in_bshift, int out_bshift, int fd,
                      int fh, int fw, float* partials /* may be NULL */, const float* in_scale, const float* in_shift, int in_act,
                      const int32_t* active_list, int n_active, int* partial_rows_written /* (host) out, may be NULL */, void* stream);

/* am_conv3d whose output y is the gradient wrt a = act(nb_x * nb_scale + nb_shift) -- the output of a norm + activation
 * (P/decoder3D.py:20-22 BatchNorm3d + ReLU6, P/STUNet_head.py:96-103 InstanceNorm3d + LeakyReLU): the launch also leaves, per
 * workgroup and channel, (sum g, sum g * nb_x) with g = y * act'(.) in `partials` [rows][Cout][2] -- the sums autograd's norm
 * backward needs, without a pass of its own over y and nb_x.  bf16 only; nb_x has y's shape and layout; nb_act AM_ACT_NONE needs
 * no nb_scale / nb_shift.  Follow with am_norm_bwd_from_partials. */
int am_conv3d_nbred(int mode, int dtype, int ksize, int stride, const void* x, const void* w_packed, void* y, int B, int Di,
                    int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, const uint8_t* in_mask, int in_bshift,
                    const uint8_t* out_mask, int out_bshift, int fd, int fh, int fw, int accumulate, float* partials,
                    const void* nb_x, const float* nb_scale, const float* nb_shift, int nb_act,
                    int* partial_rows_written, void* stream);
int am_packed_dims(int dtype, int rows, int k, int* rows_padded, int* k_padded);
int am_conv3d_partials_rows(int mode, int dtype, int ksize, int stride, int B, int Do, int Ho, int Wo, int Cin, int Cout,
                            int out_sparse, int out_bshift /* the launch's out_mask != NULL and its block shift: they select the brick */,
                            int n_active, int* rows /* upper bound: size the partials buffer with it */);
/* partials [rows][C][2] -> sums[C][2] (double, overwritten; may be NULL) and/or sum_accum[C] += sum (may be NULL) */
int am_partials_reduce(const float* partials, int rows, int C, double* sums, float* sum_accum, void* stream);

/* Weight gradient (autograd of the above).  mode = AM_CONV_FWD or AM_CONVT_FWD.  x = forward input
 * [B][Dx][Hx][Wx][Cx], dy = gradient of the forward output [B][Dy][Hy][Wy][Cy];
 * dw_packed fp32 [tap][Cy][Cx] is ACCUMULATED into (zero it first).
 * The partial sums of the workgroups are reduced with fp32 atomics (bits depend on the arrival order, as torch's own
 * non-deterministic cuDNN/MIOpen weight gradients); with det_workspace != NULL every workgroup slot stores its partial sum to
 * det_workspace[slot][tap][Cy][Cx] instead and a second kernel adds the slots in slot order: bit-identical from run to run
 * (the counterpart of torch.use_deterministic_algorithms for this path).  det_workspace_floats >= k^3*Cy*Cx, more slots =
 * more parallelism (the launch uses min(its own slot count, what fits)). */
int am_conv3d_wgrad(int mode, int dtype, int ksize, int stride, const void* x, const void* dy, float* dw_packed,
                    int B, int Dx, int Hx, int Wx, int Cx, int Dy, int Hy, int Wy, int Cy,
                    const uint8_t* x_mask, int x_bshift, const uint8_t* y_mask, int y_bshift, int fd, int fh, int fw,
                    float* det_workspace, long det_workspace_floats,
                    const int32_t* active_list, int n_active /* NULL / 0, or am_mask_compact's list of the patch mask (both masks are views of ONE
                    patch mask): block-sparse dY whose bricks lie inside one patch is then walked over its LIVE bricks only, the same number
                    per workgroup slot (a contiguous run of all bricks gives the slots 40 % +- 45 % live ones, and the slowest slot is the launch) */,
                    void* gather_workspace, long gather_workspace_bytes /* NULL / 0, or scratch for the GATHER form (bf16, k3, stride 1 / 2, dY patches 1 or
                    2 voxels wide, an active list, Cy % 128 == 0, Cx % 64 == 0): the active voxels are gathered into K-major operands YT[Cy][Np],
                    XT[tap][Cx][Np] (Np = active dY voxels rounded up to 32) and the gradient is 27 plain GEMMs; holds YT + at least one tap:
                    2 * Np * (Cy + taps_per_round * Cx) bytes, fewer taps per round = more rounds.  am_conv3d_wgrad_gather_bytes sizes it (0: the
                    gather form does not apply to this launch) */,
                    void* stream);
/* 1 / 0 (an answer, not an error code): am_conv3d_wgrad serves this launch on the 8-wave LDS-DMA kernel of conv_wgk3.hip (dense bf16 k3 s1,
 * H % 8 == W % 16 == 0, Cx % 64 == 0, Cy % 32 == 0 (64- or 32-wide cy tiles), >= 512 bricks, atomic accumulation) rather than on conv_wgrad.hip's brick walk.  Same x / dy
 * geometry arguments as am_conv3d_wgrad (D, H, W of dY = of X). */
int am_conv3d_wgrad_uses_k3(int mode, int dtype, int ksize, int stride, int B, int D, int H, int W, int Cx, int Cy, int has_masks,
                            int deterministic);
int am_conv3d_wgrad_gather_bytes(int mode, int dtype, int ksize, int stride, int B, int Cx, int Cy, int y_bshift, int has_y_mask, int n_active,
                                 int fd, int fh, int fw, long* bytes /* (host) out */);

/* x (n floats, n % 4 == 0) -> hi = bf16(x), lo = bf16(x - hi) as two bf16 tensors of the same shape (AM_DT_F32S weight gradients: three
 * bf16 am_conv3d_wgrad launches on the planes -- hi hi, lo hi, hi lo -- accumulate into ONE fp32 gradient).  The reference computes these
 * contractions in fp32 (AMP = False, P/pretrain_AntoMask.py:239); the split keeps 16 significant bits per operand. */
int am_split_bf16(const float* x, void* hi, void* lo, long n, void* stream);

/* dst[t][r][k] = src[r*stride_r + k*stride_k + t] (zero in the padding): torch-layout fp32 master -> packed [taps][Rp][Kp].
 * dtype AM_DT_F32S: rows of Kp fp32-sized elements whose every 16-channel group is [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] in bf16. */
int am_pack_weight(int dtype, const float* src, void* dst, int R, int K, int taps, long stride_r, long stride_k, int Rp, int Kp,
                   void* stream);
/* All weight repacks of a step in ONE launch (the masters are views of one flat buffer and the packed copies persist, so the
 * descriptor table is built once): descs[i] repacks like am_pack_weight; first_block is the running sum of Rp * ceil(Kp/64). */
typedef struct am_pack_desc {
  const float* src; void* dst; long stride_r, stride_k; int R, K, taps, Rp, Kp, first_block;
} am_pack_desc;                                       /* 56 bytes, device memory */
int am_pack_weights_batched(int dtype, const am_pack_desc* descs, int ndesc, int total_blocks, void* stream);
/* dst[r*stride_r + k*stride_k + t] (+)= src[t][r][k]: packed fp32 gradient -> torch layout. */
int am_unpack_grad(const float* src_packed, float* dst, int R, int K, int taps, long stride_r, long stride_k, int accumulate,
                   void* stream);

/* Cin = 1 stem convolutions (STUNet stage 0 conv1 k3 / conv3 k1 on the masked input volume,
 * P/STUNet_head.py:81,92 under P/encoder3D.py:12-15) and their weight/bias gradients. */
int am_stem_conv_fwd(int dtype, const float* x, int B, int D, int H, int W, int C, int ksize, const uint8_t* mask, int bshift,
                     int fd, int fh, int fw, const float* w, const float* bias, void* y,
                     float* partials /* NULL or [B*(D/4)*(H/8)*(W/16)][C][2]: per-workgroup sum / sum of squares of y, as am_conv3d */,
                     const int32_t* active_list, int n_active /* am_mask_compact: bf16 k3 stems then run on the matrix cores, one
                     workgroup per active patch (K = 27 taps padded to one 16x16x32 MFMA); NULL / 0: the VALU kernel */,
                     int* partial_rows_written /* NULL or (host) the partials rows written (<= the bound above) */, void* stream);
int am_stem_conv_wgrad(int dtype, const float* x, const void* dy, int B, int D, int H, int W, int C, int ksize,
                       const uint8_t* mask, int bshift, int fd, int fh, int fw, float* dw_accum, float* db_accum,
                       const int32_t* active_list, int n_active /* am_mask_compact: bf16 stems then contract voxels on the matrix
                       cores (persistent workgroups over the active patches); NULL / 0: the VALU kernel */,
                       float* det_workspace, long det_workspace_floats /* NULL / 0, or (matrix-core path only) rows of C * (k^3 + 1) floats:
                       one per workgroup, folded in workgroup order instead of fp32 atomics (deterministic mode, as am_conv3d_wgrad) */,
                       void* stream);

/* Pooled sparse InstanceNorm (P/encoder3D.py:138-165: statistics over ALL active voxels of the local
 * batch) and BatchNorm3d (P/decoder3D.py:21-22) share these: stats -> finalize -> apply. */
int am_chan_stats(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                  int fw, double* sums /* [AM_NREP][C][2] replicated accumulators, zeroed inside */,
                  const int32_t* active_list, int n_active /* see am_mask_compact; NULL / 0: linear walk with per-voxel mask lookups */,
                  void* stream);
int am_mask_count(const uint8_t* mask, int n, int voxels_per_patch, double* out, void* stream);
/* Active-patch list of a patch mask: list[i] = b << 24 | pd << 16 | ph << 8 | pw of the i-th active patch (memory order),
 * count[0] = their number.  The reference rebuilds `nonzero` index tuples of the UP-SAMPLED mask in every sparse layer call
 * (P/encoder3D.py:7-10, 30x per forward); here the list is built once per mask and shared by every streaming kernel of the
 * forward and the backward, which then walk only the rows of active patches (no per-voxel mask lookup). */
int am_mask_compact(const uint8_t* mask, int B, int fd, int fh, int fw, int32_t* list, int32_t* count, void* stream);
int am_norm_finalize(const double* sums, int nrep /* AM_NREP after am_chan_stats, 1 after am_partials_reduce */, const double* count_ptr, double count_host, int C, const float* gamma,
                     const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                     float* run_mean /* NULL or BN running stats, updated */, float* run_var, float momentum, void* stream);
/* The statistics plumbing of one norm in ONE launch: conv partials [rows][C][2] -> per-channel sums -> (last workgroup) mean /
 * rstd / folded scale+shift, BatchNorm running-stat update and num_batches_tracked += 1 (nn.BatchNorm3d in train mode,
 * P/decoder3D.py:21-22).  gamma == NULL: only sum_accum[c] += sum (a bias gradient).  workspace: AM_FIN_REP*2*C + 1 doubles
 * (replicated accumulators + a ticket) that are ZERO on entry and are left ZERO on exit (allocate once, zero once, share between
 * consecutive calls on one stream). */
#define AM_FIN_REP 16
int am_partials_finalize(const float* partials, int rows, int C, double* workspace, const double* count_ptr, double count_host,
                         const float* gamma, const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                         float* run_mean, float* run_var, float momentum, long* num_batches_tracked, float* sum_accum, void* stream);
/* The rows of am_conv3d_nbred -> the coefficients of am_norm_bwd_apply (k0 = gamma*rstd, k1 = k0 * mean(g), k2 = k0 * mean(g*xhat))
 * and the affine gradients (dgamma += sum g*xhat, dbeta += sum g, dbeta2 likewise or NULL): what am_norm_bwd_reduce +
 * am_norm_bwd_finalize produce.  workspace: as am_partials_finalize (zero on entry, left zero). */
int am_norm_bwd_from_partials(const float* partials, int rows, int C, double* workspace, const double* count_ptr, double count_host,
                              const float* gamma, const float* mean, const float* rstd, float* k0, float* k1, float* k2,
                              float* dgamma, float* dbeta, float* dbeta2, void* stream);
int am_norm_fold_running(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var, float eps,
                         float* scale, float* shift, void* stream);   /* eval-mode BN (teacher) */
/* y = act(x*scale + shift [+ res] [+ stem_w*stem_x + stem_b]); fill != NULL: inactive voxels := mask token
 * (densify, P/AnatoMask.py:158-163).  Fuses norm + LeakyReLU/ReLU6 + residual add (P/STUNet_head.py:96-103). */
int am_norm_apply(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                  int fw, const float* scale, const float* shift, int act, const void* res, const float* stem_x,
                  const float* stem_w, const float* stem_b, const float* fill, void* y,
                  const int32_t* active_list, int n_active, void* stream);
/* backward: bsum[c] = {sum dpre, sum dpre*xhat, sum_{inactive} dout}, dpre = dout*act'(out) */
int am_norm_bwd_reduce(int dtype, const void* dout, const void* out, const void* x, int B, int D, int H, int W, int C,
                       const uint8_t* mask, int bshift, int fd, int fh, int fw, const float* mean, const float* rstd, int act,
                       int fill, double* bsum /* [AM_NREP][C][3], zeroed inside */,
                       const float* pre_scale, const float* pre_shift /* out == NULL with an activation (no residual): the
                       derivative is taken from the recomputed pre-activation x*pre_scale + pre_shift -- one read less */,
                       const int32_t* active_list, int n_active,
                       const double* count_ptr, double count_host, const float* gamma, float* k0, float* k1, float* k2,
                       float* dgamma_accum, float* dbeta_accum, float* dtoken_accum, float* dbeta2_accum
                       /* k0 != NULL: am_norm_bwd_finalize is FUSED into this launch (its last workgroup does it); bsum is then a
                          workspace of AM_NREP*C*3 doubles + 1 that is ZERO on entry and left ZERO (no memset, no finalize launch) */,
                       void* stream);
int am_norm_bwd_finalize(const double* bsum, const double* count_ptr, double count_host, int C, const float* gamma,
                         const float* rstd, float* k0, float* k1, float* k2, float* dgamma_accum, float* dbeta_accum,
                         float* dtoken_accum, float* dbeta2_accum /* bias of a conv added after the norm: same sum */, void* stream);
int am_norm_bwd_apply(int dtype, const void* dout, const void* out, const void* x, int B, int D, int H, int W, int C,
                      const uint8_t* mask, int bshift, int fd, int fh, int fw, const float* mean, const float* rstd,
                      const float* k0, const float* k1, const float* k2, int act, void* dx, void* dres,
                      float* dxsum_accum /* NULL or += per-channel sum of dx: bias gradient of the conv feeding the norm */,
                      double* dxsum_scratch /* [AM_DXREP][C] fp64 workspace, required when dxsum_accum != NULL: the per-workgroup sums are added in
                      double so that their arrival order shows at 1e-16, not in the fp32 result */,
                      const float* pre_scale, const float* pre_shift /* as in am_norm_bwd_reduce */,
                      const int32_t* active_list, int n_active,
                      int scratch_is_zero_workspace /* 1: dxsum_scratch = AM_DXREP*C doubles + 1, zero on entry / left zero; the fold into
                      dxsum_accum happens in the last workgroup of this launch */, void* stream);
int am_chan_sum(int dtype, const void* x, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                int fw, float* out_accum, void* stream);   /* conv bias gradients */
int am_add(int dtype, const void* a, const void* b, void* y, long n_elems, void* stream);   /* x + to_dec[i], P/decoder3D.py:59 */

/* The decoder's tail in EVAL mode as ONE C -> 1 stencil (the EMA teacher's pass, P/pretrain_AntoMask.py:421-425, and the validation pass
 * of P/pretrain.py:426-441): the last UNetBlock ends conv3x3x3(C -> Cmid, no bias) -> BatchNorm3d(running statistics) -> Conv3d(Cmid -> 1, k1)
 * (P/decoder3D.py:20-22,51,61), which without batch statistics is linear in the block's ReLU6 output r:
 *   rec[q] = b_eff + sum_t sum_ci W_eff[t][ci] r[q + t][ci],  W_eff[t][ci] = sum_c w_proj[c] scale[c] W2[c][ci][t],  b_eff = b_proj + sum_c w_proj[c] shift[c]
 * am_head_fold: w2 = the conv's torch-layout weight (Cmid, Cin, 3, 3, 3), scale / shift = the folded BatchNorm (am_norm_fold_running)
 * -> weff fp32 [27][Cin], beff fp32 [1].  am_head_stencil: r channels-last [B][D][H][W][C] (bf16: W_eff enters as hi + lo bf16 parts, fp32
 * accumulation; fp32 storage, either product mode: exact fp32 products), evaluated on the 16^3 patches of patch_list (am_mask_compact
 * entries) only; rec fp32 [B][D][H][W] (NULL: not written; other patches untouched); l2 (NULL or fp32 [B][fd*fh*fw]): the raw per-patch
 * mean((rec - inp)^2) of the listed patches (inp fp32 [B][D][H][W]); entries of other patches untouched.
 * am_head_stencil_supported(dtype, C) -> 1 / 0 (the one entry point whose return value is an answer, not an error code). */
int am_head_fold(const float* w2, int cmid, int cin, const float* scale, const float* shift, const float* wproj, const float* bproj,
                 float* weff, float* beff, void* stream);
int am_head_stencil(int dtype, const void* r, int B, int D, int H, int W, int C, const float* weff, const float* beff,
                    const int32_t* patch_list, int n_patches, float* rec /* may be NULL */, const float* inp /* NULL unless l2 */,
                    float* l2 /* may be NULL */, void* stream);
int am_head_stencil_supported(int dtype, int C);

/* 1x1 projection C -> 1 (P/decoder3D.py:51,61) and its backward. rec/drec are fp32 [B][D][H][W]. */
int am_proj_fwd(int dtype, const void* x, long nvox, int C, const float* w, const float* b,
                const float* pre_scale, const float* pre_shift /* NULL, or x is the INPUT of a per-channel affine map (the train-mode
                BatchNorm of P/decoder3D.py:22 whose output feeds only this projection): rec = proj(x * pre_scale + pre_shift) without
                that tensor ever being written */, float* rec, void* stream);
/* Backward of the projection head TOGETHER with the train-mode BatchNorm in front of it (P/decoder3D.py:22 -> :51,61; autograd of
 * both), for the last decoder block (no activation, no skip add): x = the BatchNorm's input, drec = d loss / d rec.
 * The gradient wrt the BatchNorm output is the rank-1 tensor drec[v] * w[c]; neither it nor the BatchNorm output is materialised.
 * One reduce pass (S0 = sum drec, S1[c] = sum drec * (x - mean)) gives proj.weight / proj.bias / BN weight / BN bias gradients
 * (all accumulated) and the coefficients of the second pass, which writes dx = d loss / d x.
 * workspace: AM_NREP * (C + 1) doubles + 1, ZERO on entry and left zero; coef: 3 * C floats of scratch. */
int am_proj_norm_bwd(int dtype, const void* x, const float* drec, long nvox, int C, const float* w, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, const float* scale, double* workspace, float* coef, void* dx,
                     float* dgamma_accum, float* dbeta_accum, float* dw_accum, float* db_accum, void* stream);
int am_proj_bwd(int dtype, const void* x, const float* drec, long nvox, int C, const float* w, void* dx, float* dw_accum,
                float* db_accum, float* det_workspace, long det_workspace_floats /* NULL / 0, or >= 1024 * (C + 1) floats: per-workgroup rows folded in order
                instead of fp32 atomics (deterministic mode) */, void* stream);

/* Per-patch reconstruction loss without materialising patchify (P/AnatoMask.py:190-202,221-228;
 * teacher variant normalized=0: P/pretrain_AntoMask.py:423-425).  l2m = per-patch MSE * non_active (B*L);
 * lossinfo[0] = loss, lossinfo[1] = 1/(count+1e-8). */
int am_patch_loss_fwd(const float* inp, const float* rec, const uint8_t* active, int B, int D, int H, int W, int normalized,
                      float* l2m, float* pmean, float* prstd, float* lossinfo, void* stream);
int am_patch_loss_bwd(const float* inp, const float* rec, const uint8_t* active, int B, int D, int H, int W, const float* pmean,
                      const float* prstd, const float* lossinfo, const float* gout, float* drec, void* stream);

/* Reconstruction-guided hard-mask sampler, SparK.generate_mask P/AnatoMask.py:81-128 (mask output),
 * key-driven: the len_loss highest-loss patches are never visible, of the rest the len_keep smallest
 * keys are.  No host round trip (the reference does .cpu().numpy() per sample, :112-114). */
int am_mask_sampler(const float* loss, const float* keys, int B, int L, int len_keep, int len_loss, uint8_t* mask, void* stream);

/* clip_grad_norm_(max_norm) + AdamW + ModelEma.update fused over a flat fp32 parameter buffer
 * (P/pretrain_AntoMask.py:437-440; torch.optim.AdamW; timm.utils.ModelEma). n % 4 == 0. */
int am_sumsq(const float* g, long n, double* out, void* stream);
int am_adamw_ema(float* p, const float* g, float* m, float* v, float* ema /* may be NULL */, long n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, const double* sumsq /* NULL: no clipping */,
                 double max_norm, double ema_decay, double grad_scale /* g is multiplied by it before the norm and the update: 1/world
                 when g holds the all-reduced SUM (DDP's gradient mean, P/pretrain_AnatoMask_DDP.py:239-240), else 1 */,
                 float* gnorm_out,
                 const float* dyn_scalars /* NULL, or DEVICE float[4] = {lr, 1 - beta1^step, sqrt(1 - beta2^step), ema_decay} that override
                 the by-value arguments: a step captured in a hipGraph replays with per-step values the host writes before each launch */,
                 int* guard /* NULL, or DEVICE int[4] = {latched, first bad call (1-based), calls, 0}: the per-step non-finite stop of
                 P/pretrain_AntoMask.py:441-446 without a host round trip.  A call whose gradient norm is not finite (am_patch_loss_bwd makes the
                 gradient of a non-finite loss NaN, so the all-reduced norm is non-finite on EVERY rank) writes nothing to p / m / v / ema
                 and latches guard[0]; every later call is skipped as well, so the state the driver finds at its one per-epoch look is the
                 state before the bad step */,
                 void* stream);
int am_ema(float* ema, const float* p, long n, double decay, const int* guard /* NULL, or am_adamw_ema's: skipped when latched */, void* stream);
/* ModelEma.update on the integer entries (BatchNorm num_batches_tracked): ema = int64(float32(ema) * decay + (1 - decay) * float32(p)),
 * the promotion / truncation `ema_v.copy_(ema_v * decay + (1. - decay) * model_v)` performs on an int64 tensor (timm.utils.ModelEma). */
int am_ema_i64(long* ema, const long* p, int n, double decay, const int* guard, void* stream);
/* guard latched: dst = snapshot (the BatchNorm running statistics / counters the student's forward of the bad step updated). nbytes % 4 == 0. */
int am_guard_restore(void* dst, const void* snapshot, long nbytes, const int* guard, void* stream);

/* Device-side spatial augmentation of the data feed: batchgenerators' SpatialTransform (rotation, isotropic scale,
 * order-3 spline interpolation, constant border) + MirrorTransform as the reference configures them (P/pretrain_AntoMask.py:78-113),
 * applied on the GPU to the enlarged patch the loader crops.  vol / src / dst: fp32 [D][H][W] device buffers of ONE sample.
 * am_spline_prefilter: in place, == scipy.ndimage.spline_filter(order=3, mode='mirror').  am_resample_affine: dst[o] =
 * interp(src, A (o,1)) with A a HOST array of 12 floats (3x4, row-major: output index -> source coordinate); order 0 (integer crop /
 * flip), 1 (trilinear) or 3 (cubic B-spline over prefiltered coefficients, scipy map_coordinates(mode='constant') rules). */
int am_spline_prefilter(float* vol, int D, int H, int W, void* stream);
int am_resample_affine(const float* src, int Ds, int Hs, int Ws, float* dst, int D, int H, int W, const float* affine_host12, int order,
                       float cval, void* stream);

/* ---- sparse layer zoo of the SparK encoder for other backbones (P/encoder3D.py = nnunetv2/training/nnUNetTrainer/
 * variants/pretrain/encoder3D.py).  Same tensors as above: channels-last [B][D][H][W][C] + uint8 patch mask (fd, fh, fw) + block shift;
 * inactive voxels hold don't-care bits, are read as zeros (the reference keeps explicit zeros there) and are never written.  The
 * streaming kernels walk the ACTIVE voxels through the active-patch list (am_mask_compact): mask != NULL requires it (else -2).
 * mask == NULL: dense tensor. */

/* Per-voxel normalisation over channel groups.  kind 0: y = (x - mean_g) * rsqrt(var_g + eps) * gamma_c + beta_c with the statistics
 * of the voxel's own channel group (biased variance) -- SparseConvNeXtLayerNorm (encoder3D.py:181-232; groups = 1) and SparseGroupNorm
 * (:47-78: nn.GroupNorm applied to the (N, C) matrix of active voxels, hence per voxel).  kind 1: SparseGRN's sparse branch
 * (:116-127): G = ||x||_2 over C, y = gamma_c * x * G / (G + 1e-6) + beta_c (beta may be NULL).  C / groups must be a multiple of the
 * 16-byte chunk (8 bf16 / 4 f32 channels) or divide it.  bwd recomputes the statistics; dgamma / dbeta are ACCUMULATED with fp32
 * atomics into AM_LAYER_REP replicated rows [AM_LAYER_REP][C] that the caller zeroes and sums (a thousand workgroups adding into the
 * same two cache lines serialise in L2: 278 us -> 40 us for a 54 MB tensor). */
#define AM_LAYER_REP 64
int am_voxel_norm_fwd(int dtype, int kind, const void* x, void* y, int B, int D, int H, int W, int C, int groups, const float* gamma,
                      const float* beta, float eps, const uint8_t* mask, int bshift, const int32_t* active_list, int n_active, void* stream);
int am_voxel_norm_bwd(int dtype, int kind, const void* x, const void* dy, void* dx, int B, int D, int H, int W, int C, int groups,
                      const float* gamma, float eps, float* dgamma_accum, float* dbeta_accum, const uint8_t* mask, int bshift,
                      const int32_t* active_list, int n_active, void* stream);

/* SparseMaxPooling / SparseAvgPooling (encoder3D.py:31-36): nn.MaxPool3d / nn.AvgPool3d (cubic kernel; max: any dilation) of the
 * zero-filled tensor, output written at active output voxels.  (Do, Ho, Wo) are torch's extents for ceil_mode False or True (the
 * caller chooses; a ceil_mode window that hangs over the end sees only what exists, and an average with count_include_pad divides by
 * the window clipped to the padded extent, as torch does).  op 0 = max (argmax: int32 [.. Do Ho Wo C] input voxel index inside
 * the sample, first maximum in scan order like torch; needed by the backward), 1 = average.  The mask is shared: in_bshift /
 * out_bshift are the block shifts at the input / output resolution.  bwd is a gather over the windows covering each active INPUT voxel
 * (no atomics). */
int am_pool3d_fwd(int dtype, int op, const void* x, void* y, int32_t* argmax, int B, int Di, int Hi, int Wi, int C, int ksize, int stride,
                  int pad, int dilation, int count_include_pad, int Do, int Ho, int Wo, const uint8_t* mask, int in_bshift, int out_bshift,
                  int fd, int fh, int fw, const int32_t* active_list, int n_active, void* stream);
int am_pool3d_bwd(int dtype, int op, const void* dy, const int32_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int C, int ksize,
                  int stride, int pad, int dilation, int count_include_pad, int Do, int Ho, int Wo, const uint8_t* mask, int in_bshift,
                  int out_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active, void* stream);

/* Depthwise convolution k in {3, 5, 7}, stride 1, padding k/2 (SparseConvNeXtBlock.dwconv encoder3D.py:247 under sp_conv_forward
 * :12-15; MedNeXt blocks): w fp32 [C][k^3] (torch (C,1,k,k,k)).  data_grad = 1: the data gradient (x = dy, taps mirrored, no bias).
 * am_dwconv3d_wgrad ACCUMULATES dw [C][k^3] and db [C] (db may be NULL). */
int am_dwconv3d(int dtype, int data_grad, const void* x, const float* w, const float* bias, void* y, int B, int D, int H, int W, int C,
                int ksize, const uint8_t* mask, int bshift, int fd, int fh, int fw, void* stream);
int am_dwconv3d_wgrad(int dtype, const void* x, const void* dy, float* dw_accum, float* db_accum, int B, int D, int H, int W, int C, int ksize,
                      const uint8_t* mask, int bshift, int fd, int fh, int fw, void* stream);

/* The same with stride 2 (MedNeXtDownBlock.conv1, P/MedNeXt_head.py:335-341; even fine extents Df, Hf, Wf; output Df/2 ...): direct
 * kernels over the active voxels.  data_grad = 0: src = x (fine), dst = y (coarse); 1: src = dy (coarse), dst = dx (fine).
 * fine_bshift: the mask's block shift at the FINE resolution (>= 1). */
int am_dwconv3d_s2(int dtype, int data_grad, const void* src, const float* w, const float* bias, void* dst, int B, int Df, int Hf, int Wf,
                   int C, int ksize, const uint8_t* mask, int fine_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active,
                   void* stream);
int am_dwconv3d_s2_wgrad(int dtype, const void* x, const void* dy, float* dw_accum, float* db_accum, int B, int Df, int Hf, int Wf, int C,
                         int ksize, const uint8_t* mask, int fine_bshift, int fd, int fh, int fw, const int32_t* active_list, int n_active,
                         void* stream);

/* Pointwise tail of SparseConvNeXtBlock.forward (encoder3D.py:262-275).  am_gelu: dy == NULL: out = GELU(x) (erf form, nn.GELU());
 * else out = dy * GELU'(x).  am_scale_residual: backward == 0: out = other + gamma_c * x (layer scale + residual; gamma NULL = 1);
 * backward == 1: out = gamma_c * other (other = dy) and dgamma_accum[r][c] += sum_v other * x (AM_LAYER_REP rows, as above). */
int am_gelu(int dtype, const void* x, const void* dy, void* out, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift,
            const int32_t* active_list, int n_active, void* stream);
int am_scale_residual(int dtype, int backward, const void* x, const void* other, const float* gamma, void* out, float* dgamma_accum, int B, int D,
                      int H, int W, int C, const uint8_t* mask, int bshift, const int32_t* active_list, int n_active, void* stream);

/* SparseAdaptiveAvgPooling (encoder3D.py:171-179): mean[b][c] = sum over the active voxels of sample b / (their number + 1e-6).
 * bwd: dx = dmean[b][c] / (count[b] + 1e-6) on the active voxels; count: fp32 [B] active voxels per sample. */
int am_masked_mean_fwd(int dtype, const void* x, float* mean, int B, int D, int H, int W, int C, const uint8_t* mask, int bshift, int fd, int fh,
                       int fw, void* stream);
int am_masked_mean_bwd(int dtype, const float* dmean, const float* count, void* dx, int B, int D, int H, int W, int C, const uint8_t* mask,
                       int bshift, const int32_t* active_list, int n_active, void* stream);

#ifdef __cplusplus
}
#endif
#endif
