"""Sparse layer zoo of the SparK encoder for backbones other than STUNet, on the HIP kernels of csrc/layer_ops.hip.

Mirror of P/encoder3D.py (= nnunetv2/training/nnUNetTrainer/variants/pretrain/encoder3D.py): same class names, constructor
arguments, parameter names / shapes (so `state_dict`s interchange) and forward semantics, including the side channel: the modules
read the current patch mask from `anatomask_amd.modules._cur_active` (bool (B,1,f,f,f); the reference: encoder3D._cur_active :5), and
`dense_model_to_sparse` (:300-364) swaps the layers of a dense torch model for these.

    SparseConv3d (:27-28)                      channel-mixing conv -> am_conv3d (matrix cores); depthwise (groups == C) -> am_dwconv3d
    SparseMaxPooling / SparseAvgPooling (:31-36)            am_pool3d_fwd / bwd
    SparseBatchNorm3d / SparseSyncBatchNorm3d (:39-44)      BatchNorm over the active voxels: am_chan_stats / am_norm_* (the STUNet path's)
    SparseGroupNorm (:47-78), SparseConvNeXtLayerNorm (:181-232), SparseGRN (:100-135)   am_voxel_norm_fwd / bwd
    SparseAdaptiveAvgPooling (:171-179)                     am_masked_mean_fwd / bwd
    SparseConvNeXtBlock (:235-276)                          dwconv -> LayerNorm -> 1x1 -> GELU -> 1x1 -> layer scale -> mask -> residual

Tensors cross the module boundary as torch NCDHW tensors like the reference's; inside they are channels-last ([B,D,H,W,C], the
kernels' layout) -- a module returns an NCDHW *view* of its channels-last result (torch.channels_last_3d strides), so chains of these
modules convert nothing.  Outputs carry exact zeros at inactive voxels, as the reference's.  Compute dtype = the input's (fp32 or
bf16).  No CPU / torch fallback: without the HIP library the import of `ops` fails.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import modules as _M
from . import ops
from .hip import lib as _lib


# ------------------------------------------------------------------ plumbing
LAYER_REP = 64   # AM_LAYER_REP: replicated rows of the per-channel gradient accumulators
COMPUTE_DTYPE = torch.float32   # activation dtype a Cin = 1 stem produces from the fp32 image (torch.bfloat16: bf16 storage downstream)
_MASK_CACHE = {"act": None, "key": None, "mi": None}


def current_mask(device) -> Optional[ops.MaskInfo]:
    """MaskInfo of `modules._cur_active` (built once per mask tensor), or None (dense).
    The cache holds the tensor ITSELF and compares by identity: an `id()` alone is reused by the next tensor object once the old one is
    collected (same id, same _version 0, other contents: a stale patch list -- found as a sporadic pooling test failure in round 4)."""
    act = _M._cur_active
    if act is None:
        return None
    key = (act._version, str(device))
    if _MASK_CACHE["act"] is not act or _MASK_CACHE["key"] != key:
        _MASK_CACHE["act"], _MASK_CACHE["key"], _MASK_CACHE["mi"] = act, key, ops.MaskInfo.from_bool(act, device)
    return _MASK_CACHE["mi"]


class _all_active:
    """`with _all_active(x_cl):` -- the zoo's kernels walk an active-patch list; a DENSE branch (`sparse=False`, encoder3D.py:60-61,197-198)
    is the same kernel under a mask with every patch active (grid = the tensor's extents / their largest common power of two <= 16)."""
    _cache = {}

    def __init__(self, x_cl: torch.Tensor):
        B, D, H, W = x_cl.shape[:4]
        g = 16
        while g > 1 and (D % g or H % g or W % g):
            g //= 2
        key = (B, D // g, H // g, W // g, str(x_cl.device))
        if key not in self._cache:
            self._cache[key] = torch.ones(B, 1, D // g, H // g, W // g, dtype=torch.bool, device=x_cl.device)
        self.act = self._cache[key]

    def __enter__(self):
        self.prev, _M._cur_active = _M._cur_active, self.act

    def __exit__(self, *a):
        _M._cur_active = self.prev


def _bshift(mi: Optional[ops.MaskInfo], D: int) -> int:
    if mi is None:
        return 0
    r = D // mi.fd
    assert r >= 1 and r * mi.fd == D and (r & (r - 1)) == 0, f"resolution {D} is not a power-of-two multiple of the mask grid {mi.fd}"
    return r.bit_length() - 1


def _cl(x: torch.Tensor) -> torch.Tensor:
    """NCDHW (any strides) -> contiguous [B,D,H,W,C] (a view when x is channels_last_3d)."""
    assert x.dim() == 5 and x.dtype in (torch.float32, torch.bfloat16), (x.shape, x.dtype)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _nc(y_cl: torch.Tensor) -> torch.Tensor:
    return y_cl.permute(0, 4, 1, 2, 3)


def _geo(mi, bs):
    mp, fd, fh, fw = ops._mk(mi)
    return mp, bs, fd, fh, fw


def _al(mi):
    return ops._al(mi)


def _s():
    return ops._stream()


def _check_list(mi):
    if mi is not None and ops._al(mi)[0] is None:
        raise RuntimeError("the sparse layer kernels walk the active-patch list: mask grids above 255 per dim / empty masks are not supported")


# ------------------------------------------------------------------ per-voxel norms (LayerNorm / GroupNorm / GRN)
class _VoxelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, gamma, beta, kind, groups, eps):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        _check_list(mi)
        y = torch.zeros_like(x_cl)
        g32, b32 = gamma.float().contiguous().view(-1), (beta.float().contiguous().view(-1) if beta is not None else None)
        _lib().voxel_norm_fwd(ops._dt(x_cl), kind, x_cl.data_ptr(), y.data_ptr(), B, D, H, W, C, groups, g32.data_ptr(), ops._p(b32), float(eps),
                              ops._mk(mi)[0], bs, *_al(mi), _s())
        ctx.save_for_backward(x_cl, g32)
        ctx.cfg = (mi, bs, kind, groups, eps, gamma.shape, None if beta is None else beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, g32 = ctx.saved_tensors
        mi, bs, kind, groups, eps, gshape, bshape = ctx.cfg
        B, D, H, W, C = x_cl.shape
        dy = dy.contiguous()
        dx = torch.zeros_like(x_cl)
        dg = torch.zeros(LAYER_REP, C, device=x_cl.device, dtype=torch.float32)
        db = torch.zeros(LAYER_REP, C, device=x_cl.device, dtype=torch.float32)
        _lib().voxel_norm_bwd(ops._dt(x_cl), kind, x_cl.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, D, H, W, C, groups, g32.data_ptr(), float(eps),
                              dg.data_ptr(), db.data_ptr(), ops._mk(mi)[0], bs, *_al(mi), _s())
        return dx, dg.sum(0).view(gshape), (db.sum(0).view(bshape) if bshape is not None else None), None, None, None


class SparseGroupNorm(nn.GroupNorm):
    """encoder3D.py:47-78: GroupNorm of the (N_active, C) matrix, i.e. per voxel over each channel group."""

    def __init__(self, num_groups, num_channels, eps=1e-6, sparse=True):
        super().__init__(num_groups, num_channels, eps)
        self.sparse = sparse

    def forward(self, x):
        if x.ndim != 5 or not self.sparse:
            raise NotImplementedError("SparseGroupNorm: 5-D sparse input only")
        return _nc(_VoxelNormFn.apply(_cl(x), self.weight, self.bias, 0, self.num_groups, self.eps))


class SparseConvNeXtLayerNorm(nn.LayerNorm):
    """encoder3D.py:181-232: LayerNorm over C at every active voxel; channels_last input (B,H,W,D,C) or channels_first (B,C,H,W,D)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last", sparse=True):
        if data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        super().__init__(normalized_shape, eps, elementwise_affine=True)
        self.data_format, self.sparse = data_format, sparse

    def forward(self, x):
        if x.ndim != 5:
            raise NotImplementedError("SparseConvNeXtLayerNorm: 5-D input only")
        if not self.sparse:                             # encoder3D.py:197-198 / :207-212: plain LayerNorm over C at EVERY voxel
            xc = x.contiguous() if self.data_format == "channels_last" else _cl(x)
            with _all_active(xc):
                y = _VoxelNormFn.apply(xc, self.weight, self.bias, 0, 1, self.eps)
            return y if self.data_format == "channels_last" else _nc(y)
        if self.data_format == "channels_last":
            return _VoxelNormFn.apply(x.contiguous(), self.weight, self.bias, 0, 1, self.eps)
        return _nc(_VoxelNormFn.apply(_cl(x), self.weight, self.bias, 0, 1, self.eps))


class SparseGRN(nn.Module):
    """encoder3D.py:100-135, sparse branch (:116-127): on the (N_active, C) matrix Gx = ||row||_2, Nx = Gx / (Gx.mean(-1) + 1e-6)
    with the mean over a size-1 axis, so Nx = Gx / (Gx + 1e-6); y = gamma * (x * Nx) + beta.  Input channels-last (B,H,W,D,C).
    (The reference derives the mask resolution from x.shape[2:5] of that channels-last tensor, :118, which is only right when C equals
    the spatial size; here the mask is taken at the tensor's true resolution.)"""

    def __init__(self, dim, use_bias=True, sparse=True):
        super().__init__()
        self.use_bias, self.sparse = use_bias, sparse
        self.gamma = nn.Parameter(torch.zeros(1, dim))
        if self.use_bias:
            self.beta = nn.Parameter(torch.zeros(1, dim))

    def forward(self, x):
        if x.ndim != 5:
            raise NotImplementedError("SparseGRN supports only 5D tensors")
        if not self.sparse:
            raise NotImplementedError("SparseGRN: sparse branch only")
        return _VoxelNormFn.apply(x.contiguous(), self.gamma, self.beta if self.use_bias else None, 1, 1, 0.0)


# ------------------------------------------------------------------ pooling
class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, op, k, s, p, cip, dil=1, ceil_mode=False):
        mi = current_mask(x_cl.device)
        B, Di, Hi, Wi, C = x_cl.shape
        Do, Ho, Wo = (_pool_out(n, k, s, p, dil, ceil_mode) for n in (Di, Hi, Wi))
        bi, bo = _bshift(mi, Di), _bshift(mi, Do)
        _check_list(mi)
        y = torch.zeros(B, Do, Ho, Wo, C, device=x_cl.device, dtype=x_cl.dtype)
        idx = torch.empty(B, Do, Ho, Wo, C, device=x_cl.device, dtype=torch.int32) if op == 0 else None
        mp, fd, fh, fw = ops._mk(mi)
        _lib().pool3d_fwd(ops._dt(x_cl), op, x_cl.data_ptr(), y.data_ptr(), ops._p(idx), B, Di, Hi, Wi, C, k, s, p, dil, int(cip), Do, Ho, Wo,
                          mp, bi, bo, fd, fh, fw, *_al(mi), _s())
        ctx.save_for_backward(idx if idx is not None else torch.empty(0))
        ctx.cfg = (mi, op, k, s, p, dil, cip, (B, Di, Hi, Wi, C), (Do, Ho, Wo), bi, bo, x_cl.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        mi, op, k, s, p, dil, cip, (B, Di, Hi, Wi, C), (Do, Ho, Wo), bi, bo, dt = ctx.cfg
        dy = dy.contiguous()
        dx = torch.zeros(B, Di, Hi, Wi, C, device=dy.device, dtype=dt)
        mp, fd, fh, fw = ops._mk(mi)
        _lib().pool3d_bwd(ops._dt(dx), op, dy.data_ptr(), idx.data_ptr() if op == 0 else None, dx.data_ptr(), B, Di, Hi, Wi, C, k, s, p, dil, int(cip),
                          Do, Ho, Wo, mp, bi, bo, fd, fh, fw, *_al(mi), _s())
        return dx, None, None, None, None, None, None, None


def _pool_out(n, k, s, p, dil, ceil_mode):
    """Output extent of torch's pooling layers (ceil_mode: the last window must start inside the input or its left padding)."""
    num = n + 2 * p - dil * (k - 1) - 1
    o = (-(-num // s) if ceil_mode else num // s) + 1
    if ceil_mode and (o - 1) * s >= n + p:
        o -= 1
    return o


def _one(v):
    if isinstance(v, (tuple, list)):
        assert all(a == v[0] for a in v), f"cubic pooling windows only, got {v}"
        return int(v[0])
    return int(v)


class SparseMaxPooling(nn.MaxPool3d):
    """encoder3D.py:31-32 (sp_conv_forward :12-15): MaxPool3d, then the output is masked."""

    def forward(self, x):
        if self.return_indices:
            raise NotImplementedError("SparseMaxPooling: return_indices is not supported (sp_conv_forward could not mask a tuple either)")
        k = _one(self.kernel_size)
        return _nc(_PoolFn.apply(_cl(x), 0, k, _one(self.stride if self.stride is not None else k), _one(self.padding), True, _one(self.dilation),
                                 bool(self.ceil_mode)))


class SparseAvgPooling(nn.AvgPool3d):
    """encoder3D.py:35-36."""

    def forward(self, x):
        k = _one(self.kernel_size)
        if self.divisor_override is not None:           # window SUM / divisor: the count_include_pad average (sum / k^3) rescaled
            if self.ceil_mode:
                raise NotImplementedError("SparseAvgPooling: divisor_override together with ceil_mode (windows clipped at the padded end)")
            y = _PoolFn.apply(_cl(x), 1, k, _one(self.stride if self.stride is not None else k), _one(self.padding), True, 1, bool(self.ceil_mode))
            return _nc(y) * (float(k ** 3) / float(self.divisor_override))
        return _nc(_PoolFn.apply(_cl(x), 1, k, _one(self.stride if self.stride is not None else k), _one(self.padding), self.count_include_pad, 1,
                                 bool(self.ceil_mode)))


class _AdaptiveAvgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        mean = torch.empty(B, C, device=x_cl.device, dtype=torch.float32)
        mp, fd, fh, fw = ops._mk(mi)
        _lib().masked_mean_fwd(ops._dt(x_cl), x_cl.data_ptr(), mean.data_ptr(), B, D, H, W, C, mp, bs, fd, fh, fw, _s())
        ctx.cfg = (mi, bs, x_cl.shape, x_cl.dtype)
        return mean.to(x_cl.dtype)

    @staticmethod
    def backward(ctx, dmean):
        mi, bs, (B, D, H, W, C), dt = ctx.cfg
        _check_list(mi)
        if mi is None:
            cnt = torch.full((B,), float(D * H * W), device=dmean.device)
        else:
            cnt = mi.t.reshape(B, -1).sum(1).float() * float(1 << (3 * bs))
        dx = torch.zeros(B, D, H, W, C, device=dmean.device, dtype=dt)
        _lib().masked_mean_bwd(ops._dt(dx), dmean.float().contiguous().data_ptr(), cnt.data_ptr(), dx.data_ptr(), B, D, H, W, C, ops._mk(mi)[0], bs,
                               *_al(mi), _s())
        return dx


class SparseAdaptiveAvgPooling(nn.AdaptiveAvgPool3d):
    """encoder3D.py:171-179: masked mean over the active voxels of each sample -> (B,C,1,1,1)."""

    def __init__(self, output_size, sparse=True):
        super().__init__(output_size)
        self.sparse = sparse

    def forward(self, x):
        return _AdaptiveAvgFn.apply(_cl(x))[:, :, None, None, None]


# ------------------------------------------------------------------ BatchNorm over the active voxels
class _SparseBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, weight, bias, run_mean, run_var, nbt, training, momentum, eps, sync):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        st = ops.NormStats(C, x_cl.device)
        g32, b32 = weight.float().contiguous(), bias.float().contiguous()
        use_batch = training or run_mean is None
        if use_batch:
            ops.chan_stats(x_cl, mi, bs, st)
            if mi is None:
                n = float(B * D * H * W)
            else:
                n = float(int(mi.t.count_nonzero().item()) if mi.n_active is None else mi.n_active) * float(1 << (3 * bs))
            if sync:
                import torch.distributed as dist
                cnt = torch.tensor([n], device=x_cl.device, dtype=torch.float64)
                dist.all_reduce(st.sums[:ops.NREP]); dist.all_reduce(cnt)
                n = float(cnt.item()); st.sync_world = dist.get_world_size()
            st.count_host, st.count_ptr = n, None
            mom = momentum
            if training and run_mean is not None:
                nbt += 1
                if momentum is None:
                    mom = 1.0 / float(nbt.item())
            ops.norm_finalize(st, g32, b32, eps, run_mean if training else None, run_var if training else None, 0.0 if mom is None else mom)
        else:
            ops.norm_fold_running(st, g32, b32, run_mean, run_var, eps)
        y = ops.norm_apply(x_cl, st, ops.ACT_NONE, mi, bs, out=torch.zeros_like(x_cl))
        ctx.save_for_backward(x_cl, g32)
        ctx.cfg = (mi, bs, st, use_batch)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, g32 = ctx.saved_tensors
        mi, bs, st, use_batch = ctx.cfg
        C = x_cl.shape[-1]
        dg = torch.zeros(C, device=x_cl.device, dtype=torch.float32)
        db = torch.zeros(C, device=x_cl.device, dtype=torch.float32)
        if not use_batch:
            raise NotImplementedError("SparseBatchNorm3d backward in eval mode")
        dx = ops.norm_backward(dy.contiguous(), None, x_cl, st, g32, ops.ACT_NONE, mi, bs, dg, db, dx=torch.zeros_like(x_cl))
        return dx, dg, db, None, None, None, None, None, None, None


class _DensifyFn(torch.autograd.Function):
    """P/AnatoMask.py:158-163 in one pair of kernels: pooled sparse InstanceNorm of the active voxels, mask token everywhere else
    (`densify_norms[i]` + `torch.where(active, x, mask_token)`): am_chan_stats -> am_norm_finalize -> am_norm_apply(fill = token), and
    backward am_norm_bwd_reduce / apply with the token gradient = the sum of dy over the inactive voxels -- what the fused STUNet
    engine runs for its densify levels (engine.densify_forward / densify_backward)."""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, token, eps):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        st = ops.NormStats(C, x_cl.device)
        g32, b32, t32 = weight.float().contiguous(), bias.float().contiguous(), token.float().contiguous().view(-1)
        ops.chan_stats(x_cl, mi, bs, st)
        st.count_host = float(int(mi.t.count_nonzero().item()) if mi.n_active is None else mi.n_active) * float(1 << (3 * bs))
        st.count_ptr = None
        ops.norm_finalize(st, g32, b32, eps)
        y = ops.norm_apply(x_cl, st, ops.ACT_NONE, mi, bs, fill=t32)
        ctx.save_for_backward(x_cl, g32)
        ctx.cfg = (mi, bs, st, token.shape, token.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, g32 = ctx.saved_tensors
        mi, bs, st, tshape, tdtype = ctx.cfg
        C = x_cl.shape[-1]
        dg, db, dt = (torch.zeros(C, device=x_cl.device, dtype=torch.float32) for _ in range(3))
        dx = ops.norm_backward(dy.contiguous(), None, x_cl, st, g32, ops.ACT_NONE, mi, bs, dg, db, dtoken=dt, fill=True, dx=torch.zeros_like(x_cl))
        return dx, dg, db, dt.view(tshape).to(tdtype), None


class _FillTokensFn(torch.autograd.Function):
    """`torch.where(active, x, mask_token)` of P/AnatoMask.py:161-163 for densify norms other than the pooled InstanceNorm (which has its
    own fused pair, _DensifyFn): one am_norm_apply pass with unit scale / zero shift and the token as fill; backward: the gradient passes
    through at the active voxels (inactive ones are don't-care for every sparse consumer), the token gradient is the per-channel sum of
    dy over the INACTIVE voxels = (sum over all) - (sum over the active ones), two am_chan_sum passes."""

    @staticmethod
    def forward(ctx, x_cl, token):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        st = ops.NormStats(C, x_cl.device)
        st.scale.fill_(1.0); st.shift.zero_(); st.mean.zero_(); st.rstd.fill_(1.0)
        y = ops.norm_apply(x_cl, st, ops.ACT_NONE, mi, bs, fill=token.float().contiguous().view(-1))
        ctx.cfg = (mi, bs, token.shape, token.dtype, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        mi, bs, tshape, tdtype, C = ctx.cfg
        dy = dy.contiguous()
        tot, act = torch.zeros(C, device=dy.device, dtype=torch.float32), torch.zeros(C, device=dy.device, dtype=torch.float32)
        ops.chan_sum(dy, None, 0, tot)
        ops.chan_sum(dy, mi, bs, act)
        return dy, (tot - act).view(tshape).to(tdtype)


def fill_tokens(x: torch.Tensor, token: torch.Tensor) -> torch.Tensor:
    """NCDHW map with zeros at the inactive voxels -> the same map with `token` (1, C, 1, 1, 1) there (the current mask: modules._cur_active)."""
    return _nc(_FillTokensFn.apply(_cl(x), token))


class SparseBatchNorm3d(nn.BatchNorm1d):
    """encoder3D.py:39-40 (sp_bn_forward :17-25): BatchNorm1d over the (N_active, C) matrix -- batch statistics over all active voxels
    of the (local) batch, running statistics with the unbiased variance, eval mode on the running statistics."""
    _sync = False

    def forward(self, x):
        if not self.affine:
            raise NotImplementedError("SparseBatchNorm3d: affine=True only")
        track = self.track_running_stats and self.running_mean is not None
        y = _SparseBNFn.apply(_cl(x), self.weight, self.bias, self.running_mean if track else None, self.running_var if track else None,
                              self.num_batches_tracked if track else None, self.training, self.momentum, self.eps,
                              self._sync and self.training and torch.distributed.is_available() and torch.distributed.is_initialized()
                              and torch.distributed.get_world_size() > 1)
        return _nc(y)


class SparseSyncBatchNorm3d(SparseBatchNorm3d):
    """encoder3D.py:43-44: the statistics are all-reduced over the process group (RCCL)."""
    _sync = True


# ------------------------------------------------------------------ convolutions
class _DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, weight, bias, k):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        w32 = weight.float().contiguous().view(C, -1)
        y = torch.zeros_like(x_cl)
        mp, fd, fh, fw = ops._mk(mi)
        _lib().dwconv3d(ops._dt(x_cl), 0, x_cl.data_ptr(), w32.data_ptr(), ops._p(bias.float().contiguous() if bias is not None else None),
                        y.data_ptr(), B, D, H, W, C, k, mp, bs, fd, fh, fw, _s())
        ctx.save_for_backward(x_cl, w32)
        ctx.cfg = (mi, bs, k, weight.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, w32 = ctx.saved_tensors
        mi, bs, k, wshape, has_bias = ctx.cfg
        B, D, H, W, C = x_cl.shape
        dy = dy.contiguous()
        mp, fd, fh, fw = ops._mk(mi)
        dx = torch.zeros_like(x_cl)
        _lib().dwconv3d(ops._dt(x_cl), 1, dy.data_ptr(), w32.data_ptr(), None, dx.data_ptr(), B, D, H, W, C, k, mp, bs, fd, fh, fw, _s())
        dw = torch.zeros(C, k ** 3, device=x_cl.device, dtype=torch.float32)
        db = torch.zeros(C, device=x_cl.device, dtype=torch.float32) if has_bias else None
        _lib().dwconv3d_wgrad(ops._dt(x_cl), x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), ops._p(db), B, D, H, W, C, k, mp, bs, fd, fh, fw, _s())
        return dx, dw.view(wshape), db, None


class _DwConvS2Fn(torch.autograd.Function):
    """depthwise convolution with stride 2 (MedNeXtDownBlock.conv1): am_dwconv3d_s2 / am_dwconv3d_s2_wgrad."""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, k):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        _check_list(mi)
        if mi is not None and bs < 1:
            raise RuntimeError("strided sparse conv at the mask-grid resolution: the output would be finer than the patch mask")
        w32 = weight.float().contiguous().view(C, -1)
        y = torch.zeros(B, D // 2, H // 2, W // 2, C, device=x_cl.device, dtype=x_cl.dtype)
        mp, fd, fh, fw = ops._mk(mi)
        _lib().dwconv3d_s2(ops._dt(x_cl), 0, x_cl.data_ptr(), w32.data_ptr(), ops._p(bias.float().contiguous() if bias is not None else None),
                           y.data_ptr(), B, D, H, W, C, k, mp, bs, fd, fh, fw, *_al(mi), _s())
        ctx.save_for_backward(x_cl, w32)
        ctx.cfg = (mi, bs, k, weight.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, w32 = ctx.saved_tensors
        mi, bs, k, wshape, has_bias = ctx.cfg
        B, D, H, W, C = x_cl.shape
        dy = dy.contiguous()
        mp, fd, fh, fw = ops._mk(mi)
        dx = torch.zeros_like(x_cl)
        _lib().dwconv3d_s2(ops._dt(x_cl), 1, dy.data_ptr(), w32.data_ptr(), None, dx.data_ptr(), B, D, H, W, C, k, mp, bs, fd, fh, fw, *_al(mi), _s())
        dw = torch.zeros(C, k ** 3, device=x_cl.device, dtype=torch.float32)
        db = torch.zeros(C, device=x_cl.device, dtype=torch.float32) if has_bias else None
        _lib().dwconv3d_s2_wgrad(ops._dt(x_cl), x_cl.data_ptr(), dy.data_ptr(), dw.data_ptr(), ops._p(db), B, D, H, W, C, k, mp, bs, fd, fh, fw,
                                 *_al(mi), _s())
        return dx, dw.view(wshape), db, None


class _StemConvFn(torch.autograd.Function):
    """Cin = 1 convolution k in {1, 3}, stride 1, on the stage-0 tensor (16^3 patches): am_stem_conv_fwd / am_stem_conv_wgrad.  The
    input is the (masked) image: no data gradient is produced."""

    @staticmethod
    def forward(ctx, x_b1, weight, bias, k, dtype):
        mi = current_mask(x_b1.device)
        if mi is None or _bshift(mi, x_b1.shape[1]) != 4:
            raise NotImplementedError("Cin = 1 sparse conv: the stem kernels need the patch mask at 16^3-voxel patches")
        B, D, H, W = x_b1.shape
        y = torch.zeros(B, D, H, W, weight.shape[0], device=x_b1.device, dtype=dtype)           # zeros at inactive voxels, like every layer here
        ops.stem_conv_fwd(x_b1, weight.detach().float().contiguous(), bias.detach().float().contiguous() if bias is not None else None, mi, 4, dtype, out=y)
        ctx.save_for_backward(x_b1)
        ctx.cfg = (mi, k, weight.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_b1,) = ctx.saved_tensors
        mi, k, wshape, has_bias = ctx.cfg
        C = dy.shape[-1]
        dw = torch.zeros(C, k ** 3, device=dy.device, dtype=torch.float32)
        db = torch.zeros(C, device=dy.device, dtype=torch.float32) if has_bias else None
        ops.stem_conv_wgrad(x_b1, dy.contiguous(), k, mi, 4, dw, db)
        return None, dw.view(wshape), db, None, None


class _ConvFn(torch.autograd.Function):
    """channel-mixing convolution k in {1, 3}, stride in {1, 2} on the matrix cores (am_conv3d / am_conv3d_wgrad)."""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, k, stride):
        mi = current_mask(x_cl.device)
        B, Di, Hi, Wi, Cin = x_cl.shape
        so = tuple((v + 2 * (k // 2) - k) // stride + 1 for v in (Di, Hi, Wi))
        bi, bo = _bshift(mi, Di), _bshift(mi, so[0])
        wp = ops.pack_weight(weight.detach().float(), x_cl.dtype, False, False)
        y = torch.zeros(B, *so, weight.shape[0], device=x_cl.device, dtype=x_cl.dtype)
        ops.conv3d(ops.CONV_FWD, x_cl, wp, bias.float().contiguous() if bias is not None else None, so, k, stride, in_mask=mi, in_bshift=bi,
                   out_mask=mi, out_bshift=bo, out=y)
        ctx.save_for_backward(x_cl, weight)
        ctx.cfg = (mi, bi, bo, k, stride, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, weight = ctx.saved_tensors
        mi, bi, bo, k, stride, has_bias = ctx.cfg
        dy = dy.contiguous()
        B, Di, Hi, Wi, Cin = x_cl.shape
        wpd = ops.pack_weight(weight.detach().float(), x_cl.dtype, False, True)
        dx = torch.zeros_like(x_cl)
        ops.conv3d(ops.CONV_DGRAD, dy, wpd, None, (Di, Hi, Wi), k, stride, in_mask=mi, in_bshift=bo, out_mask=mi, out_bshift=bi, out=dx)
        dwp = ops.conv3d_wgrad(ops.CONV_FWD, x_cl, dy, k, stride, x_mask=mi, x_bshift=bi, y_mask=mi, y_bshift=bo)
        dw = torch.zeros(weight.shape, device=x_cl.device, dtype=torch.float32)
        ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
        db = None
        if has_bias:
            db = torch.zeros(weight.shape[0], device=x_cl.device, dtype=torch.float32)
            ops.chan_sum(dy, mi, bo, db)
        return dx, dw, db, None, None


def dense_conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], k: int) -> torch.Tensor:
    """plain dense k in {1, 3} stride-1 convolution of an NCDHW tensor on the matrix cores (SparK's densify projections,
    P/AnatoMask.py:63-65): the patch mask of the side channel is ignored."""
    saved = _M._cur_active
    _M._cur_active = None
    try:
        return _nc(_ConvFn.apply(_cl(x), weight, bias, k, 1))
    finally:
        _M._cur_active = saved


class SparseConv3d(nn.Conv3d):
    """encoder3D.py:27-28 (sp_conv_forward :12-15): Conv3d, output masked.  Supported: 'same' padding k//2, dilation 1, zero padding;
    groups == 1 with k in {1, 3}, stride in {1, 2} (matrix cores); depthwise (groups == in == out) with k in {3, 5, 7}, stride 1 or 2;
    Cin = 1 stems k in {1, 3} (output dtype = sparse_layers.COMPUTE_DTYPE: the image itself stays fp32)."""

    def forward(self, x):
        k, s, p = _one(self.kernel_size), _one(self.stride), _one(self.padding)
        if _one(self.dilation) != 1 or self.padding_mode != "zeros" or p != k // 2:
            raise NotImplementedError("SparseConv3d: dilation 1, zero 'same' padding only")
        if self.groups == 1 and k in (1, 3) and s in (1, 2) and self.in_channels % 8 == 0 and self.out_channels % 8 == 0:
            return _nc(_ConvFn.apply(_cl(x), self.weight, self.bias, k, s))
        if self.groups == self.in_channels == self.out_channels and k in (3, 5, 7) and s == 1:
            return _nc(_DwConvFn.apply(_cl(x), self.weight, self.bias, k))
        if self.groups == self.in_channels == self.out_channels and k in (3, 5, 7) and s == 2 and self.in_channels % 8 == 0:
            return _nc(_DwConvS2Fn.apply(_cl(x), self.weight, self.bias, k))
        if self.in_channels == 1 and self.groups == 1 and k in (1, 3) and s == 1 and self.out_channels % 8 == 0:
            return _nc(_StemConvFn.apply(x[:, 0].float().contiguous(), self.weight, self.bias, k, COMPUTE_DTYPE))
        raise NotImplementedError(f"SparseConv3d: groups={self.groups} k={k} stride={s} C={self.in_channels}->{self.out_channels} has no kernel")


# ------------------------------------------------------------------ ConvNeXt block
class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        _check_list(mi)
        y = torch.zeros_like(x_cl)
        _lib().gelu(ops._dt(x_cl), x_cl.data_ptr(), None, y.data_ptr(), B, D, H, W, C, ops._mk(mi)[0], bs, *_al(mi), _s())
        ctx.save_for_backward(x_cl)
        ctx.cfg = (mi, bs)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_cl,) = ctx.saved_tensors
        mi, bs = ctx.cfg
        B, D, H, W, C = x_cl.shape
        dx = torch.zeros_like(x_cl)
        _lib().gelu(ops._dt(x_cl), x_cl.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(), B, D, H, W, C, ops._mk(mi)[0], bs, *_al(mi), _s())
        return dx


class _ScaleResidualFn(torch.autograd.Function):
    """y = res + gamma * x on active voxels, res elsewhere (encoder3D.py:266-275: gamma * x, masked, input + drop_path(x))."""

    @staticmethod
    def forward(ctx, x_cl, res_cl, gamma):
        mi = current_mask(x_cl.device)
        B, D, H, W, C = x_cl.shape
        bs = _bshift(mi, D)
        _check_list(mi)
        y = res_cl.clone()
        g32 = gamma.float().contiguous() if gamma is not None else None
        _lib().scale_residual(ops._dt(x_cl), 0, x_cl.data_ptr(), res_cl.data_ptr(), ops._p(g32), y.data_ptr(), None, B, D, H, W, C, ops._mk(mi)[0], bs,
                              *_al(mi), _s())
        ctx.save_for_backward(x_cl, g32 if g32 is not None else torch.empty(0))
        ctx.cfg = (mi, bs, gamma is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, g32 = ctx.saved_tensors
        mi, bs, has_g = ctx.cfg
        B, D, H, W, C = x_cl.shape
        dy = dy.contiguous()
        dx = torch.zeros_like(x_cl)
        dg = torch.zeros(LAYER_REP, C, device=x_cl.device, dtype=torch.float32) if has_g else None
        _lib().scale_residual(ops._dt(x_cl), 1, x_cl.data_ptr(), dy.data_ptr(), g32.data_ptr() if has_g else None, dx.data_ptr(), ops._p(dg), B, D, H, W,
                              C, ops._mk(mi)[0], bs, *_al(mi), _s())
        return dx, dy, (dg.sum(0) if has_g else None)


class SparseGELU(nn.GELU):
    """nn.GELU() of the dense backbones (P/MedNeXt_head.py:291, encoder3D.py:250) on the active voxels only -- GELU(0) = 0, so the
    reference leaves the dense module in place (encoder3D.py:264); here it runs on the HIP kernel like everything else."""

    def forward(self, x):
        if self.approximate != "none":
            raise NotImplementedError("SparseGELU: the erf form only")
        return _nc(_GeluFn.apply(_cl(x)))


class _PointwiseLinear(nn.Linear):
    """nn.Linear applied over the channel axis of a channels-last volume == a 1x1x1 convolution on the matrix cores."""

    def forward(self, x_cl):
        return _ConvFn.apply(x_cl, self.weight[:, :, None, None, None], self.bias, 1, 1)


class DropPath(nn.Module):
    """Stochastic depth per sample, as `timm.models.layers.DropPath` which encoder3D.py:4,253,275 uses (timm is a third-party package
    that is not in /root/reference; this restates its published definition): in training a sample's residual branch is dropped with
    probability `drop_prob` and the kept ones are scaled by 1 / (1 - drop_prob); identity in eval mode.  The draw comes from torch's
    generator of the tensor's device (one value per sample), so a seeded run repeats."""

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = float(drop_prob), scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            r.div_(keep)
        return x * r

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"


class SparseConvNeXtBlock(nn.Module):
    """encoder3D.py:235-276."""

    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6, sparse=True, ks=7):
        super().__init__()
        self.dwconv = SparseConv3d(dim, dim, kernel_size=ks, padding=ks // 2, groups=dim)
        self.norm = SparseConvNeXtLayerNorm(dim, eps=1e-6, sparse=sparse)
        self.pwconv1 = _PointwiseLinear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = _PointwiseLinear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True) if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.sparse = sparse

    def forward(self, x):
        if not self.sparse:
            raise NotImplementedError("SparseConvNeXtBlock: sparse=True only")
        inp = _cl(x)
        k = _one(self.dwconv.kernel_size)
        h = _DwConvFn.apply(inp, self.dwconv.weight, self.dwconv.bias, k)
        h = self.norm(h)                                  # channels-last LayerNorm
        h = self.pwconv1(h)
        h = _GeluFn.apply(h)
        h = self.pwconv2(h)
        # input + drop_path(gamma * h * mask): the per-sample factor commutes with the scale and the mask (encoder3D.py:266-275)
        return _nc(_ScaleResidualFn.apply(self.drop_path(h), inp, self.gamma))


# ------------------------------------------------------------------ converter
def dense_model_to_sparse(m: nn.Module, verbose=False, sbn=False):
    """SparseEncoder.dense_model_to_sparse (encoder3D.py:300-364): recursive swap of a dense model's layers for the sparse ones,
    copying parameters and buffers."""
    oup = m
    if isinstance(m, nn.Conv3d) and not isinstance(m, SparseConv3d):
        if not getattr(m, "skip_sparse_conversion", False):
            bias = m.bias is not None
            oup = SparseConv3d(m.in_channels, m.out_channels, kernel_size=m.kernel_size, stride=m.stride, padding=m.padding, dilation=m.dilation,
                               groups=m.groups, bias=bias, padding_mode=m.padding_mode)
            oup.weight.data.copy_(m.weight.data)
            if bias:
                oup.bias.data.copy_(m.bias.data)
    elif isinstance(m, nn.MaxPool3d) and not isinstance(m, SparseMaxPooling):
        oup = SparseMaxPooling(m.kernel_size, stride=m.stride, padding=m.padding, dilation=m.dilation, return_indices=m.return_indices,
                               ceil_mode=m.ceil_mode)
    elif isinstance(m, nn.AvgPool3d) and not isinstance(m, SparseAvgPooling):
        oup = SparseAvgPooling(m.kernel_size, m.stride, m.padding, ceil_mode=m.ceil_mode, count_include_pad=m.count_include_pad,
                               divisor_override=m.divisor_override)
    elif isinstance(m, nn.GroupNorm) and not isinstance(m, SparseGroupNorm):
        oup = SparseGroupNorm(m.num_groups, m.num_channels, eps=m.eps)         # (the reference does not copy the affine parameters either)
    elif isinstance(m, nn.AdaptiveAvgPool3d) and not isinstance(m, SparseAdaptiveAvgPooling):
        oup = SparseAdaptiveAvgPooling(output_size=(1, 1, 1))
    elif isinstance(m, (nn.BatchNorm3d, nn.SyncBatchNorm)):
        oup = (SparseSyncBatchNorm3d if sbn else SparseBatchNorm3d)(m.weight.shape[0], eps=m.eps, momentum=m.momentum, affine=m.affine,
                                                                    track_running_stats=m.track_running_stats)
        oup.weight.data.copy_(m.weight.data); oup.bias.data.copy_(m.bias.data)
        oup.running_mean.data.copy_(m.running_mean.data); oup.running_var.data.copy_(m.running_var.data)
        oup.num_batches_tracked.data.copy_(m.num_batches_tracked.data)
    elif isinstance(m, nn.LayerNorm) and not isinstance(m, SparseConvNeXtLayerNorm):
        oup = SparseConvNeXtLayerNorm(m.weight.shape[0], eps=m.eps)
        oup.weight.data.copy_(m.weight.data); oup.bias.data.copy_(m.bias.data)
    elif isinstance(m, nn.GELU) and not isinstance(m, SparseGELU):
        oup = SparseGELU()
    elif isinstance(m, nn.Conv1d):
        raise NotImplementedError
    for name, child in m.named_children():
        oup.add_module(name, dense_model_to_sparse(child, verbose=verbose, sbn=sbn))
    del m
    return oup
