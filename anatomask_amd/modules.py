"""Reference-compatible module API over the HIP engine.

Same class names, constructor signatures, attributes and ``state_dict`` keys as the reference
(SURVEY.md 8b), so a ``pretrain_AntoMask.py``-style driver works unchanged:

  STUNet / BasicResBlock      P/STUNet_head.py:8-103
  SparseEncoder               P/encoder3D.py:278-367
  UNetBlock / LightDecoder    P/decoder3D.py:13-84
  SparK                       P/AnatoMask.py:13-277
  ModelEma                    timm.utils.ModelEma (third-party; call sites P/pretrain_AntoMask.py:221,440)
  LocalDDP                    P/pretrain_AntoMask.py:201-207

The torch modules below are parameter containers (names, shapes, initialisers); all arithmetic
runs in the HIP library through ``engine`` -- there is no torch/CPU compute fallback: on a machine
without the library or without a GPU the forwards raise.
"""
from __future__ import annotations

import copy
import math
import sys
from pprint import pformat
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import engine, ops
from .engine import Spec


def _trunc_normal_(t, std=.02, a=-2., b=2.):
    return nn.init.trunc_normal_(t, mean=0., std=std, a=a, b=b)


# The reference's side channel (P/encoder3D.py:5, set by SparK.forward at P/AnatoMask.py:143, read by every Sparse* layer):
# the patch mask (B,1,f,f,f) bool the sparse encoder works under.  Kept with the same name and the same protocol so that code
# calling `model.sparse_encoder(masked)` / `sp_cnn(x, hierarchical=True)` directly (as the reference's users do) keeps working:
# set `modules._cur_active = mask` first.  None = everything active (dense).
_cur_active: Optional[torch.Tensor] = None


def _to_channels_last(t_ncdhw: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return t_ncdhw.permute(0, 2, 3, 4, 1).to(dtype).contiguous()


class _Standalone:
    """Engine context (weights dict, packed-weight cache, gradient dict) of a module used OUTSIDE a SparK (stand-alone
    SparseEncoder / STUNet / LightDecoder): plain views of the module's own torch-layout fp32 parameters."""

    def __init__(self, module: nn.Module, prefix: str, dtype: torch.dtype):
        self.module, self.prefix, self.dtype = module, prefix, dtype
        self.pack = engine.PackCache(dtype)
        self.ptrs = None

    def weights(self) -> Dict[str, torch.Tensor]:
        W = {self.prefix + n: p.data for n, p in self.module.named_parameters()}
        W.update({self.prefix + n: b for n, b in self.module.named_buffers()})
        ptrs = tuple(v.data_ptr() for v in W.values())
        if ptrs != self.ptrs:
            self.pack = engine.PackCache(self.dtype)
            self.ptrs = ptrs
        else:
            self.pack.invalidate()                    # the fp32 masters may have been stepped by an optimizer since the last call
        return W

    def grads(self) -> Dict[str, torch.Tensor]:
        return {self.prefix + n: torch.zeros_like(p.data, dtype=torch.float32) for n, p in self.module.named_parameters()}


class _EncoderFn(torch.autograd.Function):
    """feats = SparseEncoder(x | _cur_active): the 5 stage maps (NCDHW fp32 views, exact zeros at inactive voxels)."""

    @staticmethod
    def forward(ctx, enc: "SparseEncoder", x_b1, mask_info, *params):
        spec, W, pk, G = enc._engine_ctx()
        need_grad = any(ctx.needs_input_grad[3:])
        tape = engine.Tape() if need_grad else None
        counts = engine._counts(mask_info, range(5))
        if tape is not None:
            tape.counts = counts
        feats = engine.encoder_forward(spec, W, pk, x_b1, mask_info, counts, tape)
        ctx.enc, ctx.tape, ctx.inp, ctx.mask = enc, tape, x_b1, mask_info
        outs = []
        for s, f in enumerate(feats):                   # dense view of the block-sparse map: zeros where the reference has zeros
            ident = ops.NormStats(f.shape[-1], f.device)
            ident.scale.fill_(1.0); ident.shift.zero_()
            zero = torch.zeros(f.shape[-1], device=f.device)
            d = ops.norm_apply(f, ident, ops.ACT_NONE, mask_info, 4 - s, fill=zero)
            outs.append(d.permute(0, 4, 1, 2, 3).float())
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        enc = ctx.enc
        if ctx.tape is None:
            raise RuntimeError("backward through a no-grad SparseEncoder forward")
        spec, W, pk, G = enc._engine_ctx(want_grads=True)
        dt = pk.dtype
        dfeat = [None if g is None else _to_channels_last(g, dt) for g in gouts]
        engine.encoder_backward(spec, W, G, pk, ctx.inp, ctx.mask, ctx.tape, dfeat)
        ctx.tape = None
        return (None, None, None, *[G.get(n) for n in enc._param_names()])


class _DecoderFn(torch.autograd.Function):
    """rec = LightDecoder(to_dec): P/decoder3D.py:55-63 on the engine; differentiable wrt the decoder parameters AND to_dec[i]."""

    @staticmethod
    def forward(ctx, dec: "LightDecoder", n_in: int, *args):
        to_dec_ncdhw, params = args[:n_in], args[n_in:]
        spec, W, pk, G = dec._engine_ctx()
        need_grad = any(ctx.needs_input_grad[2:])
        tape = engine.Tape() if need_grad else None
        to_dec = [_to_channels_last(t, pk.dtype) for t in to_dec_ncdhw]
        rec = engine.decoder_forward(spec, W, pk, to_dec, dec.training, tape, fuse_eval=False if need_grad else None)
        ctx.dec, ctx.tape, ctx.n_in = dec, tape, n_in
        return rec.unsqueeze(1)

    @staticmethod
    def backward(ctx, drec):
        dec = ctx.dec
        if ctx.tape is None:
            raise RuntimeError("backward through a no-grad LightDecoder forward")
        spec, W, pk, G = dec._engine_ctx(want_grads=True)
        dproj = engine.decoder_backward(spec, W, G, pk, ctx.tape, drec[:, 0].float().contiguous())
        engine._join_side(drec.device)
        ctx.tape = None
        gin = [g.permute(0, 4, 1, 2, 3).float() for g in dproj]
        return (None, None, *gin, *[G.get(n) for n in dec._param_names()])


# --------------------------------------------------------------------------- backbone
class BasicResBlock(nn.Module):
    """conv3-IN-LReLU-conv3-IN (+1x1 shortcut) -add-LReLU, P/STUNet_head.py:78-103 (parameters only)."""

    def __init__(self, input_channels, output_channels, kernel_size=3, padding=1, stride=1, use_1x1conv=False):
        super().__init__()
        self.conv1 = nn.Conv3d(input_channels, output_channels, kernel_size, stride=stride, padding=padding)
        self.norm1 = nn.InstanceNorm3d(output_channels, affine=True)
        self.act1 = nn.LeakyReLU(inplace=True)
        self.conv2 = nn.Conv3d(output_channels, output_channels, kernel_size, padding=padding)
        self.norm2 = nn.InstanceNorm3d(output_channels, affine=True)
        self.act2 = nn.LeakyReLU(inplace=True)
        self.conv3 = nn.Conv3d(input_channels, output_channels, kernel_size=1, stride=stride) if use_1x1conv else None


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.deep_supervision = True


class STUNet(nn.Module):
    """Encoder-only STUNet, P/STUNet_head.py:8-76."""

    def __init__(self, input_channels, num_classes, depth=[1, 1, 1, 1, 1, 1], dims=[32, 64, 128, 256, 512, 512],
                 pool_op_kernel_sizes=None, conv_kernel_sizes=None, enable_deep_supervision=True):
        super().__init__()
        self.conv_op = nn.Conv3d
        self.input_channels, self.num_classes = input_channels, num_classes
        self.final_nonlin = lambda x: x
        self.decoder = _Decoder()
        self.decoder.deep_supervision = enable_deep_supervision
        self.upscale_logits = False
        self.dims, self.depth = list(dims), list(depth)
        pool_op_kernel_sizes = pool_op_kernel_sizes or [[2, 2, 2]] * 4 + [[1, 1, 1]]
        conv_kernel_sizes = conv_kernel_sizes or [[3, 3, 3]] * 6
        self.pool_op_kernel_sizes, self.conv_kernel_sizes = pool_op_kernel_sizes, conv_kernel_sizes
        self.conv_pad_sizes = [[i // 2 for i in k] for k in conv_kernel_sizes]
        num_pool = len(pool_op_kernel_sizes)
        assert num_pool == len(dims) - 1
        assert all(list(k) == [3, 3, 3] for k in conv_kernel_sizes) and all(list(p) == [2, 2, 2] for p in pool_op_kernel_sizes[:4]), \
            "HIP path implements the AnatoMask recipe: 3x3x3 kernels, stride-2 pooling"
        self.conv_blocks_context = nn.ModuleList()
        self.conv_blocks_context.append(nn.Sequential(
            BasicResBlock(input_channels, dims[0], 3, 1, use_1x1conv=True),
            *[BasicResBlock(dims[0], dims[0], 3, 1) for _ in range(depth[0] - 1)]))
        for d in range(1, num_pool):
            self.conv_blocks_context.append(nn.Sequential(
                BasicResBlock(dims[d - 1], dims[d], 3, 1, stride=2, use_1x1conv=True),
                *[BasicResBlock(dims[d], dims[d], 3, 1) for _ in range(depth[d] - 1)]))

    def get_downsample_ratio(self) -> int:
        return 16

    def get_feature_map_channels(self):
        return self.dims[:5]

    compute_dtype = torch.float32

    def _engine_ctx(self, want_grads=False):
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        size = tuple(self.__dict__["_input_size"])
        if owner is not None:                          # inside a SparK: its flat fp32 masters and packed MFMA copies
            owner._ensure_flat()
            spec = Spec(self.dims, self.depth, owner.spec.width, size)
            G = {n: torch.zeros_like(owner._W[n]) for n in self._param_names()} if want_grads else None
            return spec, owner._W, owner._pack, G
        sa = self.__dict__.setdefault("_sa", _Standalone(self, "sparse_encoder.sp_cnn.", self.compute_dtype))
        sa.dtype = self.compute_dtype
        return Spec(self.dims, self.depth, 16 * 8, size), sa.weights(), sa.pack, (sa.grads() if want_grads else None)

    def _param_names(self):
        return ["sparse_encoder.sp_cnn." + n for n, _ in self.named_parameters()]

    def forward(self, x, hierarchical=False):
        """P/STUNet_head.py:67-76 under the Sparse* layers of P/encoder3D.py: x (B,1,H,W,D) is the MASKED input, the patch mask
        comes from the side channel `modules._cur_active` (None: dense).  Returns the 5 stage maps (hierarchical) or the last."""
        if x.device.type != "cuda":
            raise RuntimeError("STUNet runs on the HIP engine: move the model and the input to a cuda (ROCm) device first")
        B = x.shape[0]
        size = tuple(x.shape[2:])
        self.__dict__["_input_size"] = size
        act = _cur_active
        if act is None:
            act = torch.ones(B, 1, *(v // 16 for v in size), dtype=torch.bool, device=x.device)
        if any(v % 16 for v in size) or tuple(act.shape) != (B, 1, *(v // 16 for v in size)):
            raise RuntimeError(f"input {tuple(x.shape)} does not match the patch mask {tuple(act.shape)} (16^3 patches)")
        mi = ops.MaskInfo.from_bool(act, x.device)
        params = [p for _, p in self.named_parameters()]
        feats = _EncoderFn.apply(self, x[:, 0].float().contiguous(), mi, *params)
        return list(feats) if hierarchical else feats[-1]


class SparseEncoder(nn.Module):
    """P/encoder3D.py:278-367.  The dense->sparse module rewrite of the reference is unnecessary here:
    sparsity is a property of the kernels (patch mask argument), not of the module types."""

    def __init__(self, cnn, input_size, sbn=False, verbose=False):
        super().__init__()
        self.sp_cnn = SparseEncoder.dense_model_to_sparse(cnn, verbose=verbose, sbn=sbn)
        self.input_size, self.downsample_ratio, self.enc_feat_map_chs = input_size, cnn.get_downsample_ratio(), cnn.get_feature_map_channels()

    @staticmethod
    def dense_model_to_sparse(m: nn.Module, verbose=False, sbn=False):
        """This package's STUNet needs no rewrite (its engine takes the patch mask as a kernel argument).  Any OTHER dense backbone
        (MedNeXt, ConvNeXt ...: plain torch.nn layers) is rewritten layer by layer as the reference does (P/encoder3D.py:300-364), onto
        the HIP sparse layers of anatomask_amd.sparse_layers; it then runs stand-alone (`forward`), outside the fused STUNet trainer."""
        if isinstance(m, STUNet):
            return m
        from .sparse_layers import dense_model_to_sparse
        return dense_model_to_sparse(m, verbose=verbose, sbn=sbn)

    def forward(self, x):
        """P/encoder3D.py:366-367: `self.sp_cnn(x, hierarchical=True)` -> the 5 feature maps (fine -> coarse)."""
        return self.sp_cnn(x, hierarchical=True)


# --------------------------------------------------------------------------- decoder
class UNetBlock(nn.Module):
    """ConvT(k4,s2,p1) -> conv3-BN-ReLU6-conv3-BN, P/decoder3D.py:13-29 (parameters only)."""

    def __init__(self, cin, cout, bn3d):
        super().__init__()
        self.up_sample = nn.ConvTranspose3d(cin, cin, kernel_size=4, stride=2, padding=1, bias=True)
        self.conv = nn.Sequential(
            nn.Conv3d(cin, cin, kernel_size=3, stride=1, padding=1, bias=False), bn3d(cin), nn.ReLU6(inplace=True),
            nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1, bias=False), bn3d(cout))


class LightDecoder(nn.Module):
    """P/decoder3D.py:32-84."""

    def __init__(self, up_sample_ratio, width=768, sbn=True, use_IN=False, out_channel=1):
        super().__init__()
        self.width = width
        assert up_sample_ratio > 0 and up_sample_ratio & (up_sample_ratio - 1) == 0
        # P/decoder3D.py:42-47: sbn takes precedence over use_IN.  use_IN (with sbn=False): nn.InstanceNorm3d -- no affine parameters, no
        # running statistics, per-sample statistics in train AND eval mode (engine._instance_norm); module API only, the fused trainer refuses it
        self.use_IN = bool(use_IN) and not sbn
        self.sbn = bool(sbn)              # nn.SyncBatchNorm in the reference (P/decoder3D.py:42-43; on in plain-SparK DDP, P/pretrain_DDP.py:225):
        # same parameters / buffers / state_dict keys as BatchNorm3d; the engine all-reduces the batch statistics when a process group is up
        n = round(math.log2(up_sample_ratio))
        channels = [self.width // 2 ** i for i in range(n + 1)]
        bn3d = nn.InstanceNorm3d if self.use_IN else nn.BatchNorm3d
        self.dec = nn.ModuleList([UNetBlock(cin, cout, bn3d) for cin, cout in zip(channels[:-1], channels[1:])])
        self.proj = nn.Conv3d(channels[-1], out_channel, kernel_size=1, stride=1, bias=True)
        self.initialize()

    def extra_repr(self) -> str:
        return f"width={self.width}"

    def initialize(self):
        """P/decoder3D.py:68-84 (note its elif order: Conv3d takes the trunc_normal branch, ConvTranspose3d kaiming)."""
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                _trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.ConvTranspose3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    compute_dtype = torch.float32

    def _engine_ctx(self, want_grads=False):
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        chans = [self.width // 2 ** i for i in range(len(self.dec) + 1)]
        if owner is not None:
            owner._ensure_flat()
            G = {n: torch.zeros_like(owner._W[n]) for n in self._param_names()} if want_grads else None
            return owner.spec, owner._W, owner._pack, G
        sa = self.__dict__.setdefault("_sa", _Standalone(self, "dense_decoder.", self.compute_dtype))
        sa.dtype = self.compute_dtype
        spec = Spec([8, 16, 32, 64, 128, 128], [1] * 6, self.width, (16, 16, 16), dec_inorm=self.use_IN)
        assert spec.dec_chs == chans
        return spec, sa.weights(), sa.pack, (sa.grads() if want_grads else None)

    def _param_names(self):
        return ["dense_decoder." + n for n, _ in self.named_parameters()]

    def forward(self, to_dec):
        """P/decoder3D.py:55-63.  to_dec: list of (B,C_i,D_i,H_i,W_i) maps, coarse -> fine (entries past the 4 blocks are ignored,
        as in the reference; None entries are not supported).  Returns (B,1,D,H,W) fp32."""
        n = len(self.dec)
        use = list(to_dec[:n])
        if len(use) < n or any(t is None for t in use):
            raise NotImplementedError("LightDecoder.forward needs one map per UNetBlock (the SparK path always provides them)")
        if use[0].device.type != "cuda":
            raise RuntimeError("LightDecoder runs on the HIP engine: move the model and the inputs to a cuda (ROCm) device first")
        params = [p for _, p in self.named_parameters()]
        return _DecoderFn.apply(self, n, *use, *params)


class SparseInstanceNorm(nn.Module):
    """Parameter container of the pooled sparse InstanceNorm, P/encoder3D.py:138-165 (densify norms: eps 1e-6)."""

    def __init__(self, num_features, eps=1e-6, sparse=True):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.eps, self.sparse = eps, sparse


# --------------------------------------------------------------------------- SparK
class _SparKFn(torch.autograd.Function):
    """rec = SparK(inp | mask); one autograd node around the hand-written HIP forward/backward tape."""

    @staticmethod
    def forward(ctx, model: "SparK", inp_b1, mask_info, *params):
        need_grad = any(ctx.needs_input_grad[3:])          # False under torch.no_grad() (teacher pass)
        tape = engine.Tape() if need_grad else None
        rec, dec0 = engine.forward(model.spec, model._W, model._pack, inp_b1, mask_info, model.training, tape, recompute=model.recompute,
                                   want_to_dec0=True)
        ctx.model, ctx.tape, ctx.inp, ctx.mask = model, tape, inp_b1, mask_info
        model._last_dec0 = dec0                             # (B,f,f,f,C) channels-last; read by forward(return_feat=True), carries no gradient
        return rec

    @staticmethod
    def backward(ctx, drec):
        m: SparK = ctx.model
        if ctx.tape is None:
            raise RuntimeError("backward through an eval / no-grad SparK forward")
        m._gflat.zero_()
        engine.backward(m.spec, m._W, m._G, m._pack, ctx.inp, ctx.mask, ctx.tape, drec.contiguous(), m._after_group)
        ctx.tape = None
        grads = [m._G[n] if n not in m._dead else None for n in m._pnames]
        return (None, None, None, *grads)


class _PatchLossFn(torch.autograd.Function):
    """forward_loss on patchified (B,L,4096) tensors with the fused HIP kernels."""

    @staticmethod
    def forward(ctx, inp_bln, rec_bln, active_u8_flat):
        B, L, N = inp_bln.shape
        assert N == 4096, "patch size 16^3, single channel"
        mi = ops.MaskInfo(active_u8_flat.view(B * L, 1, 1, 1))
        a, r = inp_bln.contiguous().view(B * L, 16, 16, 16), rec_bln.contiguous().view(B * L, 16, 16, 16)
        l2m, pm, pr, info = ops.patch_loss_fwd(a, r, mi, True)
        ctx.save_for_backward(a, r, pm, pr, info)
        ctx.mi, ctx.shape = mi, (B, L, N)
        ctx.mark_non_differentiable(l2m)
        return info[0], l2m.view(B, L)

    @staticmethod
    def backward(ctx, gloss, _g2):
        a, r, pm, pr, info = ctx.saved_tensors
        drec = ops.patch_loss_bwd(a, r, ctx.mi, pm, pr, info, gloss.reshape(1).float().contiguous())
        return None, drec.view(ctx.shape), None


class SparK(nn.Module):
    """P/AnatoMask.py:13-277: masked-image-modelling wrapper (mask, generate_mask, forward, forward_loss,
    patchify/unpatchify, get_config, state_dict(with_config), load_state_dict)."""

    def __init__(self, sparse_encoder: SparseEncoder, dense_decoder: LightDecoder, mask_ratio=0.6, densify_norm="in", sbn=False,
                 compute_dtype: torch.dtype = torch.float32, recompute: bool = False):
        super().__init__()
        self.recompute = recompute        # P/GC.py policy: recompute encoder stages / decoder blocks in backward
        input_size, downsample_ratio = sparse_encoder.input_size, sparse_encoder.downsample_ratio
        self.downsample_ratio = downsample_ratio
        self.fmap_h, self.fmap_w, self.fmap_d = (input_size[0] // downsample_ratio, input_size[1] // downsample_ratio,
                                                 input_size[2] // downsample_ratio)
        self.mask_ratio = mask_ratio
        self.len_keep = round(self.fmap_h * self.fmap_w * self.fmap_d * (1 - mask_ratio))
        self.sparse_encoder, self.dense_decoder, self.sbn = sparse_encoder, dense_decoder, sbn
        self.hierarchy = len(sparse_encoder.enc_feat_map_chs)
        self.densify_norm_str = densify_norm.lower()
        # densify_norm other than 'in' (P/AnatoMask.py:44-56: 'bn' / 'ln' / 'gn' / anything else = identity) runs through the sparse layer zoo
        # in the GENERIC composition (zoo-converted encoders); the fused STUNet engine / trainer are built for the drivers' 'in'
        # (P/pretrain_AntoMask.py:214-217)
        if self.densify_norm_str != "in" and isinstance(sparse_encoder.sp_cnn, STUNet):
            raise NotImplementedError("the fused STUNet path implements densify_norm='in' (what every driver uses); other densify norms run "
                                      "with zoo-converted encoders (SparseEncoder(<dense backbone>))")
        self.densify_norms, self.densify_projs, self.mask_tokens = nn.ModuleList(), nn.ModuleList(), nn.ParameterList()
        e_widths, d_width = list(sparse_encoder.enc_feat_map_chs), dense_decoder.width
        for i in range(self.hierarchy):
            e_width = e_widths.pop()
            p = nn.Parameter(torch.zeros(1, e_width, 1, 1, 1))
            _trunc_normal_(p, std=.02, a=-.02, b=.02)
            self.mask_tokens.append(p)
            if self.densify_norm_str == "in":
                self.densify_norms.append(SparseInstanceNorm(e_width, sparse=True))
            else:
                from . import sparse_layers as SL
                if self.densify_norm_str == "bn":
                    self.densify_norms.append((SL.SparseSyncBatchNorm3d if self.sbn else SL.SparseBatchNorm3d)(e_width))
                elif self.densify_norm_str == "ln":
                    self.densify_norms.append(SL.SparseConvNeXtLayerNorm(e_width, data_format="channels_first", sparse=True))
                elif self.densify_norm_str == "gn":
                    self.densify_norms.append(SL.SparseGroupNorm(e_width, e_width, sparse=True))
                else:
                    self.densify_norms.append(nn.Identity())
            if i == 0 and e_width == d_width:
                proj = nn.Identity()
            else:
                k = 1 if i <= 0 else 3
                proj = nn.Conv3d(e_width, d_width, kernel_size=k, stride=1, padding=k // 2, bias=True)
            self.densify_projs.append(proj)
            d_width //= 2
        cnn = sparse_encoder.sp_cnn
        self.compute_dtype = compute_dtype
        # fp32 storage only: matrix products from bf16 hi / lo splits (AM_DT_F32S) instead of the exact fp32 matrix instruction; a property
        # of THIS model (its PackCache marks every packed weight copy with it), see set_f32_split
        self.f32_split = bool(ops.DEFAULT_F32_SPLIT)
        # weight gradients as ordered per-slot folds instead of fp32 atomics (am_conv3d_wgrad's det_workspace); a property of THIS model too
        self.deterministic_wgrad = False
        self._flat: Optional[torch.Tensor] = None
        self._after_group = None
        # Any encoder other than this package's STUNet (a dense backbone rewritten by the sparse layer zoo, SparseEncoder.__init__) runs
        # the GENERIC composition of P/AnatoMask.py:137-188 -- zoo encoder -> densify (pooled norm, mask tokens, projections) ->
        # LightDecoder -- under torch autograd; the fused engine / trainer below are STUNet-only.
        self._generic = not isinstance(cnn, STUNet)
        if self._generic:
            self.dense_decoder.compute_dtype = compute_dtype
            self._dead = set()
            return
        self.spec = Spec(cnn.dims, cnn.depth, dense_decoder.width, tuple(input_size), sync_bn=bool(getattr(dense_decoder, "sbn", False)),
                         dec_inorm=bool(getattr(dense_decoder, "use_IN", False)))
        import weakref                       # sub-modules called on their own (model.sparse_encoder(x), model.dense_decoder(to_dec))
        cnn.__dict__["_owner"] = dense_decoder.__dict__["_owner"] = weakref.ref(self)   # run on THIS model's flat buffers
        n_dec = len(self.spec.dec_chs) - 1
        self._dead = {f"{k}.{i}{sfx}" for i in range(n_dec, self.hierarchy) for k, sfx in
                      (("densify_norms", ".weight"), ("densify_norms", ".bias"), ("densify_projs", ".weight"),
                       ("densify_projs", ".bias"), ("mask_tokens", ""))}

    # ---------------------------------------------------------------- flat storage
    def _ensure_flat(self):
        """(Re)bind every parameter to a view of ONE flat fp32 buffer [live | dead], gradients likewise,
        BN running stats to a second one: the optimizer / EMA / all-reduce kernels stream over these."""
        if self._generic:
            raise RuntimeError("the flat-buffer engine path (fused trainer, STUNet sub-module forwards) needs this package's STUNet encoder")
        named = list(self.named_parameters())
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("SparK runs on the HIP engine: move the model to a cuda (ROCm) device first")
        if self._flat is not None and self._flat.device == dev and all(
                p.data_ptr() == self._W[n].data_ptr() for n, p in named):
            return
        order = [(n, p) for n, p in named if n not in self._dead] + [(n, p) for n, p in named if n in self._dead]
        offs, tot = {}, 0
        for n, p in order:
            if n in self._dead and "live_end" not in offs:
                offs["live_end"] = tot
            offs[n] = tot
            tot += (p.numel() + 3) // 4 * 4
        offs.setdefault("live_end", tot)
        flat = torch.zeros(tot, device=dev, dtype=torch.float32)
        gflat = torch.zeros(tot, device=dev, dtype=torch.float32)
        self._W, self._G, self._pnames = {}, {}, [n for n, _ in named]
        for n, p in order:
            v = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
            v.copy_(p.data.float())
            p.data = v
            p.grad = None
            self._W[n] = v
            self._G[n] = gflat[offs[n]:offs[n] + p.numel()].view(p.shape)
        bufs = [(n, b) for n, b in self.named_buffers() if b.is_floating_point()]
        btot = sum((b.numel() + 3) // 4 * 4 for _, b in bufs)
        bflat = torch.zeros(max(btot, 4), device=dev, dtype=torch.float32)
        o = 0
        for n, b in bufs:
            v = bflat[o:o + b.numel()].view(b.shape)
            v.copy_(b.data.float())
            b.data = v
            self._W[n] = v
            o += (b.numel() + 3) // 4 * 4
        # integer buffers (BatchNorm num_batches_tracked, int64): views of ONE flat int64 buffer, so the teacher's EMA over them
        # (am_ema_i64) and the non-finite guard's snapshot / restore are one launch each
        ibufs = [(n, b) for n, b in self.named_buffers() if not b.is_floating_point()]
        assert all(b.dtype == torch.int64 for _, b in ibufs)
        iflat = torch.zeros(max(sum(b.numel() for _, b in ibufs), 1), device=dev, dtype=torch.int64)
        o = 0
        for n, b in ibufs:
            v = iflat[o:o + b.numel()].view(b.shape)
            v.copy_(b.data)
            b.data = v
            self._W[n] = v
            o += b.numel()
        self._iflat, self._n_ibuf = iflat, o
        self._flat, self._gflat, self._bflat, self._live_end, self._offs = flat, gflat, bflat, offs["live_end"], offs
        self._pack = engine.PackCache(self.compute_dtype, self.f32_split, self.deterministic_wgrad)

    def set_deterministic_wgrad(self, flag: bool):
        """True: the convolution / stem / projection weight gradients of THIS model's backward are per-slot partial sums folded in a fixed
        order (bit-reproducible, +3 % step time); False: fp32 atomics.  No process-wide switch."""
        self.deterministic_wgrad = bool(flag)
        if self._flat is not None and not self._generic:
            self._pack.deterministic = self.deterministic_wgrad
        return self

    def set_f32_split(self, flag: bool):
        """fp32-storage model: True = products from bf16 hi / lo splits of both operands with fp32 accumulation (16 significant bits per
        operand, 4x the matrix rate), False = the exact fp32 matrix instruction (parity mode).  The packed weight copies are re-made."""
        self.f32_split = bool(flag)
        if self._flat is not None and not self._generic:
            self._pack = engine.PackCache(self.compute_dtype, self.f32_split, self.deterministic_wgrad)
        return self

    def _apply(self, fn, *a, **k):               # .to()/.cuda() replace parameter storage: re-flatten lazily
        self._flat = None
        return super()._apply(fn, *a, **k)

    def weights_changed(self):
        """Tell the model its fp32 master weights were modified (optimizer step / load): drops packed copies."""
        if self._generic:
            sa = self.dense_decoder.__dict__.get("_sa")
            if sa is not None:
                sa.pack.invalidate()
            return
        if self._flat is not None:
            self._pack.invalidate()

    def _forward_generic(self, inp_bchwd: torch.Tensor, active_b1ff: torch.Tensor):
        """P/AnatoMask.py:144-170 for a zoo-converted encoder: rec volume (B,1,H,W,D), autograd through the zoo's Functions, the pooled
        norm of the engine (am_chan_stats / am_norm_*), the matrix-core convs and LightDecoder's engine node."""
        from . import sparse_layers as SL
        p = self.downsample_ratio
        act = active_b1ff.to(inp_bchwd.device)
        act_ex = act.repeat_interleave(p, 2).repeat_interleave(p, 3).repeat_interleave(p, 4)
        SL.COMPUTE_DTYPE = self.compute_dtype
        fea = list(self.sparse_encoder(inp_bchwd * act_ex))          # fine -> coarse
        fea.reverse()
        to_dec = []
        for i, f in enumerate(fea):
            if f is not None and i < len(self.dense_decoder.dec):    # (levels past the decoder's blocks are never read, :170 + decoder3D.py:56-60)
                n_ = self.densify_norms[i]
                if self.densify_norm_str == "in":
                    # pooled sparse InstanceNorm + mask-token fill in one HIP apply pass (and one backward pair), as the fused engine does
                    f = SL._nc(SL._DensifyFn.apply(SL._cl(f), n_.weight, n_.bias, self.mask_tokens[i], n_.eps))
                else:
                    # P/AnatoMask.py:158-163: the zoo norm (zeros at inactive positions), then the mask token at the inactive positions of
                    # THIS level's map (the activity mask up-sampled to the map: SL.fill_tokens, one HIP pass forward, one backward)
                    f = SL.fill_tokens(n_(f), self.mask_tokens[i])
                pr = self.densify_projs[i]
                if not isinstance(pr, nn.Identity):
                    f = SL.dense_conv(f, pr.weight, pr.bias, pr.kernel_size[0])
                to_dec.append(f)
        return self.dense_decoder(to_dec)

    # ---------------------------------------------------------------- masks
    def mask(self, B: int, device, generator=None):
        """P/AnatoMask.py:75-79 (CPU rand -> argsort, as the reference)."""
        h, w, d = self.fmap_h, self.fmap_w, self.fmap_d
        idx = torch.rand(B, h * w * d, generator=generator).argsort(dim=1)[:, :self.len_keep].to(device)
        return torch.zeros(B, h * w * d, dtype=torch.bool, device=device).scatter_(dim=1, index=idx, value=True).view(B, 1, h, w, d)

    @staticmethod
    def len_loss_for(L, len_keep, epoch, total_epoch, guide=True):
        keep_ratio = float((epoch + 1) / total_epoch) * 0.5 if guide else 2 / 3   # P/AnatoMask.py:88-93
        return max(int((L - len_keep) * keep_ratio), 0)

    @torch.no_grad()
    def generate_mask(self, loss_pred, guide=True, epoch=0, total_epoch=200, generator=None, original_mask=None, keys=None):
        """P/AnatoMask.py:81-135, on device in one kernel (no .cpu().numpy() round trips).  Returns
        (mask, easy_mask); easy_mask (not consumed by any caller of the reference) is what P/AnatoMask.py:116-134 describes, see _easy_mask.
        `keys` (B,L) teacher-forces the random permutation (keys[id] = position); default: device rand."""
        B, L = loss_pred.shape
        ll = self.len_loss_for(L, self.len_keep, epoch, total_epoch, guide)
        if keys is None:
            keys = torch.rand(B, L, device=loss_pred.device, generator=generator)
        m = ops.mask_sampler(loss_pred.float().contiguous(), keys.float().contiguous().to(loss_pred.device), self.len_keep, ll)
        mask = m.bool().view(B, 1, self.fmap_h, self.fmap_w, self.fmap_d)
        return mask, self._easy_mask(loss_pred, mask, ll)

    def _easy_mask(self, loss_pred, mask, len_loss):
        """The second output of P/AnatoMask.py:116-134.  What its construction states: the `easy_len = L - len_keep - len_loss` patches
        whose teacher loss ranks just BELOW the hard band (ascending ranks [len_keep + ... ) i.e. [L - len_loss - easy_len, L - len_loss))
        are hidden, the other len_keep + len_loss patches are visible; in the random regime (len_loss == 0) it is the mask itself.
        No driver reads it.  The reference's code for it is not reproducible bit for bit and is not reproduced: it
        re-creates `ids_shuffle2` inside the per-sample loop (only the LAST sample's row survives), its second slice assignment overwrites
        part of the first, and it shuffles with the unseeded global numpy generator -- this is the per-sample result those lines describe,
        which involves no random draw (every patch outside the band is visible).  Host-side bookkeeping on (B, L) booleans."""
        B, L = loss_pred.shape
        if len_loss <= 0:
            return mask.clone()
        easy_len = (L - self.len_keep) - len_loss
        ranks = torch.argsort(torch.argsort(loss_pred.float(), dim=1, stable=True), dim=1, stable=True)     # ascending loss rank of every patch
        hidden = (ranks >= L - len_loss - easy_len) & (ranks < L - len_loss)
        return (~hidden).view(B, 1, self.fmap_h, self.fmap_w, self.fmap_d)

    # ---------------------------------------------------------------- forward
    def reconstruct(self, inp_bchwd: torch.Tensor, active_b1ff: torch.Tensor) -> torch.Tensor:
        """(B,1,H,W,D) -> reconstruction volume (B,1,H,W,D) through the HIP engine (autograd-aware)."""
        self._ensure_flat()
        assert inp_bchwd.shape[1] == 1 and tuple(inp_bchwd.shape[2:]) == tuple(self.spec.input_size), inp_bchwd.shape
        mi = ops.MaskInfo.from_bool(active_b1ff, inp_bchwd.device)
        x = inp_bchwd[:, 0].float().contiguous()
        params = [p for _, p in self.named_parameters()]
        return _SparKFn.apply(self, x, mi, *params).unsqueeze(1)

    def forward(self, inp_bchwd: torch.Tensor, active_b1ff=None, vis=False, return_feat=False):
        if active_b1ff is None:
            active_b1ff = self.mask(inp_bchwd.shape[0], inp_bchwd.device)
        global _cur_active
        _cur_active = active_b1ff                                  # P/AnatoMask.py:143 (kept for code that reads the side channel)
        if self._generic:
            if return_feat:
                raise NotImplementedError("return_feat with a non-STUNet encoder")
            rec_bchwd = self._forward_generic(inp_bchwd, active_b1ff)
        else:
            rec_bchwd = self.reconstruct(inp_bchwd, active_b1ff)
        inp, rec = self.patchify(inp_bchwd), self.patchify(rec_bchwd)
        if return_feat:                                            # P/AnatoMask.py:172-173: to_dec[0].flatten(2).permute(0, 2, 1)
            d0 = self._last_dec0                                   # (no caller of the reference differentiates through it)
            return inp, rec, d0.reshape(d0.shape[0], -1, d0.shape[-1]).float()
        if vis:                                                    # P/AnatoMask.py:179-185
            p = self.downsample_ratio
            act = active_b1ff.repeat_interleave(p, 2).repeat_interleave(p, 3).repeat_interleave(p, 4)
            mean = inp.mean(dim=-1, keepdim=True)
            var = (inp.var(dim=-1, keepdim=True) + 1e-6) ** .5
            rec_v = self.unpatchify(rec * var + mean)
            return inp_bchwd, inp_bchwd * act, torch.where(act, inp_bchwd, rec_v)
        return inp, rec

    def forward_loss(self, inp, rec, active_b1ff):
        """P/AnatoMask.py:190-202 -> (scalar loss, per-patch masked l2 (B,L))."""
        a = active_b1ff.reshape(-1).to(device=inp.device, dtype=torch.uint8).contiguous()
        return _PatchLossFn.apply(inp.float(), rec.float(), a)

    def forward_learning_loss(self, loss_pred, loss_target):
        """P/AnatoMask.py:204-219: MSE between a predicted per-patch loss and the per-image normalised target (N, L) -- the objective of a
        loss-prediction head that no module of the reference builds and no driver calls (`loss_decoder` of P/pretrain_AnatoMask_DDP.py:230-233
        is not a parameter of SparK.__init__).  Same expression, same broadcasting, on whatever device the (tiny) tensors live; plain
        torch arithmetic on B x L scalars, differentiable."""
        mean = loss_target.mean(dim=1, keepdim=True)
        var = loss_target.var(dim=1, keepdim=True)
        loss_target = (loss_target - mean) / (var + 1.e-6) ** .5
        return ((loss_pred - loss_target) ** 2).mean()

    def patchify(self, bchwd):
        p = self.downsample_ratio
        h, w, d = self.fmap_h, self.fmap_w, self.fmap_d
        B, C = bchwd.shape[:2]
        bchwd = bchwd.reshape(shape=(B, C, h, p, w, p, d, p))
        bchwd = torch.einsum("bchpwqdg->bhwdpqgc", bchwd)
        return bchwd.reshape(shape=(B, h * w * d, C * p ** 3))

    def unpatchify(self, bln):
        p = self.downsample_ratio
        h, w, d = self.fmap_h, self.fmap_w, self.fmap_d
        B, C = bln.shape[0], bln.shape[-1] // p ** 3
        bln = bln.reshape(shape=(B, h, w, d, p, p, p, C))
        bln = torch.einsum("bhwdpqgc->bchpwqdg", bln)
        return bln.reshape(shape=(B, C, h * p, w * p, d * p))

    # ---------------------------------------------------------------- config / state
    def get_config(self):
        return {"mask_ratio": self.mask_ratio, "densify_norm_str": self.densify_norm_str, "sbn": self.sbn,
                "hierarchy": self.hierarchy, "sparse_encoder.input_size": self.sparse_encoder.input_size,
                "dense_decoder.width": self.dense_decoder.width}

    def __repr__(self):
        return f"\n[SparK.config]: {pformat(self.get_config(), indent=2, width=250)}\n[SparK.structure]: {super().__repr__()}"

    def state_dict(self, destination=None, prefix="", keep_vars=False, with_config=False):
        state = super().state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)
        if with_config:
            state["config"] = self.get_config()
        return state

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        config = state_dict.pop("config", None)
        res = super().load_state_dict(state_dict, strict=strict)
        self.weights_changed()
        if config is not None:
            for k, v in self.get_config().items():
                if config.get(k, None) != v:
                    err = f"[SparseMIM.load_state_dict] config mismatch:  this.{k}={v} (ckpt.{k}={config.get(k, None)})"
                    if strict:
                        raise AttributeError(err)
                    print(err, file=sys.stderr)
        return res

    def __deepcopy__(self, memo):
        """deepcopy (ModelEma) must not share the flat buffers: copy the module tree with plain tensors."""
        flat, self._flat = self._flat, None
        saved = {k: self.__dict__.pop(k) for k in ("_W", "_G", "_gflat", "_bflat", "_iflat", "_pack", "_offs") if k in self.__dict__}
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        for p_new, p_old in zip(new.parameters(), self.parameters()):
            p_new.data = p_old.data.clone()
        for b_new, b_old in zip(new.buffers(), self.buffers()):
            b_new.data = b_old.data.clone()
        self._flat = flat
        self.__dict__.update(saved)
        import weakref                       # the copy's sub-modules belong to the copy
        if not self._generic:                # (a generic composition has no flat engine buffers: its sub-modules stay stand-alone)
            new.sparse_encoder.sp_cnn.__dict__["_owner"] = new.dense_decoder.__dict__["_owner"] = weakref.ref(new)
        for mod in (new.sparse_encoder.sp_cnn, new.dense_decoder):
            mod.__dict__.pop("_sa", None)
        return new


# --------------------------------------------------------------------------- EMA teacher
class ModelEma:
    """timm.utils.ModelEma restated (deepcopy -> eval -> requires_grad False; update over EVERY state_dict
    entry, mapping the 'module.' prefix).  When both sides are SparK models the update is one fused HIP
    pass over the flat buffers."""

    def __init__(self, model, decay=0.9999, device="", resume=""):
        self.ema = copy.deepcopy(model)
        self.ema.eval()
        self.decay = decay
        self.device = device
        if device:
            self.ema.to(device=device)
        self.ema_has_module = hasattr(self.ema, "module")
        if resume:
            sd = torch.load(resume, map_location="cpu")
            self.ema.load_state_dict(sd.get("state_dict_ema", sd))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def update(self, model):
        src = model.module if hasattr(model, "module") and not self.ema_has_module else model
        if isinstance(src, SparK) and isinstance(self.ema, SparK) and not src._generic:
            src._ensure_flat(); self.ema._ensure_flat()
            ops.ema(self.ema._flat, src._flat, self.decay)
            ops.ema(self.ema._bflat, src._bflat, self.decay)
            if src._n_ibuf:                                       # int64 num_batches_tracked, as timm does it (promotion to float32, truncation)
                ops.ema_i64(self.ema._iflat[:src._n_ibuf], src._iflat[:src._n_ibuf], self.decay)
            self.ema.weights_changed()
            return
        msd = src.state_dict()
        for k, ema_v in self.ema.state_dict().items():
            ema_v.copy_(ema_v * self.decay + (1. - self.decay) * msd[k].detach().to(ema_v.device))
        if isinstance(self.ema, SparK):
            self.ema.weights_changed()


class LocalDDP(nn.Module):
    """P/pretrain_AntoMask.py:201-207."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


# --------------------------------------------------------------------------- optimizer-side host helpers
def get_param_groups(model, nowd_keys=()):
    """P/utils/lr_control.py:32-53: two groups carrying weight_decay_scale / lr_scale keys (never applied by
    the drivers, so both end up with the optimizer's weight_decay)."""
    groups: Dict[str, dict] = {}
    for name, para in model.named_parameters():
        if not para.requires_grad:
            continue
        if len(para.shape) == 1 or name.endswith(".bias") or any(k in name for k in nowd_keys):
            wd_scale, gname = 0., "no_decay"
        else:
            wd_scale, gname = 1., "decay"
        groups.setdefault(gname, {"params": [], "weight_decay_scale": wd_scale, "lr_scale": 1.})["params"].append(para)
    return list(groups.values())


def linear_warmup_cosine_lrs(epochs: int, base_lr: float = 1e-4, warmup: int = 20, warmup_start_lr: float = 1e-6,
                             eta_min: float = 0.0) -> List[float]:
    """nnunetv2/training/lr_scheduler/LinearWarmupCosine.py:65-100 in its chained form, one value per epoch."""
    lrs, lr = [warmup_start_lr], warmup_start_lr
    for e in range(1, epochs + 1):
        if e < warmup:
            lr = lr + (base_lr - warmup_start_lr) / (warmup - 1)
        elif e == warmup:
            lr = base_lr
        elif (e - 1 - epochs) % (2 * (epochs - warmup)) == 0:
            lr = lr + (base_lr - eta_min) * (1 - math.cos(math.pi / (epochs - warmup))) / 2
        else:
            lr = ((1 + math.cos(math.pi * (e - warmup) / (epochs - warmup)))
                  / (1 + math.cos(math.pi * (e - warmup - 1) / (epochs - warmup))) * (lr - eta_min) + eta_min)
        lrs.append(lr)
    return lrs


def ema_decay_for_epoch(i: int, total_epochs: int) -> float:
    """P/pretrain_AntoMask.py:383-386."""
    q = total_epochs // 4
    return 0.999 + i / q * (0.9999 - 0.999) if i < q else 0.9999


def build_spark(dims, depth, width, input_size, mask_ratio=0.6, compute_dtype=torch.float32, recompute=False, sbn=False) -> SparK:
    """The model build of P/pretrain_AntoMask.py:184-217 in one call (sbn=True: the decoder of P/pretrain_DDP.py:225)."""
    head = STUNet(1, 1, depth=list(depth), dims=list(dims))
    enc = SparseEncoder(head, input_size=tuple(input_size), sbn=False)
    dec = LightDecoder(enc.downsample_ratio, sbn=sbn, width=width, out_channel=1)
    return SparK(sparse_encoder=enc, dense_decoder=dec, mask_ratio=mask_ratio, densify_norm="in", compute_dtype=compute_dtype,
                 recompute=recompute)


STUNET_CONFIGS = {   # P/pretrain_AntoMask.py:188-196, P/pretrain_AnatoMask_DDP.py:223-229
    "S": dict(dims=[16, 32, 64, 128, 256, 256], depth=[1] * 6, width=256),
    "B": dict(dims=[32, 64, 128, 256, 512, 512], depth=[1] * 6, width=512),
    "L": dict(dims=[64 * x for x in [1, 2, 4, 8, 16, 16]], depth=[2] * 6, width=1024),
    "H": dict(dims=[96 * x for x in [1, 2, 4, 8, 16, 16]], depth=[3] * 6, width=1536),
}
