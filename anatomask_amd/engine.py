"""Forward / backward orchestration of SparK(STUNet) on the HIP kernels.

This is the host side of the hot path: a hand-written tape (no torch.autograd inside) that strings
the C-ABI kernels together in the order of the reference's forward
(P/AnatoMask.py:137-176 -> P/encoder3D.py:366 -> P/STUNet_head.py:67-103 -> P/AnatoMask.py:158-168
-> P/decoder3D.py:55-63) and of its autograd backward (loss.backward(), P/pretrain_AntoMask.py:435).

Data layout in HBM: every activation is channels-last [B,D,H,W,C] in the compute dtype (bf16 or
fp32).  Encoder tensors are block-sparse: only voxels of active 16^3 patches are ever written or
read (kernels consult the uint8 patch mask; nothing relies on the content of inactive voxels).
Parameters / gradients are torch-layout fp32 tensors (views into flat buffers owned by SparK).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU6, CONV_DGRAD, CONV_FWD, CONVT_DGRAD, CONVT_FWD, MaskInfo, NormStats

ENC = "sparse_encoder.sp_cnn.conv_blocks_context"
DEC = "dense_decoder.dec"


@dataclass
class Spec:
    """Static model description (what P/pretrain_AntoMask.py:184-215 hard-codes)."""
    dims: Sequence[int]
    depth: Sequence[int]
    width: int
    input_size: Tuple[int, int, int]
    in_ch: int = 1
    out_ch: int = 1
    downsample: int = 16
    n_stage: int = 5
    sync_bn: bool = False        # decoder BatchNorm statistics over ALL ranks (nn.SyncBatchNorm, P/decoder3D.py:42-43)
    dec_inorm: bool = False      # LightDecoder(sbn=False, use_IN=True): nn.InstanceNorm3d (no affine, no running statistics, P/decoder3D.py:44-45)
    enc_chs: List[int] = field(init=False)
    dec_chs: List[int] = field(init=False)
    fmap: Tuple[int, int, int] = field(init=False)

    def __post_init__(self):
        assert self.in_ch == 1 and self.out_ch == 1, "the AnatoMask recipe is single-channel CT (P/pretrain_AntoMask.py:188)"
        assert all(s % self.downsample == 0 for s in self.input_size)
        self.enc_chs = list(self.dims[:5])
        n = 4
        self.dec_chs = [self.width // 2 ** i for i in range(n + 1)]
        self.fmap = tuple(s // self.downsample for s in self.input_size)
        assert all(c % 8 == 0 for c in self.enc_chs + self.dec_chs), "channel counts must be multiples of 8"

    def stage_spatial(self, s: int) -> Tuple[int, int, int]:
        return tuple(v >> s for v in self.input_size)


class Tape:
    """What the student forward keeps for its backward."""

    def __init__(self):
        self.enc: List[dict] = []          # one dict per BasicResBlock, forward order
        self.dens: List[dict] = []         # one per densify level (coarse -> fine)
        self.dec: List[dict] = []          # one per UNetBlock
        self.last: Optional[torch.Tensor] = None
        self.counts: Dict[int, torch.Tensor] = {}
        self.recompute = False
        self.enc_in: List[Optional[torch.Tensor]] = []   # recompute mode: the input map of every encoder stage


def _counts(mask: MaskInfo, levels: Sequence[int]) -> Dict[int, object]:
    """active-voxel counts per block shift: host numbers when the number of active patches is known (the trainer's sampler
    guarantees B * len_keep), else device doubles (one tiny kernel per level, no host synchronisation)."""
    if mask.n_active is not None:
        return {bs: float(mask.n_active) * float((1 << bs) ** 3) for bs in levels}
    out = {}
    buf = torch.empty(len(levels), device=mask.t.device, dtype=torch.float64)
    for i, bs in enumerate(levels):
        ops.mask_count(mask, (1 << bs) ** 3, buf[i:i + 1])
        out[bs] = buf[i:i + 1]
    return out


def _set_count(st: NormStats, cnt):
    if isinstance(cnt, float):
        st.count_host, st.count_ptr = cnt, None
    else:
        st.count_ptr = cnt


def _sparse_norm(x, mask, bs, counts, gamma, beta, eps, part=None) -> NormStats:
    """pooled sparse InstanceNorm statistics; `part` = partial sums left by the producing conv's epilogue."""
    st = NormStats(x.shape[-1], x.device)
    _set_count(st, counts[bs])
    if part is not None:
        part.finalize(st, gamma, beta, eps)                       # reduce + finalize in one launch
        return st
    ops.chan_stats(x, mask, bs, st)
    ops.norm_finalize(st, gamma, beta, eps)
    return st


def _sync_world(sync: bool) -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (sync and dist.is_available() and dist.is_initialized()) else 1


def _batch_norm(x, W, prefix, train: bool, part=None, update_running: bool = True, sync: bool = False) -> NormStats:
    st = NormStats(x.shape[-1], x.device)
    world = _sync_world(sync and train)
    if train and world > 1:
        # SyncBatchNorm: per-channel (sum, sum of squares) of the LOCAL batch -> all-reduce -> statistics of the global batch
        # (every rank holds the same number of voxels: equal per-GPU batches), running stats from the global unbiased variance
        import torch.distributed as dist
        st.count_host = float(x.numel() // x.shape[-1]) * world
        if part is not None:
            part.reduce(sums=st.sums); st.nrep = 1
        else:
            ops.chan_stats(x, None, 0, st)
        red = st.sums[:st.nrep]
        dist.all_reduce(red)
        st.sync_world = world
        if update_running:
            ops.norm_finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5, W[f"{prefix}.running_mean"], W[f"{prefix}.running_var"], 0.1)
            W[f"{prefix}.num_batches_tracked"].add_(1)
        else:
            ops.norm_finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5)
        return st
    if train:
        st.count_host = float(x.numel() // x.shape[-1])
        if part is not None:                                      # reduce + finalize (+ running stats, num_batches_tracked) in one launch
            if update_running:
                part.finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5, W[f"{prefix}.running_mean"], W[f"{prefix}.running_var"],
                              0.1, W[f"{prefix}.num_batches_tracked"])
            else:
                part.finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5)
            return st
        ops.chan_stats(x, None, 0, st)
        if update_running:
            ops.norm_finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5, W[f"{prefix}.running_mean"],
                              W[f"{prefix}.running_var"], 0.1)
            W[f"{prefix}.num_batches_tracked"].add_(1)
        else:                                     # recomputation pass: same batch statistics, running stats already updated
            ops.norm_finalize(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], 1e-5)
    else:
        ops.norm_fold_running(st, W[f"{prefix}.weight"], W[f"{prefix}.bias"], W[f"{prefix}.running_mean"],
                              W[f"{prefix}.running_var"], 1e-5)
    return st


_UNIT: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}


def _unit_affine(C: int, dev) -> Tuple[torch.Tensor, torch.Tensor]:
    k = (C, str(dev))
    if k not in _UNIT:
        _UNIT[k] = (torch.ones(C, device=dev), torch.zeros(C, device=dev))
    return _UNIT[k]


def _instance_norm(x, act: int, res=None) -> Tuple[torch.Tensor, List[NormStats]]:
    """nn.InstanceNorm3d (affine=False, track_running_stats=False, eps 1e-5; the same in train and eval mode) + activation (+ residual):
    the BatchNorm kernels on one sample at a time (a sample of a channels-last batch is a contiguous [1][D][H][W][C] view)."""
    one, zero = _unit_affine(x.shape[-1], x.device)
    out, sts = torch.empty_like(x), []
    for b in range(x.shape[0]):
        st = NormStats(x.shape[-1], x.device)
        st.count_host = float(x[b].numel() // x.shape[-1])
        ops.chan_stats(x[b:b + 1], None, 0, st)
        ops.norm_finalize(st, one, zero, 1e-5)
        ops.norm_apply(x[b:b + 1], st, act, res=None if res is None else res[b:b + 1], out=out[b:b + 1])
        sts.append(st)
    return out, sts


def _instance_norm_backward(g, x, sts: List[NormStats], act: int) -> torch.Tensor:
    one, _ = _unit_affine(x.shape[-1], x.device)
    dx = torch.empty_like(x)
    for b, st in enumerate(sts):
        ops.norm_backward(g[b:b + 1], None, x[b:b + 1], st, one, act, None, 0, None, None, dx=dx[b:b + 1])
    return dx


class PackCache:
    """Compute-dtype MFMA-layout copies of the conv weights, rebuilt when the fp32 master changes.
    The first step packs each weight when it is first used; from then on the packed copies persist and ONE batched launch
    (`am_pack_weights_batched`) rebuilds all of them after every optimizer / EMA update (84 launches -> 1)."""

    def __init__(self, dtype, f32_split: bool = False, deterministic: bool = False):
        self.dtype = dtype
        self.deterministic = bool(deterministic)      # weight gradients as ordered per-slot folds instead of fp32 atomics (SparK.deterministic_wgrad)
        self.f32_split = bool(dtype == torch.float32 and f32_split)      # fp32 storage: products from bf16 hi / lo splits (ops.py)
        self.store: Dict[Tuple[str, bool], torch.Tensor] = {}
        self.src: Dict[Tuple[str, bool], torch.Tensor] = {}
        self.dirty = False
        self.table = None

    def invalidate(self):
        self.dirty = bool(self.store)

    def _refresh(self):
        pairs = [(self.src[k], self.store[k]) for k in self.store]
        if self.table is None or not self.table.matches(pairs):
            self.table = ops.PackTable(pairs, self.dtype, pairs[0][1].device, self.f32_split)
        self.table.repack()
        self.dirty = False

    def get(self, W, name: str, transposed: bool, dgrad: bool) -> torch.Tensor:
        key = (name, dgrad)
        t = self.store.get(key)
        if t is not None and ops._is_split(t) != self.f32_split:
            self.store.clear(); self.src.clear(); self.table = None; self.dirty = False      # (the fp32 product mode changed: other layout)
            t = None
        if t is not None and self.src[key].data_ptr() != W[name].data_ptr():    # the master moved (new flat buffer): start over
            self.store.clear(); self.src.clear(); self.table = None; self.dirty = False
            t = None
        if t is None:
            if self.dirty:
                self._refresh()
            t = ops.pack_weight(W[name], self.dtype, transposed, dgrad, self.f32_split)
            self.store[key] = t
            self.src[key] = W[name]
            self.table = None
            return t
        if self.dirty:
            self._refresh()
        return t


# ====================================================================================== forward
FUSED_PRENORM = True    # tools/step_ab.py engine.FUSED_PRENORM=1,0: passes without a tape fold norm1 + LeakyReLU into conv2's source staging


def _enc_block(W, pk, inp, mask, counts, sp, s: int, b: int, x, keep: bool = True):
    """One BasicResBlock (P/STUNet_head.py:96-103) -> (out, record for backward).
    keep=False (no tape: the EMA teacher, validation, stand-alone forwards): where the kernel exists (the thin level-0 layers, conv_rw.hip)
    a1 = LReLU(norm1(y1)) is never written -- conv2 applies norm1 + LeakyReLU to y1 while it stages its source rows."""
    dt = pk.dtype
    bs = 4 - s
    p = f"{ENC}.{s}.{b}"
    first = b == 0
    stride = 2 if (first and s > 0) else 1
    rec_ = {"p": p, "s": s, "b": b, "first": first, "stride": stride, "x": x}
    if s == 0 and first:
        y1, pt1 = ops.stem_conv_fwd(inp, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"], mask, bs, dt, want_partials=True)
    else:
        y1, pt1 = ops.conv3d(CONV_FWD, x, pk.get(W, f"{p}.conv1.weight", False, False), W[f"{p}.conv1.bias"], sp, 3, stride,
                             in_mask=mask, in_bshift=bs + (1 if stride == 2 else 0), out_mask=mask, out_bshift=bs,
                             want_partials=True)
    st1 = _sparse_norm(y1, mask, bs, counts, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], 1e-5, pt1)
    w2 = pk.get(W, f"{p}.conv2.weight", False, False)
    if not keep and FUSED_PRENORM and ops.conv3d_prenorm_supported(y1, w2, sp, 3, 1, mask, bs):
        a1 = None
        y2, pt2 = ops.conv3d_prenorm(y1, st1, ACT_LRELU, w2, W[f"{p}.conv2.bias"], sp, 3, 1, mask, bs, want_partials=True)
    else:
        a1 = ops.norm_apply(y1, st1, ACT_LRELU, mask, bs)
        y2, pt2 = ops.conv3d(CONV_FWD, a1, w2, W[f"{p}.conv2.bias"], sp, 3, 1,
                             in_mask=mask, in_bshift=bs, out_mask=mask, out_bshift=bs, want_partials=True)
    st2 = _sparse_norm(y2, mask, bs, counts, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], 1e-5, pt2)
    if s == 0 and first:      # 1x1 Cin=1 shortcut folded into the apply pass
        out = ops.norm_apply(y2, st2, ACT_LRELU, mask, bs, stem=(inp, W[f"{p}.conv3.weight"].view(-1), W[f"{p}.conv3.bias"]))
    elif first:
        sc = ops.conv3d(CONV_FWD, x, pk.get(W, f"{p}.conv3.weight", False, False), W[f"{p}.conv3.bias"], sp, 1, stride,
                        in_mask=mask, in_bshift=bs + (1 if stride == 2 else 0), out_mask=mask, out_bshift=bs)
        out = ops.norm_apply(y2, st2, ACT_LRELU, mask, bs, res=sc)
    else:
        out = ops.norm_apply(y2, st2, ACT_LRELU, mask, bs, res=x)
    rec_.update(y1=y1, st1=st1, a1=a1, y2=y2, st2=st2, out=out)
    return out, rec_


def _dec_block(W, pk, i: int, x, nxt, train: bool, update_running: bool = True, fuse_eval: bool = False, out_skip=None, sync: bool = False,
               head: bool = False, inorm: bool = False, stencil_tail: bool = False):
    """One UNetBlock (P/decoder3D.py:13-29) (+ the `x + to_dec[i+1]` of the next iteration, :59) -> (out, record).
    head (last block, train mode): the block's final BatchNorm feeds only the 1x1 projection, which applies it on the fly
    (ops.proj_fwd(pre=st2)) -- the block returns the BatchNorm's INPUT c2 and the normalised map is never written."""
    q = f"{DEC}.{i}"
    so = tuple(2 * v for v in x.shape[1:4])
    u = ops.conv3d(CONVT_FWD, x, pk.get(W, f"{q}.up_sample.weight", True, False), W[f"{q}.up_sample.bias"], so, 4, 2)
    if inorm:
        assert not head
        c1 = ops.conv3d(CONV_FWD, u, pk.get(W, f"{q}.conv.0.weight", False, False), None, so, 3, 1)
        r, st1 = _instance_norm(c1, ACT_RELU6)
        c2 = ops.conv3d(CONV_FWD, r, pk.get(W, f"{q}.conv.3.weight", False, False), None, so, 3, 1)
        o, st2 = _instance_norm(c2, ACT_NONE, res=nxt)
        return o, {"q": q, "xin": x, "u": u, "c1": c1, "st1": st1, "r": r, "c2": c2, "st2": st2, "inorm": True}
    if not train and fuse_eval:
        # eval-mode BatchNorm (running statistics: the EMA teacher) is a per-channel affine map: it is folded, with the ReLU6
        # and the skip add, into the store of the convolution that feeds it -- no separate pass over the 128^3 tensors
        C1, C2 = W[f"{q}.conv.0.weight"].shape[0], W[f"{q}.conv.3.weight"].shape[0]
        st1, st2 = NormStats(C1, x.device), NormStats(C2, x.device)
        ops.norm_fold_running(st1, W[f"{q}.conv.1.weight"], W[f"{q}.conv.1.bias"], W[f"{q}.conv.1.running_mean"], W[f"{q}.conv.1.running_var"], 1e-5)
        ops.norm_fold_running(st2, W[f"{q}.conv.4.weight"], W[f"{q}.conv.4.bias"], W[f"{q}.conv.4.running_mean"], W[f"{q}.conv.4.running_var"], 1e-5)
        r = ops.conv3d(CONV_FWD, u, pk.get(W, f"{q}.conv.0.weight", False, False), None, so, 3, 1,
                       ep_scale=st1.scale, ep_shift=st1.shift, ep_act=ACT_RELU6)
        if stencil_tail and nxt is None and ops.head_stencil_supported(r):
            # last block of an eval pass: conv2 -> BatchNorm(running statistics) -> projection is linear in r and is evaluated as one
            # C -> 1 stencil on the needed patches (decoder_forward, ops.head_stencil); the block's output map never exists
            return r, {"stencil": True, "st2": st2}
        # out_skip (last block of a loss-only teacher pass): patch mask of the output voxels anyone will read -- the rest of the
        # brick grid is not computed at all (P/pretrain_AntoMask.py:421-425 keeps the teacher's l2 of the MASKED patches only)
        o = ops.conv3d(CONV_FWD, r, pk.get(W, f"{q}.conv.3.weight", False, False), None, so, 3, 1,
                       ep_scale=st2.scale, ep_shift=st2.shift, ep_res=nxt, out_mask=out_skip, out_bshift=4 if out_skip is not None else 0)
        return o, None
    c1, pt1 = ops.conv3d(CONV_FWD, u, pk.get(W, f"{q}.conv.0.weight", False, False), None, so, 3, 1, want_partials=train), None
    if train:
        c1, pt1 = c1
    st1 = _batch_norm(c1, W, f"{q}.conv.1", train, pt1, update_running, sync)
    r = ops.norm_apply(c1, st1, ACT_RELU6)
    c2, pt2 = ops.conv3d(CONV_FWD, r, pk.get(W, f"{q}.conv.3.weight", False, False), None, so, 3, 1, want_partials=train), None
    if train:
        c2, pt2 = c2
    st2 = _batch_norm(c2, W, f"{q}.conv.4", train, pt2, update_running, sync)
    if head:
        assert train and nxt is None and st2.sync_world <= 1
        return c2, {"q": q, "xin": x, "u": u, "c1": c1, "st1": st1, "r": r, "c2": c2, "st2": st2, "head": True}
    o = ops.norm_apply(c2, st2, ACT_NONE, res=nxt)            # x = x + to_dec[i+1] fused into the BN apply
    return o, {"q": q, "xin": x, "u": u, "c1": c1, "st1": st1, "r": r, "c2": c2, "st2": st2}


def encoder_forward(spec: Spec, W, pk: PackCache, inp: torch.Tensor, mask: MaskInfo, counts, tape: Optional[Tape] = None,
                    recompute: bool = False) -> List[torch.Tensor]:
    """STUNet.forward(hierarchical=True) under SparseEncoder (P/STUNet_head.py:67-76, P/encoder3D.py:366-367): the 5 stage
    output maps, channels-last, block-sparse (only voxels of active patches are defined)."""
    feats = []
    x = None
    for s in range(spec.n_stage):
        sp = spec.stage_spatial(s)
        if tape is not None and recompute:
            tape.enc_in.append(x)
        for b in range(spec.depth[s]):
            x, rec_ = _enc_block(W, pk, inp, mask, counts, sp, s, b, x, keep=tape is not None)
            if tape is not None and not recompute:
                tape.enc.append(rec_)
        feats.append(x)
    return feats


def densify_forward(spec: Spec, W, pk: PackCache, feats, mask: MaskInfo, counts, tape: Optional[Tape] = None) -> List[torch.Tensor]:
    """P/AnatoMask.py:158-168: norm -> mask-token fill -> projection per level, coarse -> fine (level 4 is dead: P/decoder3D.py:57-60)."""
    n_dec = len(spec.dec_chs) - 1
    to_dec = []
    for i in range(n_dec):
        f = feats[4 - i]
        st = _sparse_norm(f, mask, i, counts, W[f"densify_norms.{i}.weight"], W[f"densify_norms.{i}.bias"], 1e-6)
        d = ops.norm_apply(f, st, ACT_NONE, mask, i, fill=W[f"mask_tokens.{i}"].view(-1))
        pw = f"densify_projs.{i}.weight"
        if pw in W:
            k = W[pw].shape[-1]
            pr = ops.conv3d(CONV_FWD, d, pk.get(W, pw, False, False), W[f"densify_projs.{i}.bias"], tuple(d.shape[1:4]), k, 1)
        else:
            pr = d
        to_dec.append(pr)
        if tape is not None:
            tape.dens.append({"i": i, "f": f, "st": st, "d": d, "k": W[pw].shape[-1] if pw in W else 0})
    return to_dec


HEAD_STENCIL = True   # tools/step_ab.py engine.HEAD_STENCIL=1,0: the eval-mode tail conv -> BatchNorm -> projection as one C -> 1 stencil


def decoder_forward(spec: Spec, W, pk: PackCache, to_dec, train: bool, tape: Optional[Tape] = None, recompute: bool = False,
                    needed_patches: Optional[MaskInfo] = None, fuse_eval: Optional[bool] = None, l2_of: Optional[tuple] = None) -> torch.Tensor:
    """LightDecoder.forward (P/decoder3D.py:55-63): x = 0; per block x += to_dec[i]; UNetBlock; 1x1 proj -> rec fp32 [B,D,H,W].
    needed_patches (eval passes without a tape): only these 16^3 patches of rec are defined.
    l2_of = (inp fp32 [B,D,H,W], l2 fp32 [B,L]): where the eval-mode tail runs as the fused stencil (ops.head_stencil), the raw per-patch
    l2 of the needed patches is written into l2 by the same kernel and None is returned instead of rec (the teacher pass,
    P/pretrain_AntoMask.py:421-425, needs nothing else); otherwise rec is returned and l2 is left untouched."""
    n_dec = len(spec.dec_chs) - 1
    if fuse_eval is None:
        fuse_eval = tape is None
    x = to_dec[0]
    # train mode: the last block's BatchNorm is applied inside the projection (and differentiated with it, ops.proj_norm_bwd)
    head = bool(train and FUSED_HEAD and _sync_world(spec.sync_bn) <= 1 and not spec.dec_inorm)
    st_head = None
    for i in range(n_dec):
        nxt = to_dec[i + 1] if i + 1 < n_dec else None
        last = i == n_dec - 1
        o, rec_ = _dec_block(W, pk, i, x, nxt, train, fuse_eval=fuse_eval,
                             out_skip=needed_patches if (last and tape is None and not train) else None, sync=spec.sync_bn,
                             head=head and last, inorm=spec.dec_inorm,
                             stencil_tail=bool(HEAD_STENCIL and last and tape is None and not train and fuse_eval and not spec.dec_inorm))
        if rec_ is not None and rec_.get("stencil"):
            q = f"{DEC}.{i}"
            weff, beff = ops.head_fold(W[f"{q}.conv.3.weight"], rec_["st2"].scale, rec_["st2"].shift, W["dense_decoder.proj.weight"].view(-1),
                                       W["dense_decoder.proj.bias"])
            B_, D_, H_, W_ = o.shape[:4]
            patches = needed_patches
            if patches is None:                              # the whole volume: every patch is needed
                patches = MaskInfo(torch.ones(B_, D_ // 16, H_ // 16, W_ // 16, device=o.device, dtype=torch.uint8), n_active=B_ * (D_ // 16) * (H_ // 16) * (W_ // 16))
            if l2_of is not None:
                ops.head_stencil(o, weff, beff, patches, inp=l2_of[0], l2=l2_of[1])
                return None
            rec = torch.empty(B_, D_, H_, W_, device=o.device, dtype=torch.float32)
            ops.head_stencil(o, weff, beff, patches, rec=rec)
            return rec
        if head and last:
            st_head = rec_["st2"]
        if tape is not None:
            tape.dec.append({"q": rec_["q"], "xin": x, "nxt": nxt, "head": head and last} if recompute else rec_)
        x = o
    rec = ops.proj_fwd(x, W["dense_decoder.proj.weight"].view(-1), W["dense_decoder.proj.bias"], pre=st_head)
    if tape is not None:
        tape.last = x
    return rec


def forward(spec: Spec, W: Dict[str, torch.Tensor], pk: PackCache, inp: torch.Tensor, mask: MaskInfo, train: bool,
            tape: Optional[Tape] = None, want_feats: bool = False, encoder_only: bool = False, recompute: bool = False,
            want_to_dec0: bool = False, needed_patches: Optional[MaskInfo] = None, l2_out: Optional[torch.Tensor] = None):
    """inp: fp32 [B,D,H,W] (single channel).  Returns rec fp32 [B,D,H,W] (and the 5 encoder maps).
    recompute=True is the P/GC.py policy (torch.utils.checkpoint per encoder stage :324 and per decoder block :68): the tape
    keeps only stage / block INPUTS; backward re-runs that stage's forward before differentiating it."""
    counts = _counts(mask, range(5))
    if tape is not None:
        tape.counts = counts
        tape.recompute = recompute
    feats = encoder_forward(spec, W, pk, inp, mask, counts, tape, recompute)
    if encoder_only:
        return feats
    to_dec = densify_forward(spec, W, pk, feats, mask, counts, tape)
    rec = decoder_forward(spec, W, pk, to_dec, train, tape, recompute, needed_patches, l2_of=(inp, l2_out) if l2_out is not None else None)
    if want_to_dec0:                         # coarsest densified map, channels-last (SparK.forward(return_feat=True), P/AnatoMask.py:172-173)
        return rec, to_dec[0]
    return (rec, feats) if want_feats else rec


# ====================================================================================== backward
# Weight gradients leave the critical chain (norm backward -> dgrad -> norm backward ...): they run on a side HIP stream, where
# the MFMA-bound wgrad kernels overlap the HBM-bound norm-backward passes of the next layer.  Ordering: the side stream waits
# for an event recorded when dy exists; the main stream waits for the side stream before a parameter group is declared final
# (all-reduce hook) and at the end of backward.  The operands of a side-stream launch (x, dy) are kept alive in _SIDE_KEEP until the
# main stream has joined the side stream (`_join_side`): they are then freed on the main stream, behind the wait, like any other tensor.
# (Until round 4 `Tensor.record_stream` did this job: the caching allocator parks such blocks until the side stream's events complete --
# the whole backward, the side stream lags that far -- and meanwhile grows the pool: 140-213 GiB reserved for 47 GiB of live tensors at
# B=16, allocation retries (device synchronisation + release of the whole cache) from B=24 on and in bench.py's second window:
# profiles/r04_y_batch_cliff.txt.  That was the "batch cliff" of DESIGN.md 6.)
FUSED_HEAD = True     # tools/step_ab.py engine.FUSED_HEAD=1,0: the projection head applies / differentiates the last BatchNorm itself
_SIDE: Dict[int, "torch.cuda.Stream"] = {}
_SIDE_KEEP: Dict[int, List[torch.Tensor]] = {}   # per device: operands of side-stream launches not yet joined by the main stream
_USE_SIDE = True      # tools/step_ab.py flips this attribute for same-process A/B timing; no environment switch exists


def _dev_index(dev) -> int:
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _side_stream(dev) -> "torch.cuda.Stream":
    i = _dev_index(dev)
    if i not in _SIDE:
        _SIDE[i] = torch.cuda.Stream(device=dev)
    return _SIDE[i]


def _on_side(dev, tensors, fn):
    if not _USE_SIDE:
        return fn()
    s2 = _side_stream(dev)
    ev = torch.cuda.Event()
    ev.record()
    s2.wait_event(ev)
    _SIDE_KEEP.setdefault(_dev_index(dev), []).extend(tensors)      # (before the launch: kept even if fn raises half-way)
    with torch.cuda.stream(s2):
        fn()


def _join_side(dev):
    """the main stream of `dev` waits for ITS side stream; that device's kept operands are released (another device's side stream may
    still be reading its own)."""
    i = _dev_index(dev)
    if _USE_SIDE and i in _SIDE:
        ev = torch.cuda.Event()
        ev.record(_SIDE[i])
        torch.cuda.current_stream(dev).wait_event(ev)
    _SIDE_KEEP.pop(i, None)               # (freed on the main stream, which now runs behind everything the side stream was given)


def _wgrad_into(pk: PackCache, G, name, mode, x, dy, k, stride, transposed=False, **masks):
    def run():
        dwp = ops.conv3d_wgrad(mode, x, dy, k, stride, f32_split=pk.f32_split, deterministic=pk.deterministic, **masks)
        ops.unpack_grad(dwp, G[name], transposed, accumulate=True)
    _on_side(x.device, (x, dy), run)


def _dec_block_backward(W, G, pk: PackCache, t: dict, g: Optional[torch.Tensor], drec: Optional[torch.Tensor], bias_sum_into: Optional[torch.Tensor]):
    """backward of ONE UNetBlock (`_dec_block`'s record t): g = gradient wrt the block output (None for a fused-head record: the gradient
    arrives as drec, through the projection); accumulates the block's parameter gradients into G; returns the gradient wrt the block INPUT.
    bias_sum_into: densify_projs[i].bias gradient (= per-channel sum of the returned tensor), or None."""
    q = t["q"]
    if t.get("inorm"):
        so = tuple(t["r"].shape[1:4])
        dc2 = _instance_norm_backward(g, t["c2"], t["st2"], ACT_NONE)
        dr = ops.conv3d(CONV_DGRAD, dc2, pk.get(W, f"{q}.conv.3.weight", False, True), None, so, 3, 1)
        _wgrad_into(pk, G, f"{q}.conv.3.weight", CONV_FWD, t["r"], dc2, 3, 1)
        dc1 = _instance_norm_backward(dr, t["c1"], t["st1"], ACT_RELU6)
    elif t.get("head"):
        dc2 = ops.proj_norm_bwd(t["c2"], t["st2"], drec, W["dense_decoder.proj.weight"].view(-1), W[f"{q}.conv.4.weight"],
                                W[f"{q}.conv.4.bias"], G[f"{q}.conv.4.weight"], G[f"{q}.conv.4.bias"],
                                G["dense_decoder.proj.weight"].view(-1), G["dense_decoder.proj.bias"])
    else:
        dc2 = ops.norm_backward(g, None, t["c2"], t["st2"], W[f"{q}.conv.4.weight"], ACT_NONE, None, 0,
                                G[f"{q}.conv.4.weight"], G[f"{q}.conv.4.bias"])
    if not t.get("inorm"):
        so = tuple(t["r"].shape[1:4])
        # (bf16) the reduce pass of the BatchNorm backward rides in this dgrad's epilogue: dr and c1 are not re-read for it.  Only where
        # the dgrad contracts >= 128 channels: the epilogue's work per output tile is fixed (64 activation derivatives per lane), and
        # on the 32- / 64-channel levels it costs as much as the pass it replaces (32->64 @128^3: 4.15 -> 5.61 ms for a 1.7 ms pass)
        fuse = ops.FUSED_NORM_BWD_REDUCE and dc2.dtype == torch.bfloat16 and t["st1"].sync_world <= 1 and dc2.shape[-1] >= 128
        dr = ops.conv3d(CONV_DGRAD, dc2, pk.get(W, f"{q}.conv.3.weight", False, True), None, so, 3, 1,
                        norm_bwd=(t["c1"], t["st1"], ACT_RELU6) if fuse else None)
        dr, red1 = dr if fuse else (dr, None)
        _wgrad_into(pk, G, f"{q}.conv.3.weight", CONV_FWD, t["r"], dc2, 3, 1)
        dc1 = ops.norm_backward(dr, None, t["c1"], t["st1"], W[f"{q}.conv.1.weight"], ACT_RELU6, None, 0,
                                G[f"{q}.conv.1.weight"], G[f"{q}.conv.1.bias"], reduced=red1)
    # (only the per-channel SUM of du is read -- the transposed conv's bias gradient --: the cheaper statistics epilogue)
    du, ptu = ops.conv3d(CONV_DGRAD, dc1, pk.get(W, f"{q}.conv.0.weight", False, True), None, so, 3, 1, want_partials=True, partials_sum_only=True)
    ptu.finalize(None, sum_accum=G[f"{q}.up_sample.bias"])  # ConvT bias gradient = per-channel sum of du
    _wgrad_into(pk, G, f"{q}.conv.0.weight", CONV_FWD, t["u"], dc1, 3, 1)
    si = tuple(t["xin"].shape[1:4])
    gin = ops.conv3d(CONVT_DGRAD, du, pk.get(W, f"{q}.up_sample.weight", True, True), None, si, 4, 2, want_partials=bias_sum_into is not None)
    if bias_sum_into is not None:
        gin, ptg = gin
        ptg.finalize(None, sum_accum=bias_sum_into)
    _wgrad_into(pk, G, f"{q}.up_sample.weight", CONVT_FWD, t["xin"], du, 4, 2, transposed=True)
    return gin


def decoder_backward(spec: Spec, W, G, pk: PackCache, tape: Tape, drec: torch.Tensor, after_group=None) -> List[Optional[torch.Tensor]]:
    """backward of decoder_forward: accumulates the decoder's parameter gradients into G, returns dproj[i] = gradient wrt
    to_dec[i].  (densify_projs[i].bias for i >= 1 -- the per-channel sum of dproj[i] -- is folded into the ConvT-dgrad epilogue.)"""
    n_dec = len(spec.dec_chs) - 1
    # ---- projection (fused-head tapes differentiate it together with the last BatchNorm, in the block's backward)
    fused_head = bool(tape.dec[n_dec - 1].get("head"))
    g = None
    if not fused_head:
        g = ops.proj_bwd(tape.last, drec, W["dense_decoder.proj.weight"].view(-1), G["dense_decoder.proj.weight"].view(-1),
                         G["dense_decoder.proj.bias"], deterministic=pk.deterministic)
        if after_group:
            after_group("proj")
    # ---- decoder, fine -> coarse.  g = grad wrt block output (= grad wrt to_dec[i+1] too)
    dproj: List[Optional[torch.Tensor]] = [None] * n_dec
    for i in reversed(range(n_dec)):
        t = tape.dec[i]
        if tape.recompute:                        # P/GC.py:68: re-run the block forward from its saved input
            _, t = _dec_block(W, pk, i, t["xin"], t["nxt"], True, update_running=False, sync=spec.sync_bn, head=bool(t.get("head")),
                              inorm=spec.dec_inorm)
            tape.dec[i] = None
        if i + 1 < n_dec:
            dproj[i + 1] = g
        need_sum = i > 0 and f"densify_projs.{i}.bias" in G        # densify_projs[i].bias gradient = sum of the block-input gradient
        g = _dec_block_backward(W, G, pk, t, g, drec, G[f"densify_projs.{i}.bias"] if need_sum else None)
        if t.get("head") and after_group:
            after_group("proj")
        if after_group:
            after_group(f"dec{i}")
    dproj[0] = g
    return dproj


def densify_backward(spec: Spec, W, G, pk: PackCache, mask: MaskInfo, tape: Tape, dproj) -> List[Optional[torch.Tensor]]:
    """backward of densify_forward: returns dfeat[s] = gradient wrt the stage-s encoder map (active voxels only; None for s = 0)."""
    n_dec = len(spec.dec_chs) - 1
    dfeat: List[Optional[torch.Tensor]] = [None] * spec.n_stage
    for i in range(n_dec):
        t = tape.dens[i]
        dp = dproj[i]
        if t["k"]:
            pw = f"densify_projs.{i}.weight"
            dd = ops.conv3d(CONV_DGRAD, dp, pk.get(W, pw, False, True), None, tuple(dp.shape[1:4]), t["k"], 1)
            _wgrad_into(pk, G, pw, CONV_FWD, t["d"], dp, t["k"], 1)
            if i == 0:                                            # (levels >= 1 got it from the ConvT-dgrad epilogue)
                ops.chan_sum(dp, None, 0, G[f"densify_projs.{i}.bias"])
        else:
            dd = dp
        dfeat[4 - i] = ops.norm_backward(dd, None, t["f"], t["st"], W[f"densify_norms.{i}.weight"], ACT_NONE, mask, i,
                                         G[f"densify_norms.{i}.weight"], G[f"densify_norms.{i}.bias"],
                                         dtoken=G[f"mask_tokens.{i}"].view(-1), fill=True)
    return dfeat


def _enc_block_backward(W, G, pk: PackCache, inp: torch.Tensor, mask: MaskInfo, t: dict, gout: torch.Tensor, base: Optional[torch.Tensor],
                        free_saved: bool = False) -> Optional[torch.Tensor]:
    """backward of ONE BasicResBlock (`_enc_block`'s record t): gout = gradient wrt the block output (active voxels only); accumulates
    the block's parameter gradients into G; returns the gradient wrt the block INPUT (None for the Cin = 1 stem block).
    base: first block of a stage -- the gradient the block input already has from its densify branch (accumulated into, and returned), or None."""
    p, s = t["p"], t["s"]
    bs = 4 - s
    y2, a1, y1, out, x = t["y2"], t["a1"], t["y1"], t["out"], t["x"]
    dpre = torch.empty_like(y2)                      # gradient of the shortcut branch
    # conv2.bias gradient = sum of dy2 (folded into the apply pass); conv3.bias gradient = sum of dpre = norm2's dbeta
    dy2 = ops.norm_backward(gout, out, y2, t["st2"], W[f"{p}.norm2.weight"], ACT_LRELU, mask, bs,
                            G[f"{p}.norm2.weight"], G[f"{p}.norm2.bias"], dres=dpre,
                            dbeta2=G.get(f"{p}.conv3.bias"), dxsum=G[f"{p}.conv2.bias"])
    sp = tuple(y2.shape[1:4])
    da1 = ops.conv3d(CONV_DGRAD, dy2, pk.get(W, f"{p}.conv2.weight", False, True), None, sp, 3, 1,
                     in_mask=mask, in_bshift=bs, out_mask=mask, out_bshift=bs)
    red1 = None           # (the fused norm-backward reduce, ops.conv3d(norm_bwd=...), does not pay on block-sparse tensors: 64->64 @64^3 +0.12 ms for a 0.05 ms pass)
    _wgrad_into(pk, G, f"{p}.conv2.weight", CONV_FWD, a1, dy2, 3, 1, x_mask=mask, x_bshift=bs, y_mask=mask, y_bshift=bs)
    dy1 = ops.norm_backward(da1, None, y1, t["st1"], W[f"{p}.norm1.weight"], ACT_LRELU, mask, bs,
                            G[f"{p}.norm1.weight"], G[f"{p}.norm1.bias"], dxsum=G[f"{p}.conv1.bias"], reduced=red1)
    stride = t["stride"]
    if s == 0 and t["first"]:                        # Cin = 1 stem: weight/bias gradients only
        ops.stem_conv_wgrad(inp, dy1, 3, mask, bs, G[f"{p}.conv1.weight"].view(-1, 27), None, deterministic=pk.deterministic)
        ops.stem_conv_wgrad(inp, dpre, 1, mask, bs, G[f"{p}.conv3.weight"].view(-1, 1), None, deterministic=pk.deterministic)
        return None
    bsx = bs + (1 if stride == 2 else 0)
    spx = tuple(x.shape[1:4])
    _wgrad_into(pk, G, f"{p}.conv1.weight", CONV_FWD, x, dy1, 3, stride, x_mask=mask, x_bshift=bsx, y_mask=mask, y_bshift=bs)
    if free_saved:
        t["y1"] = t["a1"] = t["y2"] = t["out"] = None      # free as we go
    if t["first"]:
        # block input = output map of stage s-1: add onto its densify gradient if it has one
        gx = ops.conv3d(CONV_DGRAD, dy1, pk.get(W, f"{p}.conv1.weight", False, True), None, spx, 3, stride,
                        in_mask=mask, in_bshift=bs, out_mask=mask, out_bshift=bsx, out=base, accumulate=base is not None)
        _wgrad_into(pk, G, f"{p}.conv3.weight", CONV_FWD, x, dpre, 1, stride, x_mask=mask, x_bshift=bsx, y_mask=mask, y_bshift=bs)
        ops.conv3d(CONV_DGRAD, dpre, pk.get(W, f"{p}.conv3.weight", False, True), None, spx, 1, stride,
                   in_mask=mask, in_bshift=bs, out_mask=mask, out_bshift=bsx, out=gx, accumulate=True)
        return gx
    # identity shortcut
    gx = ops.conv3d(CONV_DGRAD, dy1, pk.get(W, f"{p}.conv1.weight", False, True), None, spx, 3, 1,
                    in_mask=mask, in_bshift=bs, out_mask=mask, out_bshift=bs)
    return ops.add(gx, dpre, out=gx)


def encoder_backward(spec: Spec, W, G, pk: PackCache, inp: torch.Tensor, mask: MaskInfo, tape: Tape, dfeat, after_group=None):
    """backward of encoder_forward, deep -> shallow.  dfeat[s] = gradient wrt the stage-s output map (active voxels only), or
    None (the stage-0 map feeds only stage 1 on the SparK path: its densify branch is dead)."""
    counts = tape.counts
    gstage: List[Optional[torch.Tensor]] = list(dfeat)
    by_stage: Dict[int, List[dict]] = {}
    for t in tape.enc:
        by_stage.setdefault(t["s"], []).append(t)
    for s in reversed(range(spec.n_stage)):
        if tape.recompute:                        # P/GC.py:324: re-run the stage forward from its saved input
            xs, recs = tape.enc_in[s], []
            for b in range(spec.depth[s]):
                xs, r_ = _enc_block(W, pk, inp, mask, counts, spec.stage_spatial(s), s, b, xs)
                recs.append(r_)
            by_stage[s] = recs
        gout = gstage[s]
        if gout is None:                          # (stand-alone SparseEncoder.forward: the caller used only some of the maps)
            gout = torch.zeros_like(by_stage[s][-1]["out"])
        for t in reversed(by_stage[s]):
            gx = _enc_block_backward(W, G, pk, inp, mask, t, gout, gstage[s - 1] if (t["first"] and s > 0) else None, tape.recompute)
            if t["first"] and s > 0:
                gstage[s - 1] = gx
            elif gx is not None:
                gout = gx
            if after_group:
                after_group(f"stage{s}.{t['b']}")
            if gx is None:
                break
    _join_side(inp.device)


def backward(spec: Spec, W: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], pk: PackCache, inp: torch.Tensor, mask: MaskInfo,
             tape: Tape, drec: torch.Tensor, after_group=None, join_before_hook: bool = True):
    """Accumulates parameter gradients into G (fp32, torch layout).  `after_group(tag)` is called when every gradient of a group has
    been ENQUEUED (main stream, or the side stream for the weight gradients): tags 'proj', 'dec3' .. 'dec0', 'densify',
    'stage4.<b>' .. 'stage0.0' in that order (DDP overlap hook).
    join_before_hook=True (default, what SparK._after_group hooks get): the main stream waits for the side stream's weight gradients
    before every hook call, so a hook may read G on the current stream.  False (the trainer): no join -- the hook itself must order
    its work behind BOTH streams (the trainer issues its collectives from the side stream behind a main-stream event)."""
    if after_group is not None and join_before_hook:
        user_hook = after_group

        def after_group(tag, _h=user_hook, _dev=inp.device):
            _join_side(_dev)
            _h(tag)
    try:
        dproj = decoder_backward(spec, W, G, pk, tape, drec, after_group)
        dfeat = densify_backward(spec, W, G, pk, mask, tape, dproj)
        if after_group:
            after_group("densify")
        encoder_backward(spec, W, G, pk, inp, mask, tape, dfeat, after_group)
    finally:
        _join_side(inp.device)            # (also when backward raises: the kept activations must not outlive the call)
