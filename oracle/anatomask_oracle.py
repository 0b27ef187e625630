"""CPU oracle for the AnatoMask MIM pretraining step.

TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path in
``anatomask_amd``; it is never the thing shipped or measured (except as the
``cpu_baseline`` leg of ``bench.py``, where it is the baseline and says so).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import it.

It is a functional restatement (plain torch-CPU fp32 ops on a flat
``{state_dict key: tensor}`` dict, no nn.Module) of the reference algorithm in
``/root/reference/nnunetv2/training/nnUNetTrainer/variants/pretrain/`` (``P/``).
Each function cites the reference lines it follows.  Backward passes come from
torch-CPU autograd over these functions.

Parity status: PINNED against outputs of the reference itself, imported and run
in the build container by ``tests/golden/make_fixtures.py`` (the reference has
no tests or golden vectors of its own for this path).  The
committed fixtures ``tests/golden/*.npz`` are checked by
``tests/test_oracle_golden.py``.  ``ModelEma`` is third-party (``timm``,
version unpinned by the reference, not vendored): restated from its published
behaviour, pinned by the same fixtures at its call sites
(``P/pretrain_AntoMask.py:221,440``).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

ENC = "sparse_encoder.sp_cnn.conv_blocks_context"
DEC = "dense_decoder.dec"


# ----------------------------------------------------------------------------
# configuration helpers
# ----------------------------------------------------------------------------
class Config:
    """Static description of one SparK(STUNet) model (what the reference driver
    hard-codes at P/pretrain_AntoMask.py:184-215)."""

    def __init__(self, dims: Sequence[int], depth: Sequence[int], width: int,
                 input_size: Sequence[int], mask_ratio: float = 0.6,
                 in_ch: int = 1, out_ch: int = 1, downsample: int = 16):
        self.dims = list(dims)
        self.depth = list(depth)
        self.width = int(width)
        self.input_size = tuple(int(v) for v in input_size)
        self.mask_ratio = float(mask_ratio)
        self.in_ch, self.out_ch, self.downsample = in_ch, out_ch, downsample
        self.fmap = tuple(s // downsample for s in self.input_size)       # P/AnatoMask.py:21
        self.L = self.fmap[0] * self.fmap[1] * self.fmap[2]
        self.len_keep = round(self.L * (1 - mask_ratio))                   # P/AnatoMask.py:23
        self.n_stage = 5                                                   # P/STUNet_head.py:41 (range(1,num_pool))
        self.enc_chs = self.dims[:5]                                       # P/STUNet_head.py:65
        self.strides = [1, 2, 2, 2, 2]
        n = round(math.log2(downsample))
        self.dec_chs = [self.width // 2 ** i for i in range(n + 1)]       # P/decoder3D.py:39-40

    @staticmethod
    def stunet_b(input_size=(128, 128, 128), mask_ratio=0.6):
        return Config([32, 64, 128, 256, 512, 512], [1] * 6, 512, input_size, mask_ratio)

    @staticmethod
    def stunet_s(input_size=(48, 48, 48), mask_ratio=0.6):
        return Config([16, 32, 64, 128, 256, 256], [1] * 6, 256, input_size, mask_ratio)


def param_shapes(cfg: Config) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape, in the reference's registration order
    (P/STUNet_head.py:78-94, P/decoder3D.py:13-53, P/AnatoMask.py:37-71)."""
    s: Dict[str, Tuple[int, ...]] = {}
    cin = cfg.in_ch
    for st in range(cfg.n_stage):
        c = cfg.dims[st]
        for b in range(cfg.depth[st]):
            p = f"{ENC}.{st}.{b}"
            ci = cin if b == 0 else c
            s[f"{p}.conv1.weight"] = (c, ci, 3, 3, 3); s[f"{p}.conv1.bias"] = (c,)
            s[f"{p}.norm1.weight"] = (c,); s[f"{p}.norm1.bias"] = (c,)
            s[f"{p}.conv2.weight"] = (c, c, 3, 3, 3); s[f"{p}.conv2.bias"] = (c,)
            s[f"{p}.norm2.weight"] = (c,); s[f"{p}.norm2.bias"] = (c,)
            if b == 0:                                                     # use_1x1conv only on the first block
                s[f"{p}.conv3.weight"] = (c, ci, 1, 1, 1); s[f"{p}.conv3.bias"] = (c,)
        cin = c
    ch = cfg.dec_chs
    for i in range(len(ch) - 1):
        ci, co = ch[i], ch[i + 1]
        p = f"{DEC}.{i}"
        s[f"{p}.up_sample.weight"] = (ci, ci, 4, 4, 4); s[f"{p}.up_sample.bias"] = (ci,)
        s[f"{p}.conv.0.weight"] = (ci, ci, 3, 3, 3)
        for k, c in (("1", ci), ("4", co)):
            if k == "4":
                s[f"{p}.conv.3.weight"] = (co, ci, 3, 3, 3)
            s[f"{p}.conv.{k}.weight"] = (c,); s[f"{p}.conv.{k}.bias"] = (c,)
            s[f"{p}.conv.{k}.running_mean"] = (c,); s[f"{p}.conv.{k}.running_var"] = (c,)
            s[f"{p}.conv.{k}.num_batches_tracked"] = ()
    s["dense_decoder.proj.weight"] = (cfg.out_ch, ch[-1], 1, 1, 1); s["dense_decoder.proj.bias"] = (cfg.out_ch,)
    e = list(reversed(cfg.enc_chs))
    dw = cfg.width
    for i, ew in enumerate(e):
        s[f"densify_norms.{i}.weight"] = (ew,); s[f"densify_norms.{i}.bias"] = (ew,)
    for i, ew in enumerate(e):
        if not (i == 0 and ew == dw):                                      # P/AnatoMask.py:59-65
            k = 1 if i <= 0 else 3
            s[f"densify_projs.{i}.weight"] = (dw, ew, k, k, k); s[f"densify_projs.{i}.bias"] = (dw,)
        dw //= 2
    for i, ew in enumerate(e):
        s[f"mask_tokens.{i}"] = (1, ew, 1, 1, 1)
    return s


def is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def closed_form_state(cfg: Config, salt: float = 0.0) -> Params:
    """Deterministic closed-form weights both the reference and the build can
    evaluate (SURVEY.md §8c 'Weights without big files').  Magnitudes follow the
    reference's initialisers loosely (fan-in scaled) so activations stay O(1)."""
    out: Params = {}
    for idx, (k, shp) in enumerate(param_shapes(cfg).items()):
        n = int(np.prod(shp)) if len(shp) else 1
        t = np.arange(n, dtype=np.float64)
        phi = 0.61 * idx + salt
        wave = np.sin(0.37 * t + phi) + 0.5 * np.sin(1.91 * t + 2.3 * phi)
        if k.endswith("num_batches_tracked"):
            v = np.zeros((), dtype=np.int64)
            out[k] = torch.from_numpy(v.reshape(shp))
            continue
        if k.endswith("running_var"):
            v = 1.0 + 0.25 * wave / 1.5
        elif k.endswith("running_mean"):
            v = 0.1 * wave
        elif len(shp) == 1 and k.endswith("weight"):          # norm gammas
            v = 1.0 + 0.2 * wave / 1.5
        elif len(shp) == 1:                                    # biases / norm betas
            v = 0.05 * wave
        elif k.startswith("mask_tokens"):
            v = 0.02 * wave / 1.5
        else:                                                  # conv kernels
            fan_in = int(np.prod(shp[1:]))
            if "up_sample" in k:
                fan_in = shp[0] * 8                            # 8 taps reach each output voxel
            v = wave * (1.2 / math.sqrt(fan_in))
        out[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shp).copy())
    return out


def seeded_state(cfg: Config, seed: int = 0) -> Params:
    """File-free, WELL-CONDITIONED weights both the reference and the build can evaluate: `np.random.RandomState(seed)` draws
    (legacy MT19937: stable across numpy versions) scaled as the reference's initialisers scale them (SURVEY.md §8
    'Initialisation facts'):
      encoder / densify_projs Conv3d: nn.Conv3d default = kaiming_uniform(a=sqrt 5) -> U(+-1/sqrt(fan_in)), bias likewise;
      decoder Conv3d (incl. proj): trunc_normal(std .02), bias 0                       (P/decoder3D.py:70-73);
      ConvTranspose3d: kaiming_normal(fan_out, relu) -> N(0, 2/(shape[0]*k^3))         (P/decoder3D.py:74-77);
      mask_tokens: trunc_normal(0, .02, +-.02)                                         (P/AnatoMask.py:41-42).
    Norm affines, biases the reference starts at 0 and the BN running statistics get SMALL random offsets (gamma 1 +- 0.1,
    beta / bias +- 0.05, running_mean +- 0.1, running_var 1 +- 0.2) so that every parameter's role is exercised; a freshly
    initialised gamma = 1 / beta = 0 would hide a wrong handling of them.  One stream, registration order."""
    rs = np.random.RandomState(seed)
    out: Params = {}
    for k, shp in param_shapes(cfg).items():
        n = int(np.prod(shp)) if len(shp) else 1
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.int64)
            continue
        if k.endswith("running_var"):
            v = 1.0 + 0.2 * rs.uniform(-1, 1, n)
        elif k.endswith("running_mean"):
            v = 0.1 * rs.uniform(-1, 1, n)
        elif len(shp) == 1 and k.endswith("weight"):           # norm gammas
            v = 1.0 + 0.1 * rs.uniform(-1, 1, n)
        elif k.startswith("mask_tokens"):
            v = np.clip(0.02 * rs.standard_normal(n), -0.02, 0.02)
        elif len(shp) == 1:                                     # biases / norm betas
            wk = k[:-4] + "weight"
            if re.search(r"\.conv[123]\.bias$", k) or k.startswith("densify_projs"):     # nn.Conv3d default bias init
                v = rs.uniform(-1, 1, n) / math.sqrt(int(np.prod(param_shapes(cfg)[wk][1:])))
            else:
                v = 0.05 * rs.uniform(-1, 1, n)
        elif "up_sample" in k:
            v = rs.standard_normal(n) * math.sqrt(2.0 / (shp[0] * int(np.prod(shp[2:]))))
        elif k.startswith("dense_decoder"):
            v = np.clip(0.02 * rs.standard_normal(n), -2.0, 2.0)
        else:                                                   # encoder convs, densify_projs
            v = rs.uniform(-1, 1, n) / math.sqrt(int(np.prod(shp[1:])))
        out[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shp).copy())
    return out


def smooth_volume(B: int, size: Sequence[int], seed: int = 0, noise: float = 0.05) -> torch.Tensor:
    """A learnable synthetic CT-like volume (low-frequency field + a little noise), fp32 (B,1,*size): masked patches are
    predictable from their visible neighbours, so a working training step drives the loss well below 1 (overfit tests)."""
    rs = np.random.RandomState(seed)
    d, h, w = (np.arange(s, dtype=np.float64) for s in size)
    out = np.zeros((B, 1, *size), dtype=np.float64)
    for b in range(B):
        f = rs.uniform(0.03, 0.12, size=(4, 3)); ph = rs.uniform(0, 2 * np.pi, size=(4, 3)); am = rs.uniform(0.5, 1.0, size=4)
        for j in range(4):
            out[b, 0] += am[j] * (np.sin(2 * np.pi * f[j, 0] * d + ph[j, 0])[:, None, None]
                                  * np.sin(2 * np.pi * f[j, 1] * h + ph[j, 1])[None, :, None]
                                  * np.sin(2 * np.pi * f[j, 2] * w + ph[j, 2])[None, None, :])
    out += noise * rs.standard_normal(out.shape)
    return torch.from_numpy(out.astype(np.float32))


def synthetic_volume(B: int, size: Sequence[int], seed: int = 1234) -> torch.Tensor:
    """x ~ N(0,1) fp32 (B,1,*size) from a seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 1, *size, generator=g, dtype=torch.float32)


# ----------------------------------------------------------------------------
# masks  (a2, a4, a6)
# ----------------------------------------------------------------------------
def upsample_mask(active_b1fff: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """P/encoder3D.py:7-10 (_get_active_ex_or_ii, returning_active_ex=True)."""
    a = active_b1fff
    for dim, s in zip((2, 3, 4), size):
        a = a.repeat_interleave(s // a.shape[dim], dim=dim)
    return a


def random_mask(cfg: Config, B: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """P/AnatoMask.py:75-79 (SparK.mask): CPU rand -> argsort -> first len_keep visible."""
    idx = torch.rand(B, cfg.L, generator=generator).argsort(dim=1)[:, :cfg.len_keep]
    m = torch.zeros(B, cfg.L, dtype=torch.bool).scatter_(1, idx, True)
    return m.view(B, 1, *cfg.fmap)


def len_loss_for(cfg: Config, epoch: int, total_epoch: int, guide: bool = True) -> int:
    """P/AnatoMask.py:88-107: how many highest-loss patches are forced masked."""
    keep_ratio = 2 / 3
    if guide:
        keep_ratio = float((epoch + 1) / total_epoch) * 0.5
    return max(int((cfg.L - cfg.len_keep) * keep_ratio), 0)


def generate_mask_from_keys(cfg: Config, loss_pred: torch.Tensor, keys: torch.Tensor, len_loss: int) -> torch.Tensor:
    """Key-driven form of P/AnatoMask.py:81-128 (the `mask` output; `easy_mask`
    is unused by every caller).

    Per sample: the `len_loss` highest-loss patches are never visible (:110);
    among the remaining ids the `len_keep` with the SMALLEST key are visible.
    With keys[id] = position of id in the reference's shuffled permutation this
    reproduces the reference mask exactly; with random keys it is the same
    distribution as np.random.shuffle (:112-114) / argsort(randn) (:99-103)."""
    B, L = loss_pred.shape
    order = torch.argsort(loss_pred, dim=1)                        # ascending, :86
    k = keys.clone().to(torch.float64)
    if len_loss > 0:
        hard = order[:, L - len_loss:]
        k.scatter_(1, hard, float("inf"))
    vis = torch.argsort(k, dim=1, stable=True)[:, :cfg.len_keep]
    m = torch.zeros(B, L, dtype=torch.bool).scatter_(1, vis, True)
    return m.view(B, 1, *cfg.fmap)


# ----------------------------------------------------------------------------
# bf16-STORAGE emulation (tolerance derivation only; identity unless `with storage("bf16")`)
# ----------------------------------------------------------------------------
# The HIP path may keep activations, activation gradients and the MFMA weight copies in bf16 (fp32 accumulation, fp32
# statistics, fp32 master weights).  Under `with storage("bf16"):` the functions below round to bf16 at exactly the points
# where that path stores a C > 1 tensor (forward value AND the gradient flowing back through the same point), so that the
# tests can state: "HIP-bf16 is no further from the fp32 reference than an ideal bf16-storage evaluation of the SAME graph".
_STORAGE = [None]


class storage:
    def __init__(self, kind: Optional[str]):
        assert kind in (None, "bf16")
        self.kind = kind

    def __enter__(self):
        self.prev, _STORAGE[0] = _STORAGE[0], self.kind

    def __exit__(self, *a):
        _STORAGE[0] = self.prev


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def _q(t):
    """storage point of a C > 1 activation (identity in the default fp32 mode)."""
    return _RoundBF16.apply(t) if _STORAGE[0] == "bf16" else t


def _qw(w, x):
    """MFMA weight copy (bf16 in bf16 mode); the Cin == 1 stem convs see the fp32 volume and fp32 weights (on the matrix cores as
    hi + lo bf16 parts: 16 significant bits per operand) -- neither is a bf16-storage tensor."""
    return _RoundBF16.apply(w) if (_STORAGE[0] == "bf16" and x.shape[1] > 1) else w


# ----------------------------------------------------------------------------
# encoder  (a7, a8, a9)
# ----------------------------------------------------------------------------
def sparse_conv3d(x, w, b, stride: int, active_b1fff):
    """P/encoder3D.py:12-15: dense Conv3d, then multiply by the up-sampled mask."""
    pad = w.shape[-1] // 2
    y = _q(F.conv3d(x, _qw(w, x), b, stride=stride, padding=pad))
    return y * upsample_mask(active_b1fff, y.shape[2:]).to(y.dtype)


def sparse_instance_norm(x, gamma, beta, eps: float, active_b1fff):
    """P/encoder3D.py:149-158: statistics pooled over ALL active voxels of the
    local batch per channel (InstanceNorm1d on an unbatched (C,N) input), biased
    variance, affine; inactive voxels come out exactly 0."""
    m = upsample_mask(active_b1fff, x.shape[2:])                  # (B,1,D,H,W) bool
    mf = m.to(x.dtype)
    n = mf.sum()
    mean = (x * mf).sum(dim=(0, 2, 3, 4), keepdim=True) / n
    var = (((x - mean) * mf) ** 2).sum(dim=(0, 2, 3, 4), keepdim=True) / n
    y = (x - mean) / torch.sqrt(var + eps) * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    return y * mf


def basic_res_block(p: Params, prefix: str, x, stride: int, active, has_sc: bool):
    """P/STUNet_head.py:96-103 with every conv/norm in its Sparse* form
    (P/encoder3D.py:301-329; converted InstanceNorm3d keeps eps=1e-5)."""
    y = sparse_conv3d(x, p[f"{prefix}.conv1.weight"], p[f"{prefix}.conv1.bias"], stride, active)
    y = _q(F.leaky_relu(sparse_instance_norm(y, p[f"{prefix}.norm1.weight"], p[f"{prefix}.norm1.bias"], 1e-5, active), 0.01))
    y = sparse_conv3d(y, p[f"{prefix}.conv2.weight"], p[f"{prefix}.conv2.bias"], 1, active)
    y = sparse_instance_norm(y, p[f"{prefix}.norm2.weight"], p[f"{prefix}.norm2.bias"], 1e-5, active)
    if has_sc:
        x = sparse_conv3d(x, p[f"{prefix}.conv3.weight"], p[f"{prefix}.conv3.bias"], stride, active)
    return _q(F.leaky_relu(y + x, 0.01))


def encoder_forward(cfg: Config, p: Params, masked_inp, active) -> List[torch.Tensor]:
    """P/STUNet_head.py:67-76 (hierarchical=True) under P/encoder3D.py:366-367."""
    feats = []
    x = masked_inp
    for st in range(cfg.n_stage):
        for b in range(cfg.depth[st]):
            x = basic_res_block(p, f"{ENC}.{st}.{b}", x, cfg.strides[st] if b == 0 else 1, active, has_sc=(b == 0))
        feats.append(x)
    return feats


# ----------------------------------------------------------------------------
# densify + decoder  (a11, a12)
# ----------------------------------------------------------------------------
def batch_norm3d(p: Params, prefix: str, x, train: bool, new_buffers: Optional[Params]):
    """nn.BatchNorm3d(eps=1e-5, momentum=0.1) as used at P/decoder3D.py:21-22.
    train: batch statistics (biased var) + running update (unbiased var);
    eval: running statistics.  Updated buffers are returned through new_buffers
    (functional, so autograd stays clean)."""
    g, b = p[f"{prefix}.weight"], p[f"{prefix}.bias"]
    if train:
        mean = x.mean(dim=(0, 2, 3, 4))
        var = x.var(dim=(0, 2, 3, 4), unbiased=False)
        if new_buffers is not None:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                new_buffers[f"{prefix}.running_mean"] = 0.9 * p[f"{prefix}.running_mean"] + 0.1 * mean
                new_buffers[f"{prefix}.running_var"] = 0.9 * p[f"{prefix}.running_var"] + 0.1 * var * (n / (n - 1))
                new_buffers[f"{prefix}.num_batches_tracked"] = p[f"{prefix}.num_batches_tracked"] + 1
    else:
        mean, var = p[f"{prefix}.running_mean"], p[f"{prefix}.running_var"]
    sh = (1, -1, 1, 1, 1)
    return (x - mean.view(sh)) / torch.sqrt(var.view(sh) + 1e-5) * g.view(sh) + b.view(sh)


def densify(cfg: Config, p: Params, feats: List[torch.Tensor], active) -> List[torch.Tensor]:
    """P/AnatoMask.py:151-168.  Level 4 (finest) is never consumed by the
    decoder (P/decoder3D.py:57-60 loops over 4 blocks) so it is skipped here;
    its parameters exist but receive no gradient."""
    fe = list(reversed(feats))
    cur = active
    to_dec = []
    n_dec = len(cfg.dec_chs) - 1
    for i, f in enumerate(fe):
        if i < n_dec:
            f = sparse_instance_norm(f, p[f"densify_norms.{i}.weight"], p[f"densify_norms.{i}.bias"], 1e-6, active)
            f = _q(torch.where(upsample_mask(cur, f.shape[2:]), f, p[f"mask_tokens.{i}"].expand_as(f)))
            if f"densify_projs.{i}.weight" in p:
                w = p[f"densify_projs.{i}.weight"]
                f = _q(F.conv3d(f, _qw(w, f), p[f"densify_projs.{i}.bias"], padding=w.shape[-1] // 2))
            to_dec.append(f)
        cur = cur.repeat_interleave(2, 2).repeat_interleave(2, 3).repeat_interleave(2, 4)
    return to_dec


def decoder_forward(cfg: Config, p: Params, to_dec: List[torch.Tensor], train: bool, new_buffers: Optional[Params]):
    """P/decoder3D.py:13-29,55-63: x=0; per block x+=to_dec[i]; ConvT(k4,s2,p1);
    conv3-BN-ReLU6-conv3-BN; finally 1x1 proj."""
    x = 0
    for i in range(len(cfg.dec_chs) - 1):
        x = _q(x + to_dec[i]) if i else x + to_dec[i]
        q = f"{DEC}.{i}"
        x = _q(F.conv_transpose3d(x, _qw(p[f"{q}.up_sample.weight"], x), p[f"{q}.up_sample.bias"], stride=2, padding=1))
        x = _q(F.conv3d(x, _qw(p[f"{q}.conv.0.weight"], x), None, padding=1))
        x = _q(F.relu6(batch_norm3d(p, f"{q}.conv.1", x, train, new_buffers)))
        x = _q(F.conv3d(x, _qw(p[f"{q}.conv.3.weight"], x), None, padding=1))
        x = batch_norm3d(p, f"{q}.conv.4", x, train, new_buffers)
    return F.conv3d(_q(x), p["dense_decoder.proj.weight"], p["dense_decoder.proj.bias"])


# ----------------------------------------------------------------------------
# SparK.forward / forward_loss  (a5, a13)
# ----------------------------------------------------------------------------
def patchify(cfg: Config, x):
    """P/AnatoMask.py:221-228."""
    p = cfg.downsample
    h, w, d = cfg.fmap
    B, C = x.shape[:2]
    x = x.reshape(B, C, h, p, w, p, d, p)
    x = torch.einsum("bchpwqdg->bhwdpqgc", x)
    return x.reshape(B, h * w * d, C * p ** 3)


def unpatchify(cfg: Config, bln):
    """P/AnatoMask.py:230-237."""
    p = cfg.downsample
    h, w, d = cfg.fmap
    B, C = bln.shape[0], bln.shape[-1] // p ** 3
    x = bln.reshape(B, h, w, d, p, p, p, C)
    x = torch.einsum("bhwdpqgc->bchpwqdg", x)
    return x.reshape(B, C, h * p, w * p, d * p)


def spark_forward(cfg: Config, p: Params, inp, active, train: bool, new_buffers: Optional[Params] = None,
                  return_volume: bool = False):
    """P/AnatoMask.py:137-176 -> (patchify(inp), patchify(rec)), both (B,L,N)."""
    masked = inp * upsample_mask(active, inp.shape[2:]).to(inp.dtype)      # :144-147
    feats = encoder_forward(cfg, p, masked, active)                        # :150
    to_dec = densify(cfg, p, feats, active)                                # :151-168
    rec = decoder_forward(cfg, p, to_dec, train, new_buffers)              # :170
    if return_volume:
        return rec
    return patchify(cfg, inp), patchify(cfg, rec)


def forward_loss(inp_bln, rec_bln, active):
    """P/AnatoMask.py:190-202: per-patch normalised target (unbiased var, eps 1e-6),
    mean squared error per patch, averaged over masked patches."""
    mean = inp_bln.mean(dim=-1, keepdim=True)
    var = inp_bln.var(dim=-1, keepdim=True)
    tgt = (inp_bln - mean) / (var + 1.e-6) ** .5
    l2 = ((rec_bln - tgt) ** 2).mean(dim=2)
    non_active = active.logical_not().int().view(active.shape[0], -1)
    rec_loss = l2 * non_active
    loss = rec_loss.sum() / (non_active.sum() + 1e-8)
    return loss, rec_loss


def teacher_patch_loss(inp_bln, rec_bln, active):
    """P/pretrain_AntoMask.py:423-425: RAW (un-normalised) target, zero at visible patches."""
    l2 = ((rec_bln - inp_bln) ** 2).mean(dim=2)
    return l2 * active.logical_not().int().view(active.shape[0], -1)


# ----------------------------------------------------------------------------
# optimiser side  (a15, a16, a17)
# ----------------------------------------------------------------------------
def trainable_keys(cfg: Config) -> List[str]:
    return [k for k in param_shapes(cfg) if not is_buffer(k)]


def clip_grad_norm(grads: Params, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (P/pretrain_AntoMask.py:437): norm of per-tensor
    L2 norms; coef = max_norm/(total+1e-6) clamped to 1; every grad is scaled."""
    norms = [g.norm(2) for g in grads.values() if g is not None]
    total = torch.stack(norms).norm(2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        if g is not None:
            g.mul_(coef)
    return total


def adamw_step(p: Params, grads: Params, state: Dict[str, Dict[str, torch.Tensor]], step: int, lr: float,
               wd: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.AdamW as configured at P/pretrain_AntoMask.py:349-356.  Both
    param groups get weight_decay=wd (the groups only carry *_scale keys that are
    never applied).  Params with grad None are skipped."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, g in grads.items():
        if g is None:
            continue
        w = p[k]
        st = state.setdefault(k, {"m": torch.zeros_like(w), "v": torch.zeros_like(w)})
        w.mul_(1 - lr * wd)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(st["m"], denom, value=-(lr / bc1))


def ema_update(ema: Params, model: Params, decay: float):
    """timm.utils.ModelEma.update (third-party; call site P/pretrain_AntoMask.py:440):
    for EVERY state_dict entry (params and BN buffers, incl. int64 num_batches_tracked)
    ema = decay*ema + (1-decay)*model, written back in the entry's own dtype."""
    with torch.no_grad():
        for k, e in ema.items():
            e.copy_(e * decay + (1.0 - decay) * model[k])


def ema_decay_for_epoch(i: int, total_epochs: int) -> float:
    """P/pretrain_AntoMask.py:383-386."""
    q = total_epochs // 4
    return 0.999 + i / q * (0.9999 - 0.999) if i < q else 0.9999


def lr_schedule(epochs: int, base_lr: float = 1e-4, warmup: int = 20, warmup_start_lr: float = 1e-6,
                eta_min: float = 0.0) -> List[float]:
    """nnunetv2/training/lr_scheduler/LinearWarmupCosine.py:65-100, chained form,
    stepped once per epoch (P/pretrain_AntoMask.py:359,452).  Returns the lr in
    force during epoch 0..epochs (inclusive)."""
    lrs = [warmup_start_lr]
    lr = warmup_start_lr
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


# ----------------------------------------------------------------------------
# one training step  (a1)
# ----------------------------------------------------------------------------
class StepState:
    """Student weights, teacher (EMA) weights, AdamW state, step counter."""

    def __init__(self, cfg: Config, weights: Params):
        self.cfg = cfg
        self.student: Params = {k: v.clone() for k, v in weights.items()}
        self.teacher: Params = {k: v.clone() for k, v in weights.items()}   # deepcopy at P/pretrain_AntoMask.py:221
        self.opt: Dict[str, Dict[str, torch.Tensor]] = {}
        self.step = 0


def student_loss_and_grads(cfg: Config, weights: Params, inp, mask, train: bool = True):
    """Student forward + loss + backward (P/pretrain_AntoMask.py:429-435).
    Returns (loss, rec_loss(B,L), grads, new_buffers)."""
    keys = trainable_keys(cfg)
    leaves = {k: weights[k].detach().clone().requires_grad_(True) for k in keys}
    p = dict(weights)
    p.update(leaves)
    new_buffers: Params = {}
    inp_p, rec_p = spark_forward(cfg, p, inp, mask, train=train, new_buffers=new_buffers)
    loss, rec_loss = forward_loss(inp_p, rec_p, mask)
    loss.backward()
    grads = {k: (leaves[k].grad.detach().clone() if leaves[k].grad is not None else None) for k in keys}
    return loss.detach(), rec_loss.detach(), grads, new_buffers


def train_step(st: StepState, inp, mask1, sampler_keys, epoch: int, total_epoch: int, lr: float,
               ema_decay: float, clip: float = 12.0, wd: float = 1e-5, eps: float = 1e-8, return_grads: bool = False):
    """One AnatoMask step, P/pretrain_AntoMask.py:418-441 (fp32 branch), with the
    random draws (mask1, sampler keys) supplied by the caller (teacher-forced).
    return_grads: also return every live parameter gradient as autograd left it (before clip_grad_norm_ scales it)."""
    cfg = st.cfg
    with torch.no_grad():                                                   # :421-425
        inp1, rec1 = spark_forward(cfg, st.teacher, inp, mask1, train=False)
        recon = teacher_patch_loss(inp1, rec1, mask1)
    ll = len_loss_for(cfg, epoch, total_epoch, guide=True)                  # :427
    mask = generate_mask_from_keys(cfg, recon, sampler_keys, ll)
    loss, rec_loss, grads, new_buf = student_loss_and_grads(cfg, st.student, inp, mask, train=True)
    st.student.update(new_buf)                                              # BN running stats
    live = {k: g for k, g in grads.items() if g is not None}
    raw = {k: g.clone() for k, g in live.items()} if return_grads else None   # what loss.backward() left in .grad (before clipping)
    gnorm = clip_grad_norm(live, clip)                                      # :437
    st.step += 1
    adamw_step(st.student, live, st.opt, st.step, lr, wd, eps=eps)          # :438
    ema_update(st.teacher, st.student, ema_decay)                           # :440
    out = {"loss": float(loss), "grad_norm": float(gnorm), "mask": mask, "recon_loss": recon, "rec_loss": rec_loss}
    if return_grads:
        out["grads"] = raw
    return out


def plain_spark_step(st: StepState, inp, mask, lr: float, clip: float = 12.0, wd: float = 1e-5, eps: float = 1e-8):
    """One plain-SparK step, the fp32 branch of P/pretrain.py:388-409 over P/spark3D.py:98-146: the random mask IS the training
    mask (supplied by the caller), the loss is the normalised masked MSE computed in `forward`, then backward, clip_grad_norm_,
    AdamW.  No teacher, no EMA."""
    cfg = st.cfg
    loss, rec_loss, grads, new_buf = student_loss_and_grads(cfg, st.student, inp, mask, train=True)
    st.student.update(new_buf)
    live = {k: g for k, g in grads.items() if g is not None}
    gnorm = clip_grad_norm(live, clip)
    st.step += 1
    adamw_step(st.student, live, st.opt, st.step, lr, wd, eps=eps)
    return {"loss": float(loss), "grad_norm": float(gnorm), "rec_loss": rec_loss}

