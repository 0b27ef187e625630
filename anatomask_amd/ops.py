"""Thin torch-tensor front end of the C ABI: every function takes device tensors (channels-last
[B,D,H,W,C]; fp32 or bf16), extracts raw pointers + the current HIP stream, and calls the HIP
library.  torch is used for memory and streams only; no arithmetic happens here."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip
from .hip import ACT_LRELU, ACT_NONE, ACT_RELU6, CONV_DGRAD, CONV_FWD, CONVT_DGRAD, CONVT_FWD  # noqa: F401


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return hip.DT_F32
    if t.dtype == torch.bfloat16:
        return hip.DT_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


# fp32-storage models compute their matrix products in one of two ways: exact fp32 products (v_mfma_f32_16x16x4_f32, the parity mode: 1/16
# of the bf16 matrix rate) or AM_DT_F32S: products from bf16 hi / lo splits of both operands with fp32 accumulation (16 significant bits
# per operand, 1/4 of the bf16 rate in the convolutions, three bf16 passes in the weight gradients).  The mode is a property of a MODEL
# (SparK.f32_split -> its PackCache): it travels with every packed weight copy (`_am_f32_split`, set by pack_weight) into conv3d and is an
# explicit argument of conv3d_wgrad -- no process-wide switch.  DEFAULT_F32_SPLIT is only what a newly CONSTRUCTED model starts with
# (anatomask_amd.set_f32_products).  The reference computes in fp32 (AMP = False, P/pretrain_AntoMask.py:239).
DEFAULT_F32_SPLIT = False


def _dtc(t: torch.Tensor, f32_split: bool) -> int:
    """dtype code of a CONVOLUTION operand (am_conv3d / am_pack_weight / am_packed_dims / am_conv3d_partials_rows)."""
    return hip.DT_F32S if (t.dtype == torch.float32 and f32_split) else _dt(t)


def _is_split(w_packed: torch.Tensor) -> bool:
    """the fp32 product mode a packed weight copy was made for (False for anything pack_weight did not mark)."""
    return getattr(w_packed, "_am_f32_split", False) is True


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class MaskInfo:
    """uint8 patch mask [B, fd, fh, fw] (1 = active)."""

    def __init__(self, mask_u8: torch.Tensor, n_active: Optional[int] = None):
        """n_active: number of active patches when the caller knows it (the trainer: B * len_keep); otherwise it is counted on
        first use (one host synchronisation, as the reference's `nonzero` calls incur in every layer)."""
        assert mask_u8.dtype == torch.uint8 and mask_u8.dim() == 4 and mask_u8.is_contiguous()
        self.t = mask_u8
        self.B, self.fd, self.fh, self.fw = mask_u8.shape
        self.n_active = n_active
        self._list: Optional[torch.Tensor] = None

    def active_list(self):
        """(device int32 list of the active patches, their number) -- built once per mask (am_mask_compact)."""
        if self._list is None:
            if max(self.B, self.fd, self.fh, self.fw) > 255:
                return None, 0
            if self.n_active is None:
                self.n_active = int(self.t.count_nonzero().item())
            buf = torch.empty(self.t.numel() + 1, device=self.t.device, dtype=torch.int32)
            hip.lib().mask_compact(self.t.data_ptr(), self.B, self.fd, self.fh, self.fw, buf.data_ptr(), buf[-1:].data_ptr(), _stream())
            self._list = buf
        return self._list, self.n_active

    @staticmethod
    def from_bool(active_b1fff: torch.Tensor, device) -> "MaskInfo":
        m = active_b1fff.reshape(active_b1fff.shape[0], *active_b1fff.shape[-3:]).to(device=device, dtype=torch.uint8).contiguous()
        return MaskInfo(m)


def _mk(mask: Optional[MaskInfo]):
    if mask is None:
        return None, 1, 1, 1
    return mask.t.data_ptr(), mask.fd, mask.fh, mask.fw


def _al(mask: Optional[MaskInfo]):
    """(active-patch list pointer, n_active) for the row-walk kernels, or (None, 0)."""
    if mask is None:
        return None, 0
    lst, n = mask.active_list()
    return (lst.data_ptr(), n) if lst is not None and n > 0 else (None, 0)


# ------------------------------------------------------------------ weights
def packed_dims(dtype: torch.dtype, rows: int, k: int):
    """Same rule as am_packed_dims: rows to the 32/64-channel output tile, K to the 64-byte channel slab."""
    tile = 32 if rows <= 32 else 64
    kc = 32 if dtype == torch.bfloat16 else 16
    return (rows + tile - 1) // tile * tile, (k + kc - 1) // kc * kc


class ConvPartials:
    """Per-workgroup per-channel (sum, sumsq) rows a conv launch leaves behind for the norm / bias-grad that follows."""

    def __init__(self, mode, ksize, stride, B, out_spatial, cout, device, out_sparse=False, out_bshift=0, dtype=hip.DT_F32, cin=64, n_active=0):
        import ctypes
        n = ctypes.c_int(0)
        hip.lib().conv3d_partials_rows(mode, dtype, ksize, stride, B, *out_spatial, cin, cout, int(out_sparse), out_bshift, n_active, ctypes.addressof(n))
        self.rows, self.C = n.value, cout
        self.t = torch.empty(self.rows, cout, 2, device=device, dtype=torch.float32)

    def reduce(self, sums: Optional[torch.Tensor] = None, sum_accum: Optional[torch.Tensor] = None):
        hip.lib().partials_reduce(self.t.data_ptr(), self.rows, self.C, _p(sums), _p(sum_accum), _stream())

    def finalize(self, st: Optional["NormStats"], gamma=None, beta=None, eps: float = 0.0, run_mean=None, run_var=None,
                 momentum: float = 0.1, num_batches_tracked=None, sum_accum: Optional[torch.Tensor] = None):
        """partials -> per-channel sums -> mean / rstd / scale / shift of `st` (+ BN running stats, num_batches_tracked) in ONE
        launch (am_partials_finalize); st None: only `sum_accum[c] += sum` (bias gradients)."""
        ws = _stats_workspace(self.t.device, self.C)
        if st is None:
            hip.lib().partials_finalize(self.t.data_ptr(), self.rows, self.C, ws.data_ptr(), None, 1.0, None, None, 0.0, None, None, None,
                                        None, None, None, 0.0, None, _p(sum_accum), _stream())
            return
        hip.lib().partials_finalize(self.t.data_ptr(), self.rows, self.C, ws.data_ptr(), _p(st.count_ptr), float(st.count_host),
                                    gamma.data_ptr(), beta.data_ptr(), eps, st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(),
                                    st.shift.data_ptr(), _p(run_mean), _p(run_var), momentum, _p(num_batches_tracked), _p(sum_accum), _stream())
        st.nrep = 1


    def finalize_bwd(self, st: "NormStats", gamma: torch.Tensor, sc: "NormBwdScratch", dgamma=None, dbeta=None, dbeta2=None):
        """rows of a conv3d(norm_bwd=...) launch -> k0 / k1 / k2 of the apply pass + the affine gradients (am_norm_bwd_from_partials)."""
        ws = _stats_workspace(self.t.device, self.C)
        hip.lib().norm_bwd_from_partials(self.t.data_ptr(), self.rows, self.C, ws.data_ptr(), _p(st.count_ptr), float(st.count_host),
                                         gamma.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(),
                                         sc.k[2].data_ptr(), _p(dgamma), _p(dbeta), _p(dbeta2), _stream())


_WS = {}


FIN_REP = 16   # AM_FIN_REP


def _stats_workspace(device, C: int) -> torch.Tensor:
    """zero-initialised scratch of am_partials_finalize (left zero by every call), one per (device, stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < FIN_REP * 2 * C + 2:
        ws = _WS[key] = torch.zeros(FIN_REP * 2 * max(C, 2048) + 2, device=device, dtype=torch.float64)
    return ws


import ctypes as _ct

_ROWS_OUT = _ct.c_int(0)
_ROWS_ADDR = _ct.addressof(_ROWS_OUT)


def pack_weight(w: torch.Tensor, dtype: torch.dtype, transposed_conv: bool, for_dgrad: bool, f32_split: bool = False) -> torch.Tensor:
    """torch-layout fp32 weight -> packed [taps][rows][K] in `dtype`.
    Conv3d weight (Cout,Cin,k,k,k); ConvTranspose3d weight (Cin,Cout,k,k,k).
    forward: rows = output channels, K = input channels; dgrad: swapped."""
    d0, d1 = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    cout, cin = (d1, d0) if transposed_conv else (d0, d1)
    s_out, s_in = (taps, d1 * taps) if transposed_conv else (d1 * taps, taps)
    if not for_dgrad:
        R, K, sr, sk = cout, cin, s_out, s_in
    else:
        R, K, sr, sk = cin, cout, s_in, s_out
    Rp, Kp = packed_dims(dtype, R, K)
    out = torch.empty(taps, Rp, Kp, device=w.device, dtype=dtype)
    split = bool(out.dtype == torch.float32 and f32_split)
    hip.lib().pack_weight(_dtc(out, split), w.data_ptr(), out.data_ptr(), R, K, taps, sr, sk, Rp, Kp, _stream())
    out.logical = (R, K)
    out._am_f32_split = split      # (what the copy was made for: conv3d launches the matching kernels, PackCache re-makes it when the mode changes)
    out.pack_args = (R, K, taps, sr, sk, Rp, Kp)
    return out


class PackTable:
    """Device table of am_pack_desc for `am_pack_weights_batched`: every (master weight, packed copy) pair of a model."""

    def __init__(self, pairs, dtype: torch.dtype, device, f32_split: bool = False):
        import struct
        self.f32_split = bool(dtype == torch.float32 and f32_split)
        blob, first = b"", 0
        for w, out in pairs:
            R, K, taps, sr, sk, Rp, Kp = out.pack_args
            assert taps <= 64
            blob += struct.pack("<QQqqiiiiii", w.data_ptr(), out.data_ptr(), sr, sk, R, K, taps, Rp, Kp, first)
            first += Rp * ((Kp + 63) // 64)
        self.n, self.blocks, self.dtype = len(pairs), first, dtype
        self.ptrs = [(w.data_ptr(), out.data_ptr()) for w, out in pairs]
        self.dev = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)

    def matches(self, pairs) -> bool:
        return len(pairs) == self.n and all(p == (w.data_ptr(), o.data_ptr()) for p, (w, o) in zip(self.ptrs, pairs))

    def repack(self):
        code = hip.DT_BF16 if self.dtype == torch.bfloat16 else (hip.DT_F32S if self.f32_split else hip.DT_F32)
        hip.lib().pack_weights_batched(code, self.dev.data_ptr(), self.n, self.blocks, _stream())


def unpack_grad(dw_packed: torch.Tensor, grad_out: torch.Tensor, transposed_conv: bool, accumulate: bool):
    """packed fp32 [taps][Cy][Cx] (Cy = forward-output channels) -> torch-layout gradient."""
    taps, R, K = dw_packed.shape
    d1 = grad_out.shape[1]
    if transposed_conv:      # weight (Cin=Cx, Cout=Cy, taps)
        sr, sk = taps, d1 * taps
    else:                    # weight (Cout=Cy, Cin=Cx, taps)
        sr, sk = d1 * taps, taps
    hip.lib().unpack_grad(dw_packed.data_ptr(), grad_out.data_ptr(), R, K, taps, sr, sk, int(accumulate), _stream())


# ------------------------------------------------------------------ convolutions
def conv3d(mode: int, x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], out_spatial: Tuple[int, int, int],
           ksize: int, stride: int, in_mask: Optional[MaskInfo] = None, in_bshift: int = 0,
           out_mask: Optional[MaskInfo] = None, out_bshift: int = 0, out: Optional[torch.Tensor] = None,
           accumulate: bool = False, want_partials: bool = False, ep_scale: Optional[torch.Tensor] = None,
           ep_shift: Optional[torch.Tensor] = None, ep_res: Optional[torch.Tensor] = None, ep_act: int = 0,
           norm_bwd: Optional[tuple] = None, partials_sum_only: bool = False):
    """partials_sum_only (with want_partials): only the per-channel SUM of the rows will be read (AM_CONV_PARTIALS_SUM_ONLY).
    ep_*: fused store epilogue y = act(conv * scale + shift + res) (eval-mode BatchNorm / skip add / activation).
    norm_bwd = (x_pre, st, act): the output is the gradient wrt act(norm(x_pre)); the launch also leaves the norm-backward sums in
    its partial rows (am_conv3d_nbred) -> returns (out, rows) for norm_backward(..., reduced=rows).  bf16 only."""
    B, Di, Hi, Wi, Cin = x.shape
    Cout, Kl = w_packed.logical
    assert Kl == Cin and w_packed.dtype == x.dtype, (w_packed.logical, x.shape)
    Do, Ho, Wo = out_spatial
    mk = in_mask or out_mask
    mp, fd, fh, fw = _mk(mk)
    # the active-patch list: thin block-sparse layers (Cin <= 32) run on the resident-weight kernel, which walks the active bricks, and the
    # levels whose patches are smaller than a brick on the voxel-list gather kernel (conv_gather.hip)
    # (the list is that of OUT_mask: launches with a block-sparse output enumerate their live bricks from it)
    alp, aln = _al(out_mask) if (out_mask is not None and x.dtype == torch.bfloat16) else (None, 0)
    if norm_bwd is not None:
        xp, st, nact = norm_bwd
        assert x.dtype == torch.bfloat16 and xp.dtype == torch.bfloat16 and tuple(xp.shape) == (B, Do, Ho, Wo, Cout) and bias is None
        part = ConvPartials(mode, ksize, stride, B, out_spatial, Cout, x.device, out_mask is not None, out_bshift, _dt(x), 64, 0)   # (cin 64: the generic kernel's row count)
        if out is None:
            out = torch.empty(B, Do, Ho, Wo, Cout, device=x.device, dtype=x.dtype)
        hip.lib().conv3d_nbred(mode, _dt(x), ksize, stride, x.data_ptr(), w_packed.data_ptr(), out.data_ptr(),
                               B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout,
                               in_mask.t.data_ptr() if in_mask else None, in_bshift,
                               out_mask.t.data_ptr() if out_mask else None, out_bshift, fd, fh, fw, int(accumulate),
                               part.t.data_ptr(), xp.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), int(nact), _ROWS_ADDR, _stream())
        part.rows = _ROWS_OUT.value
        return out, part
    split = _is_split(w_packed)                                  # (the fp32 product mode rides on the packed weight copy)
    part = ConvPartials(mode, ksize, stride, B, out_spatial, Cout, x.device, out_mask is not None, out_bshift, _dtc(x, split), Cin, aln) if want_partials else None
    if out is None:
        out = torch.empty(B, Do, Ho, Wo, Cout, device=x.device, dtype=x.dtype)
    rows = _ROWS_OUT
    hip.lib().conv3d(mode, _dtc(x, split), ksize, stride, x.data_ptr(), w_packed.data_ptr(), _p(bias), out.data_ptr(),
                     B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout,
                     in_mask.t.data_ptr() if in_mask else None, in_bshift,
                     out_mask.t.data_ptr() if out_mask else None, out_bshift, fd, fh, fw,
                     int(accumulate) | (2 if (partials_sum_only and part is not None) else 0),
                     part.t.data_ptr() if part else None, _p(ep_scale), _p(ep_shift),
                     ep_res.data_ptr() if ep_res is not None else None, int(ep_act), alp, aln, _ROWS_ADDR if part else None, _stream())
    if part is not None:
        part.rows = rows.value                                   # what this launch actually wrote (<= the allocated bound)
    return (out, part) if want_partials else out


def conv3d_prenorm_supported(x: torch.Tensor, w_packed: torch.Tensor, out_spatial, ksize: int, stride: int, mask: Optional[MaskInfo], out_bshift: int) -> bool:
    """am_conv3d_prenorm serves this launch (bf16, Cin <= 32, k3, block-sparse with an active-patch list: conv_rw.hip)."""
    if mask is None or x.dtype != torch.bfloat16 or _is_split(w_packed):
        return False
    lst, n = mask.active_list()
    if lst is None or n <= 0:
        return False
    Cout, Cin = w_packed.logical
    return hip.lib()._lib.am_conv3d_prenorm_supported(CONV_FWD, _dt(x), ksize, stride, x.shape[0], *out_spatial, Cin, Cout, 1,
                                                      out_bshift + (1 if stride == 2 else 0), out_bshift, n) == 1


def conv3d_prenorm(x: torch.Tensor, st: "NormStats", act: int, w_packed: torch.Tensor, bias: Optional[torch.Tensor], out_spatial, ksize: int,
                   stride: int, mask: MaskInfo, out_bshift: int, want_partials: bool = False):
    """conv(act(x * st.scale + st.shift)) with the norm + activation applied while the source rows are staged: the normalised map
    never exists (passes that keep no tape).  x: the norm's INPUT, block-sparse under `mask`."""
    B, Di, Hi, Wi, Cin = x.shape
    Cout, Kl = w_packed.logical
    assert Kl == Cin and w_packed.dtype == x.dtype
    Do, Ho, Wo = out_spatial
    lst, n = mask.active_list()
    part = ConvPartials(CONV_FWD, ksize, stride, B, out_spatial, Cout, x.device, True, out_bshift, _dt(x), Cin, n) if want_partials else None
    out = torch.empty(B, Do, Ho, Wo, Cout, device=x.device, dtype=x.dtype)
    hip.lib().conv3d_prenorm(CONV_FWD, _dt(x), ksize, stride, x.data_ptr(), w_packed.data_ptr(), _p(bias), out.data_ptr(), B, Di, Hi, Wi, Cin,
                             Do, Ho, Wo, Cout, mask.t.data_ptr(), out_bshift + (1 if stride == 2 else 0), out_bshift, mask.fd, mask.fh, mask.fw,
                             part.t.data_ptr() if part else None, st.scale.data_ptr(), st.shift.data_ptr(), int(act), lst.data_ptr(), n,
                             _ROWS_ADDR if part else None, _stream())
    if part is not None:
        part.rows = _ROWS_OUT.value
    return (out, part) if want_partials else out


# Deterministic weight gradients (am_conv3d_wgrad's det_workspace): per-slot partial sums + an ordered fold instead of fp32 atomics.
# A property of a MODEL (SparK.deterministic_wgrad -> its PackCache -> the `deterministic` argument of the three gradient ops below);
# DETERMINISTIC_WGRAD is only what a direct caller of these ops gets when it passes deterministic=None (tests, tools).
DETERMINISTIC_WGRAD = False


def _det(flag) -> bool:
    return DETERMINISTIC_WGRAD if flag is None else bool(flag)
_DET_WS: dict = {}


def _det_workspace(device, floats_per_slot: int):
    """one cached workspace per device and stream (the side stream's weight gradients run concurrently with the main stream's):
    256 MB, or 8 slots of the largest gradient if that is more."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    need = max(64 << 20, 8 * floats_per_slot)
    t = _DET_WS.get(key)
    if t is None or t.numel() < need:
        t = _DET_WS[key] = torch.empty(need, device=device, dtype=torch.float32)
    return t


GATHER_WGRAD = True      # (tools: same-process A/B of the gather-form weight gradient)


def conv3d_wgrad(mode: int, x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int,
                 x_mask: Optional[MaskInfo] = None, x_bshift: int = 0, y_mask: Optional[MaskInfo] = None,
                 y_bshift: int = 0, f32_split: bool = False, deterministic: Optional[bool] = None) -> torch.Tensor:
    B, Dx, Hx, Wx, Cx = x.shape
    _, Dy, Hy, Wy, Cy = dy.shape
    taps = ksize ** 3
    dw = torch.zeros(taps, Cy, Cx, device=x.device, dtype=torch.float32)
    mk = x_mask or y_mask
    mp, fd, fh, fw = _mk(mk)
    ws = _det_workspace(x.device, taps * Cy * Cx) if _det(deterministic) else None

    al = _al(y_mask)
    # levels with 1- / 2-voxel patches: the gather form (K-major copies of the active voxels + plain GEMMs) needs scratch
    gb = 0
    if GATHER_WGRAD and al[1] > 0 and x.dtype == torch.bfloat16 and y_bshift <= 1:
        out_b = _ct.c_long(0)
        hip.lib().conv3d_wgrad_gather_bytes(mode, _dt(x), ksize, stride, B, Cx, Cy, y_bshift, int(y_mask is not None), al[1], fd, fh, fw, _ct.byref(out_b))
        gb = int(out_b.value)
    gws = torch.empty(gb, device=x.device, dtype=torch.uint8) if gb > 0 else None

    def launch(xx, yy):
        hip.lib().conv3d_wgrad(mode, _dt(xx), ksize, stride, xx.data_ptr(), yy.data_ptr(), dw.data_ptr(),
                               B, Dx, Hx, Wx, Cx, Dy, Hy, Wy, Cy,
                               x_mask.t.data_ptr() if x_mask else None, x_bshift,
                               y_mask.t.data_ptr() if y_mask else None, y_bshift, fd, fh, fw,
                               ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, *al,
                               gws.data_ptr() if gws is not None else None, gb, _stream())
    if x.dtype == torch.float32 and f32_split:
        # AM_DT_F32S: dW = X^T dY with X = Xh + Xl, dY = Yh + Yl (bf16 planes, am_split_bf16): three bf16 matrix-core contractions
        # accumulate into the ONE fp32 gradient (atomics, or the deterministic fold: both add into dw); the lo lo term (2^-16 of the
        # result) is dropped.  The planes are transient (2 x 2 bytes per element = the size of the fp32 tensor).
        xh, xl = split_bf16(x)
        yh, yl = split_bf16(dy)
        launch(xh, yh); launch(xl, yh); launch(xh, yl)
        return dw
    launch(x, dy)
    return dw


def split_bf16(x: torch.Tensor):
    """fp32 tensor -> (hi, lo) bf16 tensors of the same shape: hi = bf16(x), lo = bf16(x - hi) (am_split_bf16)."""
    x = x.contiguous()
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    hip.lib().split_bf16(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), _stream())
    return hi, lo


def stem_conv_fwd(x_b1: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mask: Optional[MaskInfo], bshift: int,
                  dtype: torch.dtype, want_partials: bool = False, out: Optional[torch.Tensor] = None):
    """x_b1: fp32 [B,D,H,W]; w: (C,1,k,k,k) fp32.  want_partials: also return the (sum, sumsq) rows for the norm that follows."""
    B, D, H, W = x_b1.shape
    Cc, k = w.shape[0], w.shape[2]
    y = torch.empty(B, D, H, W, Cc, device=x_b1.device, dtype=dtype) if out is None else out
    mp, fd, fh, fw = _mk(mask)
    part = None
    if want_partials:
        part = ConvPartials.__new__(ConvPartials)
        part.rows, part.C = B * (D // 4) * (H // 8) * (W // 16), Cc
        part.t = torch.empty(part.rows, Cc, 2, device=x_b1.device, dtype=torch.float32)
    hip.lib().stem_conv_fwd(_dt(y), x_b1.data_ptr(), B, D, H, W, Cc, k, mp, bshift, fd, fh, fw, w.data_ptr(), _p(bias),
                            y.data_ptr(), part.t.data_ptr() if part else None, *_al(mask), _ROWS_ADDR if part else None, _stream())
    if part is not None:
        part.rows = _ROWS_OUT.value
    return (y, part) if want_partials else y


def stem_conv_wgrad(x_b1: torch.Tensor, dy: torch.Tensor, ksize: int, mask: Optional[MaskInfo], bshift: int,
                    dw_accum: torch.Tensor, db_accum: Optional[torch.Tensor], deterministic: Optional[bool] = None):
    B, D, H, W, Cc = dy.shape
    mp, fd, fh, fw = _mk(mask)
    ws = _det_workspace(dy.device, 1024 * Cc * (ksize ** 3 + 1) // 8) if (_det(deterministic) and dy.dtype == torch.bfloat16) else None
    hip.lib().stem_conv_wgrad(_dt(dy), x_b1.data_ptr(), dy.data_ptr(), B, D, H, W, Cc, ksize, mp, bshift, fd, fh, fw,
                              dw_accum.data_ptr(), _p(db_accum), *_al(mask), _p(ws), ws.numel() if ws is not None else 0, _stream())


# ------------------------------------------------------------------ norms
NREP = 8   # AM_NREP
DXREP = 64  # AM_DXREP


class NormStats:
    """Per-channel statistics + folded coefficients of one norm instance (all fp32, length C)."""

    def __init__(self, C: int, device):
        self.sums = torch.empty(NREP, C, 2, device=device, dtype=torch.float64)
        self.nrep = NREP
        self.buf = torch.empty(4, C, device=device, dtype=torch.float32)
        self.mean, self.rstd, self.scale, self.shift = self.buf[0], self.buf[1], self.buf[2], self.buf[3]
        self.count_ptr: Optional[torch.Tensor] = None
        self.count_host: float = 0.0
        self.sync_world: int = 1          # > 1: SyncBatchNorm statistics (the backward all-reduces its two sums as well)


def chan_stats(x: torch.Tensor, mask: Optional[MaskInfo], bshift: int, st: NormStats):
    B, D, H, W, Cc = x.shape
    mp, fd, fh, fw = _mk(mask)
    hip.lib().chan_stats(_dt(x), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw, st.sums.data_ptr(), *_al(mask), _stream())
    st.nrep = NREP


def mask_count(mask: MaskInfo, voxels_per_patch: int, out: torch.Tensor):
    hip.lib().mask_count(mask.t.data_ptr(), mask.t.numel(), voxels_per_patch, out.data_ptr(), _stream())


def norm_finalize(st: NormStats, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                  run_mean: Optional[torch.Tensor] = None, run_var: Optional[torch.Tensor] = None, momentum: float = 0.1):
    Cc = gamma.numel()
    hip.lib().norm_finalize(st.sums.data_ptr(), st.nrep, _p(st.count_ptr), float(st.count_host), Cc, gamma.data_ptr(), beta.data_ptr(),
                            eps, st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                            _p(run_mean), _p(run_var), momentum, _stream())


def norm_fold_running(st: NormStats, gamma, beta, run_mean, run_var, eps: float):
    hip.lib().norm_fold_running(gamma.numel(), gamma.data_ptr(), beta.data_ptr(), run_mean.data_ptr(), run_var.data_ptr(), eps,
                                st.scale.data_ptr(), st.shift.data_ptr(), _stream())


def norm_apply(x: torch.Tensor, st: NormStats, act: int, mask: Optional[MaskInfo] = None, bshift: int = 0,
               res: Optional[torch.Tensor] = None, stem: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
               fill: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, D, H, W, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    mp, fd, fh, fw = _mk(mask)
    sx, sw, sb = stem if stem is not None else (None, None, None)
    hip.lib().norm_apply(_dt(x), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw, st.scale.data_ptr(), st.shift.data_ptr(),
                         act, _p(res), _p(sx), _p(sw), _p(sb), _p(fill), out.data_ptr(), *_al(mask), _stream())
    return out


class NormBwdScratch:
    def __init__(self, C: int, device):
        self.k = torch.empty(3, C, device=device, dtype=torch.float32)


def norm_backward(dout: torch.Tensor, out: Optional[torch.Tensor], x: torch.Tensor, st: NormStats, gamma: torch.Tensor, act: int,
                  mask: Optional[MaskInfo], bshift: int, dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor],
                  dtoken: Optional[torch.Tensor] = None, fill: bool = False, dx: Optional[torch.Tensor] = None,
                  dres: Optional[torch.Tensor] = None, scratch: Optional[NormBwdScratch] = None,
                  dbeta2: Optional[torch.Tensor] = None, dxsum: Optional[torch.Tensor] = None,
                  reduced: Optional[ConvPartials] = None) -> torch.Tensor:
    """Backward of y = act(norm(x) [+res]) (or the densify fill).  Returns dx; accumulates dgamma/dbeta/dtoken.
    out=None with an activation (layers WITHOUT a residual): the derivative comes from the recomputed x*st.scale + st.shift.
    reduced: the partial rows of the conv3d(norm_bwd=(x, st, act)) launch that produced dout -- the reduce pass is skipped."""
    B, D, H, W, Cc = x.shape
    sc = scratch or NormBwdScratch(Cc, x.device)
    mp, fd, fh, fw = _mk(mask)
    L = hip.lib()
    s = _stream()
    ws_b, ws_x = _bwd_workspaces(x.device, Cc)
    if reduced is not None:
        assert st.sync_world <= 1 and out is None and dtoken is None and not fill
        reduced.finalize_bwd(st, gamma, sc, dgamma, dbeta, dbeta2)
        if dx is None:
            dx = torch.empty_like(x)
        L.norm_bwd_apply(_dt(x), dout.data_ptr(), None, x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                         st.mean.data_ptr(), st.rstd.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), act,
                         dx.data_ptr(), _p(dres), _p(dxsum), ws_x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), *_al(mask), 1, s)
        return dx
    if st.sync_world > 1:
        # SyncBatchNorm backward (torch.nn.SyncBatchNorm): dx needs the sums over the GLOBAL batch, the affine gradients stay LOCAL
        # sums (DDP then averages them like every other gradient): reduce -> finalize (param grads) -> all-reduce -> finalize (k's)
        import torch.distributed as dist
        L.norm_bwd_reduce(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                          st.mean.data_ptr(), st.rstd.data_ptr(), act, int(fill), ws_b.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                          *_al(mask), None, 0.0, None, None, None, None, None, None, None, None, s)
        L.norm_bwd_finalize(ws_b.data_ptr(), None, float(st.count_host), Cc, gamma.data_ptr(), st.rstd.data_ptr(),
                            sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), _p(dgamma), _p(dbeta), _p(dtoken), _p(dbeta2), s)
        dist.all_reduce(ws_b[:NREP * Cc * 3])
        L.norm_bwd_finalize(ws_b.data_ptr(), None, float(st.count_host), Cc, gamma.data_ptr(), st.rstd.data_ptr(),
                            sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), None, None, None, None, s)
        ws_b[:NREP * Cc * 3].zero_()                                   # the workspace contract: left zero
        if dx is None:
            dx = torch.empty_like(x)
        L.norm_bwd_apply(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                         st.mean.data_ptr(), st.rstd.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), act,
                         dx.data_ptr(), _p(dres), _p(dxsum), ws_x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), *_al(mask), 1, s)
        return dx
    if not FUSED_BWD_TAILS:                      # (tools/step_ab.py: the three-launch form, for same-process A/B timing)
        L.norm_bwd_reduce(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                          st.mean.data_ptr(), st.rstd.data_ptr(), act, int(fill), ws_b.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                          *_al(mask), None, 0.0, None, None, None, None, None, None, None, None, s)
        L.norm_bwd_finalize(ws_b.data_ptr(), _p(st.count_ptr), float(st.count_host), Cc, gamma.data_ptr(), st.rstd.data_ptr(),
                            sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), _p(dgamma), _p(dbeta), _p(dtoken), _p(dbeta2), s)
        ws_b[:NREP * Cc * 3].zero_()                                   # the shared workspaces' contract: zero on entry, LEFT zero
        if dx is None:
            dx = torch.empty_like(x)
        L.norm_bwd_apply(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                         st.mean.data_ptr(), st.rstd.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), act,
                         dx.data_ptr(), _p(dres), _p(dxsum), ws_x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), *_al(mask), 0, s)
        if dxsum is not None:
            ws_x[:DXREP * Cc].zero_()
        return dx
    # reduce + finalize in ONE launch (the last workgroup folds the sums into k0/k1/k2 and the parameter gradients)
    L.norm_bwd_reduce(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                      st.mean.data_ptr(), st.rstd.data_ptr(), act, int(fill), ws_b.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                      *_al(mask), _p(st.count_ptr), float(st.count_host), gamma.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(),
                      sc.k[2].data_ptr(), _p(dgamma), _p(dbeta), _p(dtoken), _p(dbeta2), s)
    if dx is None:
        dx = torch.empty_like(x)
    L.norm_bwd_apply(_dt(x), dout.data_ptr(), _p(out), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw,
                     st.mean.data_ptr(), st.rstd.data_ptr(), sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), act,
                     dx.data_ptr(), _p(dres), _p(dxsum), ws_x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), *_al(mask), 1, s)
    return dx


_BWS = {}
FUSED_BWD_TAILS = True
# bf16: the norm-backward reduce of a data gradient rides in the dgrad's epilogue (am_conv3d_nbred).  Measured neutral on the STUNet-B
# step (143.0 vs 142.7-143.3 ms, tools/step_ab.py ops.FUSED_NORM_BWD_REDUCE=1,0; profiles/r02_experiments.md), so it stays off.
FUSED_NORM_BWD_REDUCE = False


def _bwd_workspaces(device, C: int):
    """zero-initialised accumulators of the fused backward tails (left zero by every call), one pair per (device, stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _BWS.get(key)
    if ws is None or ws[2] < C:
        cap = max(C, 2048)
        ws = _BWS[key] = (torch.zeros(NREP * cap * 3 + 2, device=device, dtype=torch.float64),
                          torch.zeros(DXREP * cap + 2, device=device, dtype=torch.float64), cap)
    return ws[0], ws[1]


def chan_sum(x: torch.Tensor, mask: Optional[MaskInfo], bshift: int, out_accum: torch.Tensor):
    B, D, H, W, Cc = x.shape
    mp, fd, fh, fw = _mk(mask)
    hip.lib().chan_sum(_dt(x), x.data_ptr(), B, D, H, W, Cc, mp, bshift, fd, fh, fw, out_accum.data_ptr(), _stream())


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(a)
    hip.lib().add(_dt(a), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


# ------------------------------------------------------------------ proj / loss / sampler / optimizer
def proj_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, pre: Optional["NormStats"] = None) -> torch.Tensor:
    """pre: x is the INPUT of the (train-mode) BatchNorm `pre`; rec = proj(x * pre.scale + pre.shift), the BatchNorm output is never written."""
    B, D, H, W, Cc = x.shape
    rec = torch.empty(B, D, H, W, device=x.device, dtype=torch.float32)
    hip.lib().proj_fwd(_dt(x), x.data_ptr(), B * D * H * W, Cc, w.data_ptr(), b.data_ptr(), _p(pre.scale) if pre else None,
                       _p(pre.shift) if pre else None, rec.data_ptr(), _stream())
    return rec


def proj_norm_bwd(x: torch.Tensor, st: "NormStats", drec: torch.Tensor, w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                  dgamma: torch.Tensor, dbeta: torch.Tensor, dw_accum: torch.Tensor, db_accum: torch.Tensor) -> torch.Tensor:
    """backward of proj_fwd(x, w, b, pre=st) through the projection AND the BatchNorm (am_proj_norm_bwd): returns d loss / d x,
    accumulates the four parameter gradients."""
    B, D, H, W, Cc = x.shape
    assert st.sync_world <= 1
    dx = torch.empty_like(x)
    ws_b, _ = _bwd_workspaces(x.device, Cc)
    coef = torch.empty(3, Cc, device=x.device, dtype=torch.float32)
    hip.lib().proj_norm_bwd(_dt(x), x.data_ptr(), drec.data_ptr(), B * D * H * W, Cc, w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                            st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(), ws_b.data_ptr(), coef.data_ptr(), dx.data_ptr(),
                            dgamma.data_ptr(), dbeta.data_ptr(), dw_accum.data_ptr(), db_accum.data_ptr(), _stream())
    return dx


def proj_bwd(x: torch.Tensor, drec: torch.Tensor, w: torch.Tensor, dw_accum: torch.Tensor, db_accum: torch.Tensor,
             deterministic: Optional[bool] = None) -> torch.Tensor:
    B, D, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    ws = _det_workspace(x.device, 4096 * (Cc + 1) // 8) if _det(deterministic) else None
    hip.lib().proj_bwd(_dt(x), x.data_ptr(), drec.data_ptr(), B * D * H * W, Cc, w.data_ptr(), dx.data_ptr(), dw_accum.data_ptr(),
                       db_accum.data_ptr(), _p(ws), ws.numel() if ws is not None else 0, _stream())
    return dx


def head_stencil_supported(x: torch.Tensor) -> bool:
    """am_head_stencil serves this channels-last tensor (16^3 patches of a [B,D,H,W,C] map, C in {32, 64, 128, 192})."""
    B, D, H, W, C = x.shape
    ok = hip.lib()._lib.am_head_stencil_supported(_dt(x), C) == 1
    return bool(ok and D % 16 == 0 and H % 16 == 0 and W % 16 == 0 and max(B, D // 16, H // 16, W // 16) <= 255)


def head_fold(w2: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, wproj: torch.Tensor, bproj: torch.Tensor):
    """conv3x3x3 (Cmid, Cin, 3, 3, 3; no bias) -> eval BatchNorm (scale, shift) -> 1x1 projection, folded into (W_eff [27, Cin], b_eff [1])."""
    cmid, cin = w2.shape[0], w2.shape[1]
    assert tuple(w2.shape[2:]) == (3, 3, 3) and w2.is_contiguous() and wproj.numel() == cmid
    weff = torch.empty(27, cin, device=w2.device, dtype=torch.float32)
    beff = torch.empty(1, device=w2.device, dtype=torch.float32)
    hip.lib().head_fold(w2.data_ptr(), cmid, cin, scale.data_ptr(), shift.data_ptr(), wproj.data_ptr(), bproj.data_ptr(), weff.data_ptr(),
                        beff.data_ptr(), _stream())
    return weff, beff


def head_stencil(r: torch.Tensor, weff: torch.Tensor, beff: torch.Tensor, patches: MaskInfo, rec: Optional[torch.Tensor] = None,
                 inp: Optional[torch.Tensor] = None, l2: Optional[torch.Tensor] = None):
    """rec[q] = b_eff + sum_t W_eff[t] . r[q + t] on the ACTIVE 16^3 patches of `patches` (fp32 [B,D,H,W], other patches untouched) and / or
    their raw per-patch l2 against inp (fp32 [B, L] entries of those patches)."""
    B, D, H, W, C = r.shape
    lst, n = patches.active_list()
    assert lst is not None and (patches.fd, patches.fh, patches.fw) == (D // 16, H // 16, W // 16) and r.is_contiguous()
    assert rec is None or (rec.dtype == torch.float32 and tuple(rec.shape) == (B, D, H, W) and rec.is_contiguous())
    assert l2 is None or (inp is not None and l2.dtype == torch.float32 and l2.numel() == B * patches.fd * patches.fh * patches.fw)
    hip.lib().head_stencil(_dt(r), r.data_ptr(), B, D, H, W, C, weff.data_ptr(), beff.data_ptr(), lst.data_ptr(), n, _p(rec), _p(inp), _p(l2), _stream())


def patch_loss_fwd(inp: torch.Tensor, rec: torch.Tensor, mask: MaskInfo, normalized: bool, want_loss: bool = True):
    """inp/rec fp32 [B,D,H,W] -> (l2m [B,L], pmean, prstd, lossinfo[2] or None)."""
    B, D, H, W = inp.shape
    L = mask.fd * mask.fh * mask.fw
    l2m = torch.empty(B, L, device=inp.device, dtype=torch.float32)
    # want_loss=False (teacher pass): pmean/prstd stay NULL, which is also the kernel's cue to skip the visible patches entirely
    # (their l2 is 0 by definition and their rec voxels may never have been written, engine.forward(needed_patches=...))
    pm = torch.empty(B, L, device=inp.device, dtype=torch.float32) if want_loss else None
    pr = torch.empty(B, L, device=inp.device, dtype=torch.float32) if want_loss else None
    info = torch.empty(2, device=inp.device, dtype=torch.float32) if want_loss else None
    hip.lib().patch_loss_fwd(inp.data_ptr(), rec.data_ptr(), mask.t.data_ptr(), B, D, H, W, int(normalized), l2m.data_ptr(),
                             _p(pm), _p(pr), _p(info), _stream())
    return l2m, pm, pr, info


def patch_loss_bwd(inp, rec, mask: MaskInfo, pm, pr, info, gout: Optional[torch.Tensor]) -> torch.Tensor:
    B, D, H, W = inp.shape
    drec = torch.empty_like(rec)
    hip.lib().patch_loss_bwd(inp.data_ptr(), rec.data_ptr(), mask.t.data_ptr(), B, D, H, W, pm.data_ptr(), pr.data_ptr(),
                             info.data_ptr(), _p(gout), drec.data_ptr(), _stream())
    return drec


def mask_sampler(loss: torch.Tensor, keys: torch.Tensor, len_keep: int, len_loss: int) -> torch.Tensor:
    B, L = loss.shape
    out = torch.empty(B, L, device=loss.device, dtype=torch.uint8)
    hip.lib().mask_sampler(loss.data_ptr(), keys.data_ptr(), B, L, len_keep, len_loss, out.data_ptr(), _stream())
    return out


def sumsq(g: torch.Tensor, out: torch.Tensor):
    hip.lib().sumsq(g.data_ptr(), g.numel(), out.data_ptr(), _stream())


def adamw_ema(p, g, m, v, ema, n, lr, betas, eps, wd, step, sumsq_t, max_norm, ema_decay, gnorm_out, grad_scale: float = 1.0, dyn=None,
              guard: Optional[torch.Tensor] = None):
    """guard: device int32[4] {latched, first bad call, calls, 0} -- the per-step non-finite stop (am_adamw_ema)."""
    assert guard is None or (guard.dtype == torch.int32 and guard.numel() >= 4)
    hip.lib().adamw_ema(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(ema), n, lr, betas[0], betas[1], eps, wd, step,
                        _p(sumsq_t), max_norm, ema_decay, grad_scale, _p(gnorm_out), _p(dyn), _p(guard), _stream())


def adam_dyn_scalars(lr: float, betas, step: int, ema_decay: float):
    """the float[4] am_adamw_ema reads from device memory in graph replays (same roundings as its by-value path)."""
    import math
    return [float(lr), 1.0 - betas[0] ** step, math.sqrt(1.0 - betas[1] ** step), float(ema_decay)]


def ema(ema_t: torch.Tensor, p: torch.Tensor, decay: float, guard: Optional[torch.Tensor] = None):
    hip.lib().ema(ema_t.data_ptr(), p.data_ptr(), p.numel(), decay, _p(guard), _stream())


def ema_i64(ema_t: torch.Tensor, p: torch.Tensor, decay: float, guard: Optional[torch.Tensor] = None):
    """timm's ModelEma.update on int64 entries (num_batches_tracked), all of them in one launch."""
    assert ema_t.dtype == p.dtype == torch.int64 and ema_t.numel() == p.numel() and ema_t.is_contiguous() and p.is_contiguous()
    hip.lib().ema_i64(ema_t.data_ptr(), p.data_ptr(), p.numel(), decay, _p(guard), _stream())


def guard_restore(dst: torch.Tensor, snapshot: torch.Tensor, guard: torch.Tensor):
    """dst = snapshot where the non-finite guard is latched (device-side decision, no host round trip)."""
    assert dst.dtype == snapshot.dtype and dst.numel() == snapshot.numel() and dst.is_contiguous() and snapshot.is_contiguous()
    hip.lib().guard_restore(dst.data_ptr(), snapshot.data_ptr(), dst.numel() * dst.element_size(), guard.data_ptr(), _stream())


# ------------------------------------------------------------------ data-feed augmentation (device side)
def spline_prefilter(vol: torch.Tensor):
    """in place: cubic B-spline coefficients of an fp32 [D,H,W] volume (== scipy.ndimage.spline_filter(order=3, mode='mirror'))."""
    assert vol.dtype == torch.float32 and vol.dim() == 3 and vol.is_contiguous()
    hip.lib().spline_prefilter(vol.data_ptr(), *vol.shape, _stream())
    return vol


def resample_affine(src: torch.Tensor, dst: torch.Tensor, affine_3x4, order: int, cval: float = 0.0):
    """dst[o] = interp(src, A (o, 1)); src fp32 [Ds,Hs,Ws] (prefiltered when order == 3), dst fp32 [D,H,W]."""
    assert src.dtype == dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
    arr = (_ct.c_float * 12)(*[float(v) for v in affine_3x4])
    hip.lib().resample_affine(src.data_ptr(), *src.shape, dst.data_ptr(), *dst.shape, _ct.addressof(arr), int(order), float(cval), _stream())
    return dst
