"""GPU: every HIP op (through the C ABI) against the CPU oracle ops on the same seeded inputs.
Tolerances (stated per dtype):
  fp32 path (exact-f32 MFMA, fp32 storage):   <= 2e-4 of the tensor's max |value|  (reduction order only)
  bf16 path (bf16 storage, fp32 accumulate):  <= 2e-2 of max |value|, inputs pre-rounded to bf16 on both sides
Integer/bool outputs (sampler masks) are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import anatomask_oracle as O
from tests.helpers import load, tiny_cfg

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}


@pytest.fixture(scope="module")
def ops():
    from anatomask_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):            # round through the storage dtype so both sides see identical inputs
    return t.to(dtype).float()


def to_cl(t, dtype):        # NCDHW cpu fp32 -> channels-last device tensor
    return t.permute(0, 2, 3, 4, 1).contiguous().to(device=DEV, dtype=dtype)


def from_cl(t):             # channels-last device -> NCDHW cpu fp32
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def close(got, want, tol, what="", mask=None):
    got, want = got.double(), want.double()
    if mask is not None:                      # inactive voxels are don't-care (may hold stale/garbage bits)
        keep = mask.expand_as(want) > 0
        got, want = torch.where(keep, got, torch.zeros_like(got)), torch.where(keep, want, torch.zeros_like(want))
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def mk_mask(B, f, keep, seed=3):
    g = torch.Generator().manual_seed(seed)
    L = f[0] * f[1] * f[2]
    idx = torch.rand(B, L, generator=g).argsort(1)[:, :keep]
    return torch.zeros(B, L, dtype=torch.bool).scatter_(1, idx, True).view(B, 1, *f)


# ------------------------------------------------------------------ convolutions
CONV_CASES = [  # Cin, Cout, k, stride
    (16, 24, 3, 1), (40, 64, 3, 1), (32, 32, 3, 2), (16, 32, 1, 2), (64, 16, 1, 1), (8, 8, 3, 1), (96, 48, 3, 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(16, 24), (40, 64), (72, 32)])
@pytest.mark.parametrize("sparse", [False, True])
def test_conv_k3_wide_grid_h_run_variant(ops, dtype, case, sparse):
    """k3 s1 forward + data gradient on W >= 16 grids: the 4x4x16-brick kernel variant that shares fragment rows across h-runs of taps."""
    cin, cout = case
    B, f, bs = 2, (2, 2, 4), 2
    so = tuple(v << bs for v in f)               # (8, 8, 16)
    x = q(rnd(B, cin, *so, seed=31), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=32, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=33)
    dy = q(rnd(B, cout, *so, seed=34), dtype)
    mask = mk_mask(B, f, 9) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    mo = O.upsample_mask(mask, so).float() if sparse else None
    if sparse:
        x, dy = x * mo, dy * mo
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, bias, padding=1)
    if sparse:
        yr = yr * mo
    yr.backward(dy)
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=False)
    y = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, bias.to(DEV), so, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs)
    close(from_cl(y), yr.detach(), TOL[dtype], "conv fwd (h-run)", mo)
    wpd = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=True)
    dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, so, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs)
    close(from_cl(dx), xr.grad, TOL[dtype], "conv dgrad (h-run)", mo)


@pytest.mark.parametrize("case", [(32, 32, 3, 1), (64, 32, 3, 1), (32, 64, 3, 1), (32, 64, 3, 2), (32, 64, 1, 2), (24, 16, 3, 1)])
@pytest.mark.parametrize("sparse", [False, True])
def test_conv_wgrad_narrow_channels_wide_grid(ops, case, sparse):
    """W >= 16 grids with <= 32-channel operands take the 32-wide wgrad tiles (MI/NWX variants of conv_wgrad_kernel)."""
    cin, cout, k, s = case
    dtype = torch.bfloat16
    B, f, bs_out = 2, (2, 2, 8), 2
    so = tuple(v << bs_out for v in f)           # (8, 8, 32)
    si = tuple(v * s for v in so)
    x = q(rnd(B, cin, *si, seed=11), dtype)
    dy = q(rnd(B, cout, *so, seed=12), dtype)
    mask = mk_mask(B, f, 13) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    bs_in = bs_out + (1 if s == 2 else 0)
    if sparse:
        x = x * O.upsample_mask(mask, si).float()
        dy = dy * O.upsample_mask(mask, so).float()
    w = torch.zeros(cout, cin, k, k, k, requires_grad=True)
    F.conv3d(x, w, None, stride=s, padding=k // 2).backward(dy)
    dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out)
    dw = torch.zeros(cout, cin, k, k, k, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), w.grad, TOL[dtype], "conv wgrad (narrow channels)")


@pytest.mark.parametrize("case", [(64, 64, 3, 1, 3, (2, 2, 2)), (32, 32, 3, 1, 4, (1, 1, 2)), (64, 64, 3, 1, 2, (2, 4, 8)), (64, 96, 3, 1, 1, (4, 8, 8)),
                                  (32, 64, 3, 2, 3, (1, 2, 2)), (64, 128, 3, 2, 2, (2, 4, 4)), (128, 64, 3, 2, 1, (4, 4, 8)), (64, 64, 1, 2, 3, (1, 2, 2)),
                                  # gather form (am_conv3d_wgrad gather_workspace: 1- / 2-voxel patches, Cy % 128 == 0, Cx % 64 == 0): K-major copies of the
                                  # active voxels + plain GEMMs; odd grids, stride 1 and 2, one tile and several, k splits (atomics) and one writer (det)
                                  (64, 128, 3, 1, 1, (4, 8, 8)), (128, 128, 3, 1, 0, (6, 5, 7)), (64, 128, 3, 2, 1, (3, 4, 5)), (192, 256, 3, 2, 0, (5, 6, 4)),
                                  (256, 256, 3, 1, 0, (12, 12, 12)), (256, 128, 3, 2, 1, (3, 4, 5))])
@pytest.mark.parametrize("det", [False, True])
def test_conv_wgrad_block_sparse_ignores_inactive_voxels(ops, case, det):
    """Block-sparse weight gradients at the patch widths of the encoder levels (dY patches 16 / 8 / 4 / 2 / 1 voxels wide): bricks inside
    one patch are kept or skipped on ONE mask lookup (8-wide patches take 8x8 bricks for that), bricks that span patches test every
    staged row; the deepest levels' wide layers take the gather form.  The inactive voxels of X and dY hold NaN here: nothing may be read
    from them."""
    cin, cout, k, s, bs_out, f = case
    dtype = torch.bfloat16
    B = 2
    so = tuple(v << bs_out for v in f)
    si = tuple(v * s for v in so)
    bs_in = bs_out + (1 if s == 2 else 0)
    mask = mk_mask(B, f, max(1, (f[0] * f[1] * f[2] * 2) // 5), seed=41)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    mx, my = O.upsample_mask(mask, si).float(), O.upsample_mask(mask, so).float()
    x = q(rnd(B, cin, *si, seed=42), dtype) * mx
    dy = q(rnd(B, cout, *so, seed=43), dtype) * my
    w = torch.zeros(cout, cin, k, k, k, requires_grad=True)
    F.conv3d(x, w, None, stride=s, padding=k // 2).backward(dy)
    xd, dyd = to_cl(x, dtype), to_cl(dy, dtype)
    xd = torch.where(to_cl(mx.expand_as(x), dtype) > 0, xd, torch.full_like(xd, float("nan")))
    dyd = torch.where(to_cl(my.expand_as(dy), dtype) > 0, dyd, torch.full_like(dyd, float("nan")))
    ops.DETERMINISTIC_WGRAD = det
    try:
        dwp = ops.conv3d_wgrad(ops.CONV_FWD, xd, dyd, k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out)
    finally:
        ops.DETERMINISTIC_WGRAD = False
    dw = torch.zeros(cout, cin, k, k, k, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), w.grad, TOL[dtype], "block-sparse conv wgrad, NaN in the inactive voxels")


@pytest.mark.parametrize("case", [(64, 64, (5, 20, 40), 3), (96, 64, (9, 8, 16), 1), (64, 32, (7, 12, 24), 2), (40, 72, (33, 9, 17), 1), (96, 96, (4, 8, 32), 2), (128, 96, (3, 16, 16), 1)])
@pytest.mark.parametrize("det", [False, True])
def test_conv_wgrad_plane_bricks_ragged(ops, case, det):
    """Dense k3 s1 bf16 weight gradients walk one-plane 1x8x16 bricks d-fastest, columns interleaved over the slots of an XCD
    (conv_wgrad.hip `walk 1`): ragged extents (partial bricks in every dimension, column counts that do not divide by 8, more
    segments than slots) must still visit every brick exactly once."""
    cin, cout, sp, B = case
    dtype = torch.bfloat16
    x = q(rnd(B, cin, *sp, seed=21), dtype)
    dy = q(rnd(B, cout, *sp, seed=22), dtype)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, w, None, padding=1).backward(dy)
    ops.DETERMINISTIC_WGRAD = det
    try:
        dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), 3, 1)
    finally:
        ops.DETERMINISTIC_WGRAD = False
    dw = torch.zeros(cout, cin, 3, 3, 3, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), w.grad, TOL[dtype], "conv wgrad (plane bricks)")


@pytest.mark.parametrize("case", [(32, 64, (5, 9, 33), 2, 0), (64, 128, (6, 4, 16), 1, 1), (48, 72, (3, 21, 40), 3, 1), (128, 256, (4, 8, 16), 2, 0)])
@pytest.mark.parametrize("det", [False, True])
def test_conv_wgrad_stride2_full_resolution_bricks(ops, case, det):
    """Stride-2 k3 bf16 weight gradients on grids >= 16 wide stage the X brick at full resolution, one unit per d-tap
    (conv_wgrad_kernel<..., S2>): ragged extents, even (2n) and odd (2n - 1) input sizes, channel counts that are not tile multiples."""
    cin, cout, so, B, odd = case
    dtype = torch.bfloat16
    si = tuple(2 * v - odd for v in so)
    x = q(rnd(B, cin, *si, seed=31), dtype)
    dy = q(rnd(B, cout, *so, seed=32), dtype)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    y = F.conv3d(x, w, None, stride=2, padding=1)
    assert tuple(y.shape[2:]) == so
    y.backward(dy)
    ops.DETERMINISTIC_WGRAD = det
    try:
        dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), 3, 2)
    finally:
        ops.DETERMINISTIC_WGRAD = False
    dw = torch.zeros(cout, cin, 3, 3, 3, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), w.grad, TOL[dtype], "conv wgrad (stride 2, full-resolution bricks)")


@pytest.mark.parametrize("case", [(64, 32, "relu6", False), (64, 64, "lrelu", True), (128, 64, "none", False), (32, 32, "lrelu", True), (40, 72, "relu6", False)])
def test_conv_dgrad_with_fused_norm_backward_reduce(ops, case):
    """am_conv3d_nbred: a data-gradient launch whose output is the gradient wrt act(norm(x_pre)) leaves the norm-backward sums
    (sum g, sum g*x_pre per workgroup and channel) in its partial rows; norm_backward(..., reduced=rows) must give the same dx and
    affine gradients as the stand-alone reduce pass over the same tensors (bf16: the sums see g rounded to bf16)."""
    cdy, cdx, actn, sparse = case                        # forward conv cdx -> cdy; its dgrad produces a cdx-channel gradient
    act = {"relu6": ops.ACT_RELU6, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[actn]
    dtype = torch.bfloat16
    B, f, bs = 2, (2, 2, 4), 2
    so = tuple(v << bs for v in f)                       # (8, 8, 16)
    mask = mk_mask(B, f, 13) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    mo = O.upsample_mask(mask, so).float() if sparse else None
    w = q(rnd(cdy, cdx, 3, 3, 3, seed=2, scale=1.0 / np.sqrt(cdx * 27)), dtype)
    dy = q(rnd(B, cdy, *so, seed=4), dtype)
    xp = q(rnd(B, cdx, *so, seed=5) * 1.5 + 0.7, dtype)  # the norm's input (a conv output): off-centre on purpose
    if sparse:
        dy, xp = dy * mo, xp * mo
    gam = (torch.rand(cdx, generator=torch.Generator().manual_seed(6)) + 0.5).to(DEV)
    bet = (torch.rand(cdx, generator=torch.Generator().manual_seed(7)) - 0.5).to(DEV)
    xd = to_cl(xp, dtype)
    st = ops.NormStats(cdx, DEV)
    if sparse:
        cnt = torch.zeros(1, device=DEV, dtype=torch.float64); ops.mask_count(mi, (1 << bs) ** 3, cnt); st.count_ptr = cnt
    else:
        st.count_host = float(B * so[0] * so[1] * so[2])
    ops.chan_stats(xd, mi, bs if sparse else 0, st)
    ops.norm_finalize(st, gam, bet, 1e-5)
    wpd = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=True)
    kw = dict(in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs) if sparse else {}
    # reference route: plain dgrad, then the stand-alone reduce + apply
    dz = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, so, 3, 1, **kw)
    dg0, db0 = torch.zeros(cdx, device=DEV), torch.zeros(cdx, device=DEV)
    dx0 = ops.norm_backward(dz, None, xd, st, gam, act, mi, bs if sparse else 0, dg0, db0)
    # fused route
    dz1, rows = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, so, 3, 1, norm_bwd=(xd, st, act), **kw)
    if sparse:                                           # (bricks without an active voxel are never written)
        keep = to_cl(mo.expand(B, cdx, *so), dtype) > 0
        assert torch.equal(torch.where(keep, dz1, torch.zeros_like(dz1)), torch.where(keep, dz, torch.zeros_like(dz)))
    else:
        assert torch.equal(dz1, dz)
    dg1, db1 = torch.zeros(cdx, device=DEV), torch.zeros(cdx, device=DEV)
    dx1 = ops.norm_backward(dz1, None, xd, st, gam, act, mi, bs if sparse else 0, dg1, db1, reduced=rows)
    close(db1.cpu(), db0.cpu(), 4e-3, "dbeta (fused reduce)")
    close(dg1.cpu(), dg0.cpu(), 4e-3, "dgamma (fused reduce)")
    close(from_cl(dx1), from_cl(dx0), 1e-2, "dx (fused reduce)", mo)


@pytest.mark.parametrize("case", [(64, 64, 3, 1, False), (32, 32, 3, 1, True), (32, 64, 3, 2, True), (16, 24, 1, 2, False), (64, 64, 4, 2, False)])
def test_conv_wgrad_deterministic_mode(ops, case):
    """det_workspace: per-slot partial sums folded in slot order instead of fp32 atomics -> bit-identical from run to run (the
    atomics path differs in the last bits between runs on large grids), and equal to the atomics result up to fp32 summation order.
    Covers the k-split waves (32-channel tiles), strided units (several launches per call) and the transposed convolution."""
    cin, cout, k, s, sparse = case
    dtype = torch.bfloat16
    B, f, bs_out = 2, (2, 2, 4), 2
    convt = k == 4
    so = tuple(v << bs_out for v in f)                       # conv output / convT input grid (8, 8, 16)
    si = tuple(v * s for v in so)
    mask = mk_mask(B, f, 13) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    if convt:                                                # x = convT input (coarse), dy = gradient of its output (fine)
        x = to_cl(q(rnd(B, cin, *so, seed=11), dtype), dtype)
        dy = to_cl(q(rnd(B, cout, *si, seed=12), dtype), dtype)
        run = lambda: ops.conv3d_wgrad(ops.CONVT_FWD, x, dy, 4, 2)
    else:
        xs = q(rnd(B, cin, *si, seed=11), dtype)
        dys = q(rnd(B, cout, *so, seed=12), dtype)
        if sparse:
            xs = xs * O.upsample_mask(mask, si).float()
            dys = dys * O.upsample_mask(mask, so).float()
        x, dy = to_cl(xs, dtype), to_cl(dys, dtype)
        bs_in = bs_out + (1 if s == 2 else 0)
        run = lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out)
    ref = run()                                              # atomics
    ops.DETERMINISTIC_WGRAD = True
    try:
        a, b = run(), run()
        ops._DET_WS.clear()                                  # a workspace with stale contents must not matter either
        torch.empty(80 << 20, device=DEV).fill_(float("nan"))
        c = run()
    finally:
        ops.DETERMINISTIC_WGRAD = False
    assert torch.equal(a, b) and torch.equal(a, c)
    close(a.cpu(), ref.cpu(), 1e-5, "deterministic vs atomic wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("sparse", [False, True])
def test_conv_fwd_dgrad_wgrad(ops, dtype, case, sparse):
    cin, cout, k, s = case
    B, f = 2, (2, 3, 2)
    bs_out = 2                                   # output blocks are 4 voxels wide
    so = tuple(v << bs_out for v in f)           # (8, 12, 8)
    si = tuple(v * s for v in so)
    x = q(rnd(B, cin, *si, seed=1), dtype)
    w = q(rnd(cout, cin, k, k, k, seed=2, scale=1.0 / np.sqrt(cin * k ** 3)), dtype)
    bias = rnd(cout, seed=3)
    dy = q(rnd(B, cout, *so, seed=4), dtype)
    mask = mk_mask(B, f, 5) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    bs_in = bs_out + (1 if s == 2 else 0)
    if sparse:
        x = x * O.upsample_mask(mask, si).float()
        dy = dy * O.upsample_mask(mask, so).float()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, bias, stride=s, padding=k // 2)
    mo = O.upsample_mask(mask, so).float() if sparse else None
    if sparse:
        yr = yr * mo
    yr.backward(dy)
    # forward
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=False)
    y = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, bias.to(DEV), so, k, s, in_mask=mi, in_bshift=bs_in, out_mask=mi, out_bshift=bs_out)
    close(from_cl(y), yr.detach(), TOL[dtype], "conv fwd", mo)
    # data gradient (written only / zero outside active input blocks)
    wpd = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=True)
    dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, si, k, s, in_mask=mi, in_bshift=bs_out, out_mask=mi, out_bshift=bs_in)
    mi_in = O.upsample_mask(mask, si).float() if sparse else None
    close(from_cl(dx), xr.grad, TOL[dtype], "conv dgrad", mi_in)
    # accumulate flag
    dx2 = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, si, k, s, in_mask=mi, in_bshift=bs_out, out_mask=mi, out_bshift=bs_in,
                     out=dx.clone(), accumulate=True)
    close(from_cl(dx2), 2 * xr.grad, 2 * TOL[dtype], "conv dgrad accumulate", mi_in)
    # weight gradient
    dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out)
    dw = torch.zeros_like(w, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), wr.grad, TOL[dtype], "conv wgrad")


# ------------------------------------------------------------------ fp32 storage, split products (AM_DT_F32S, round 4)
@pytest.mark.parametrize("case", [(32, 64, 3, 1), (64, 32, 3, 2), (40, 72, 1, 2), (16, 24, 3, 1), (64, 64, 4, 2)])
@pytest.mark.parametrize("sparse", [False, True])
def test_conv_f32_split_products(ops, case, sparse):
    """f32_split=True (AM_DT_F32S): fp32 tensors, channel contraction on the bf16 matrix cores from hi / lo splits of BOTH operands (hi hi + hi lo +
    lo hi + lo lo forward / data gradient; the three leading terms in the weight gradient), fp32 accumulation.  Against F.conv3d /
    F.conv_transpose3d autograd in fp32 on inputs that are NOT bf16-representable: <= 3e-5 of the tensor's max (the exact mode's bound is
    2e-4 for reduction order alone; plain bf16 operands would be at 4e-3)."""
    cin, cout, k, s = case
    convt = k == 4
    if convt and sparse:
        pytest.skip("transposed convolutions are dense (the decoder)")
    B, f, bs_out = 2, (2, 3, 2), 2
    so = tuple(v << bs_out for v in f)
    si = tuple(v * s for v in so)
    dt, tol = torch.float32, 3e-5
    mask = mk_mask(B, f, 5) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    mo = O.upsample_mask(mask, so).float() if sparse else None
    mi_in = O.upsample_mask(mask, si).float() if sparse else None
    if True:
        if convt:                                        # ConvTranspose3d k4 s2 p1: coarse `so` -> fine 2 * so
            fine = tuple(2 * v for v in so)
            x = rnd(B, cin, *so, seed=1)
            w = rnd(cin, cout, 4, 4, 4, seed=2, scale=1.0 / np.sqrt(cin * 8))
            dy = rnd(B, cout, *fine, seed=4)
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            yr = F.conv_transpose3d(xr, wr, None, stride=2, padding=1)
            yr.backward(dy)
            y = ops.conv3d(ops.CONVT_FWD, to_cl(x, dt), ops.pack_weight(w.to(DEV), dt, True, False, f32_split=True), None, fine, 4, 2)
            close(from_cl(y), yr.detach(), tol, "convT fwd (split)")
            dx = ops.conv3d(ops.CONVT_DGRAD, to_cl(dy, dt), ops.pack_weight(w.to(DEV), dt, True, True, f32_split=True), None, so, 4, 2)
            close(from_cl(dx), xr.grad, tol, "convT dgrad (split)")
            dwp = ops.conv3d_wgrad(ops.CONVT_FWD, to_cl(x, dt), to_cl(dy, dt), 4, 2, f32_split=True)
            dw = torch.zeros_like(w, device=DEV)
            ops.unpack_grad(dwp, dw, transposed_conv=True, accumulate=False)
            close(dw.cpu(), wr.grad, tol, "convT wgrad (split)")
            return
        x = rnd(B, cin, *si, seed=1)
        w = rnd(cout, cin, k, k, k, seed=2, scale=1.0 / np.sqrt(cin * k ** 3))
        bias = rnd(cout, seed=3)
        dy = rnd(B, cout, *so, seed=4)
        if sparse:
            x, dy = x * mi_in, dy * mo
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv3d(xr, wr, bias, stride=s, padding=k // 2)
        if sparse:
            yr = yr * mo
        yr.backward(dy)
        bs_in = bs_out + (1 if s == 2 else 0)
        y = ops.conv3d(ops.CONV_FWD, to_cl(x, dt), ops.pack_weight(w.to(DEV), dt, False, False, f32_split=True), bias.to(DEV), so, k, s,
                       in_mask=mi, in_bshift=bs_in, out_mask=mi, out_bshift=bs_out)
        close(from_cl(y), yr.detach(), tol, "conv fwd (split)", mo)
        dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dt), ops.pack_weight(w.to(DEV), dt, False, True, f32_split=True), None, si, k, s,
                        in_mask=mi, in_bshift=bs_out, out_mask=mi, out_bshift=bs_in)
        close(from_cl(dx), xr.grad, tol, "conv dgrad (split)", mi_in)
        for det in (False, True):
            ops.DETERMINISTIC_WGRAD = det
            try:
                dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dt), to_cl(dy, dt), k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out, f32_split=True)
            finally:
                ops.DETERMINISTIC_WGRAD = False
            dw = torch.zeros_like(w, device=DEV)
            ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
            close(dw.cpu(), wr.grad, tol, f"conv wgrad (split, det={det})")


def test_split_bf16_planes(ops):
    """am_split_bf16: hi = bf16(x) (round to nearest even), lo = bf16(x - hi); x - hi - lo is below 2^-16 |x|."""
    x = (rnd(4, 8, 8, 16, 24, seed=5) * 37.0).to(DEV)
    hi, lo = ops.split_bf16(x)
    assert hi.dtype == torch.bfloat16 and lo.dtype == torch.bfloat16 and hi.shape == x.shape
    assert torch.equal(hi, x.to(torch.bfloat16))
    assert torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    assert ((x - hi.float() - lo.float()).abs() <= 2.0 ** -16 * x.abs() + 1e-30).all()


# ------------------------------------------------------------------ persistent LDS-DMA kernel (conv_k3.hip, round 4)
# Dense k3 s1 forward / data gradient in bf16 with >= 256 units (8x4x16 bricks x 64-channel tiles: one per CU) take the 8-wave persistent kernel:
# P/decoder3D.py:20-22 (the decoder's conv3x3x3 pairs) at training sizes.  Cases cover one / two / three / four channel slabs, one / two /
# three channel tiles (the grid is a multiple of the tile count: 255 workgroups for three), a unit count that is not a multiple of the grid,
# volume borders on every face, bias, the statistics rows and the fused store epilogue.
# (64, 32) / (128, 32): the 32-channel output tile (round 5: NS = 2 instantiations, the decoder's last conv C -> C / 2)
K3_CASES = [(64, 64, (64, 64, 64), 1), (32, 64, (32, 64, 128), 1), (128, 128, (32, 32, 64), 2), (64, 64, (64, 64, 80), 1), (96, 192, (40, 48, 48), 1),
            (64, 32, (32, 32, 64), 2), (128, 32, (24, 36, 48), 4), (192, 96, (24, 32, 32), 2)]   # (96: three 32-channel tiles, STUNet-H's 192 -> 96)


@pytest.mark.parametrize("case", K3_CASES)
def test_conv_k3_persistent_kernel_fwd_dgrad_stats(ops, case):
    cin, cout, S, B = case
    dtype = torch.bfloat16
    x = q(rnd(B, cin, *S, seed=71), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=72, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=73)
    dy = q(rnd(B, cout, *S, seed=74), dtype)
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, bias, padding=1)
    yr.backward(dy)
    wd = w.to(DEV)
    y, part = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), ops.pack_weight(wd, dtype, False, False), bias.to(DEV), S, 3, 1, want_partials=True)
    ny = cout // 32 if cout % 64 else cout // 64
    units = B * (S[0] // 8) * (S[1] // 4) * (S[2] // 16) * ny
    assert units >= 256 and part.rows == 8 * (256 - 256 % ny), "the launch must have taken conv_k3_kernel (8 rows per workgroup)"
    close(from_cl(y), yr.detach(), TOL[dtype], "conv_k3 fwd")
    # statistics rows == a separate pass over the stored output
    st_a, st_b = ops.NormStats(cout, DEV), ops.NormStats(cout, DEV)
    part.reduce(sums=st_a.sums)
    ops.chan_stats(y, None, 0, st_b)
    ref = st_b.sums.cpu().sum(0)
    assert (st_a.sums.cpu()[0] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # data gradient (the weight tap index is mirrored, the packed weight transposed)
    dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), ops.pack_weight(wd, dtype, False, True), None, S, 3, 1)
    close(from_cl(dx), xr.grad, TOL[dtype], "conv_k3 dgrad")
    # ... with the SUM-ONLY statistics epilogue (AM_CONV_PARTIALS_SUM_ONLY: what the decoder's backward asks for -- the transposed conv's bias
    # gradient is the per-channel sum of this data gradient): same output bits, same sums as the full epilogue's, sum-of-squares column zero
    if cin % 64 == 0:
        dx2, p2 = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), ops.pack_weight(wd, dtype, False, True), None, S, 3, 1, want_partials=True, partials_sum_only=True)
        dx3, p3 = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), ops.pack_weight(wd, dtype, False, True), None, S, 3, 1, want_partials=True)
        assert torch.equal(dx2, dx) and torch.equal(dx3, dx) and p2.rows == p3.rows
        assert torch.equal(p2.t[:p2.rows, :, 0], p3.t[:p3.rows, :, 0]) and float(p2.t[:p2.rows, :, 1].abs().max()) == 0.0 and float(p3.t[:p3.rows, :, 1].abs().max()) > 0.0
        acc = torch.zeros(cin, device=DEV)
        p2.finalize(None, sum_accum=acc)
        ref = from_cl(dx).double().sum(dim=(0, 2, 3, 4))
        assert (acc.cpu().double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-3


def test_conv_k3_sample_index_beyond_the_packed_field(ops):
    """conv_k3_kernel packs the sample index and the brick indices of a unit into 8-bit fields; B >= 256 (many small 16^3 maps: >= 256
    units, so the shape qualifies otherwise) must not alias the sample index into the channel-tile field -- such launches go to
    conv_igemm.hip.  Forward and data gradient of samples 0, 255, 256, 299 against F.conv3d; B = 255 still takes the persistent kernel."""
    dtype = torch.bfloat16
    cin, cout, S = 64, 64, (16, 16, 16)
    w = q(rnd(cout, cin, 3, 3, 3, seed=172, scale=1.0 / np.sqrt(cin * 27)), dtype)
    wd = w.to(DEV)
    for B, k3 in ((300, False), (255, True)):
        x = q(rnd(B, cin, *S, seed=171), dtype)
        y, part = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), ops.pack_weight(wd, dtype, False, False), None, S, 3, 1, want_partials=True)
        assert (part.rows == 8 * 256) == k3, "which kernel served the launch (conv_k3_kernel writes 8 rows per workgroup)"
        dy = q(rnd(B, cout, *S, seed=173), dtype)
        dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), ops.pack_weight(wd, dtype, False, True), None, S, 3, 1)
        pick = [0, 1, B // 2, 254, B - 1] + ([255, 256] if B > 256 else [])
        yr = F.conv3d(x[pick], w, None, padding=1)
        close(from_cl(y[pick]), yr, TOL[dtype], f"k3 fwd B={B}")
        dxr = F.conv_transpose3d(dy[pick], w, None, padding=1)     # data gradient of a k3 s1 conv = the transposed conv of dy
        close(from_cl(dx[pick]), dxr, TOL[dtype], f"k3 dgrad B={B}")


@pytest.mark.parametrize("cout", [64, 32, 96])
def test_conv_k3_persistent_kernel_fused_epilogue(ops, cout):
    """act(conv * scale + shift + res) in the store of the persistent kernel (the teacher's decoder convs at 128^3 / 64^3)."""
    dtype = torch.bfloat16
    B, cin, S = max(1, 64 // cout), 64, (64, 64, 64)
    x = q(rnd(B, cin, *S, seed=81), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=82, scale=1.0 / np.sqrt(cin * 27)), dtype)
    sc, sh = rnd(cout, seed=83).abs() + 0.5, rnd(cout, seed=84)
    res = q(rnd(B, cout, *S, seed=85), dtype)
    conv = F.conv3d(x, w, None, padding=1)
    wp = ops.pack_weight(w.to(DEV), dtype, False, False)
    bc = lambda t: t.view(1, -1, 1, 1, 1)
    y1 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_scale=sc.to(DEV), ep_shift=sh.to(DEV), ep_act=ops.ACT_RELU6)
    close(from_cl(y1), torch.clamp(conv * bc(sc) + bc(sh), 0, 6), TOL[dtype], "conv_k3 + bn + relu6")
    y2 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_scale=sc.to(DEV), ep_shift=sh.to(DEV), ep_res=to_cl(res, dtype))
    close(from_cl(y2), conv * bc(sc) + bc(sh) + res, TOL[dtype], "conv_k3 + bn + res")
    y3 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_res=to_cl(res, dtype), ep_act=ops.ACT_LRELU)
    close(from_cl(y3), F.leaky_relu(conv + res, 0.01), TOL[dtype], "conv_k3 + res + lrelu")


def test_conv_k3_persistent_kernel_is_deterministic(ops):
    """Two launches on the same inputs give the same bits, output and statistics rows (fixed unit -> workgroup map, one adder per row)."""
    dtype = torch.bfloat16
    x = to_cl(q(rnd(1, 64, 64, 64, 64, seed=91), dtype), dtype)
    wp = ops.pack_weight(q(rnd(64, 64, 3, 3, 3, seed=92, scale=0.03), dtype).to(DEV), dtype, False, False)
    y1, p1 = ops.conv3d(ops.CONV_FWD, x, wp, None, (64, 64, 64), 3, 1, want_partials=True)
    y2, p2 = ops.conv3d(ops.CONV_FWD, x, wp, None, (64, 64, 64), 3, 1, want_partials=True)
    assert torch.equal(y1, y2) and torch.equal(p1.t[:p1.rows], p2.t[:p2.rows])


@pytest.mark.parametrize("case", [(64, 64, (32, 32, 32), 4), (128, 64, (16, 32, 32), 8), (64, 128, (32, 16, 32), 4), (64, 64, (8, 4, 16), 255)])
def test_conv_transpose_persistent_kernel(ops, case):
    """ConvTranspose3d k4 s2 p1 (P/decoder3D.py:17) on the persistent 8-wave LDS-DMA kernel's transposed instantiation (conv_k3.hip, CT):
    the eight output parity classes of a coarse brick from ONE staged copy of it.  Against F.conv_transpose3d on the same bf16-rounded
    operands, bias included: two slabs (both resident for the eight classes), four slabs (re-fetched per class), two channel tiles, and a
    grid of single-brick samples (every brick is a border brick on all six faces)."""
    cin, cout, S, B = case
    dtype = torch.bfloat16
    x = q(rnd(B, cin, *S, seed=601), dtype)
    w = q(rnd(cin, cout, 4, 4, 4, seed=602, scale=1.0 / np.sqrt(cin * 8)), dtype)
    bias = rnd(cout, seed=603)
    fine = tuple(2 * v for v in S)
    ref = F.conv_transpose3d(x, w, bias, stride=2, padding=1)
    y = ops.conv3d(ops.CONVT_FWD, to_cl(x, dtype), ops.pack_weight(w.to(DEV), dtype, True, False), bias.to(DEV), fine, 4, 2)
    close(from_cl(y), ref, TOL[dtype], "convT (persistent kernel)")
    # every parity class separately (a wrong kernel index in one class hides behind the maximum of the others)
    got = from_cl(y)
    for pd in range(2):
        for ph in range(2):
            for pw in range(2):
                a_, b_ = got[:, :, pd::2, ph::2, pw::2], ref[:, :, pd::2, ph::2, pw::2]
                assert (a_ - b_).abs().max().item() <= TOL[dtype] * b_.abs().max().item(), f"class {pd}{ph}{pw}"
    # two launches: the same bits
    y2 = ops.conv3d(ops.CONVT_FWD, to_cl(x, dtype), ops.pack_weight(w.to(DEV), dtype, True, False), bias.to(DEV), fine, 4, 2)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("act", ["lrelu", "relu6", "none"])
def test_conv_prenorm_fused_input_norm(ops, act):
    """am_conv3d_prenorm: conv(act(x * scale + shift)) with the norm + activation applied while the resident-weight kernel stages its
    source rows (P/STUNet_head.py:96-103: conv2(LReLU(IN(conv1 x))) at the block-sparse 32-channel level 0; the normalised map is never
    written).  Against F.conv3d of the transformed, masked input (dense-conv-then-mask semantics of P/encoder3D.py:12-15: inactive source
    voxels read as ZERO, not as act(shift)) and against the two-launch form (norm_apply, conv3d); the statistics rows of the output too."""
    dtype = torch.bfloat16
    B, C, f, bs = 2, 32, (2, 3, 2), 4
    S = tuple(v << bs for v in f)
    x = q(rnd(B, C, *S, seed=501), dtype)
    w = q(rnd(C, C, 3, 3, 3, seed=502, scale=1.0 / np.sqrt(C * 27)), dtype)
    bias = rnd(C, seed=503)
    scale, shift = rnd(C, seed=504).abs() + 0.5, rnd(C, seed=505) * 0.5
    mask = mk_mask(B, f, 5, seed=21)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    mo = O.upsample_mask(mask, S).float()
    code = {"lrelu": ops.ACT_LRELU, "relu6": ops.ACT_RELU6, "none": ops.ACT_NONE}[act]
    fn = {"lrelu": lambda t: F.leaky_relu(t, 0.01), "relu6": lambda t: torch.clamp(t, 0, 6), "none": lambda t: t}[act]
    bc = lambda t: t.view(1, -1, 1, 1, 1)
    a = q(fn(x * bc(scale) + bc(shift)), dtype) * mo             # what norm_apply would have stored, zero where inactive
    ref = F.conv3d(a, w, bias, padding=1) * mo
    st = ops.NormStats(C, DEV)
    st.scale.copy_(scale.to(DEV)); st.shift.copy_(shift.to(DEV))
    wp = ops.pack_weight(w.to(DEV), dtype, False, False)
    xin = to_cl(x, dtype)
    assert ops.conv3d_prenorm_supported(xin, wp, S, 3, 1, mi, bs)
    y, part = ops.conv3d_prenorm(xin, st, code, wp, bias.to(DEV), S, 3, 1, mi, bs, want_partials=True)
    close(from_cl(y), ref, TOL[dtype], f"prenorm conv ({act})", mo)
    a_dev = ops.norm_apply(xin, st, code, mi, bs)
    y2 = ops.conv3d(ops.CONV_FWD, a_dev, wp, bias.to(DEV), S, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs)
    close(from_cl(y), from_cl(y2), 4e-3, f"prenorm conv vs norm_apply + conv ({act})", mo)      # (the same bf16-rounded operand up to fma ordering)
    st_a, st_b = ops.NormStats(C, DEV), ops.NormStats(C, DEV)
    part.reduce(sums=st_a.sums)
    ops.chan_stats(y, mi, bs, st_b)
    refs = st_b.sums.cpu().sum(0)
    assert (st_a.sums.cpu()[0] - refs).abs().max().item() <= 1e-5 * refs.abs().max().item()


@pytest.mark.parametrize("n_active", [4800, 5000])
def test_conv_prenorm_predicate_matches_launch_at_large_active_counts(ops, n_active):
    """am_conv3d_prenorm_supported must PREDICT am_conv3d_prenorm: the resident-weight kernel keeps a brick table of its
    workgroup in LDS that grows with the number of active patches, and at level 0 (16^3 patches, 32 channels) brick + weights + table
    stop fitting 160 KB above 4864 active patches (STUNet-B 128^3 mask 0.6 from batch 24 on).  Below: the fused launch is taken and
    equals norm_apply + conv3d.  Above: the predicate says no (the engine then runs the two-launch form, which every other kernel
    serves) -- it used to say yes and the launch failed with -7 inside the teacher's forward."""
    dtype = torch.bfloat16
    B, C, f, bs = 20, 32, (8, 8, 8), 4
    S = tuple(v << bs for v in f)
    g = torch.Generator().manual_seed(n_active)
    idx = torch.randperm(B * 512, generator=g)[:n_active]
    mask = torch.zeros(B * 512, dtype=torch.bool).scatter_(0, idx, True).view(B, 1, *f)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    assert mi.active_list()[1] == n_active
    x = (torch.randn(B, *S, C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.5).to(dtype)
    w = q(rnd(C, C, 3, 3, 3, seed=602, scale=1.0 / np.sqrt(C * 27)), dtype)
    wp = ops.pack_weight(w.to(DEV), dtype, False, False)
    st = ops.NormStats(C, DEV)
    st.scale.copy_((rnd(C, seed=604).abs() + 0.5).to(DEV)); st.shift.copy_((rnd(C, seed=605) * 0.5).to(DEV))
    ok = ops.conv3d_prenorm_supported(x, wp, S, 3, 1, mi, bs)
    assert ok == (n_active <= 4864), (n_active, ok)
    a_dev = ops.norm_apply(x, st, ops.ACT_LRELU, mi, bs)
    y2, part2 = ops.conv3d(ops.CONV_FWD, a_dev, wp, None, S, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, want_partials=True)
    torch.cuda.synchronize()
    if ok:
        y, part = ops.conv3d_prenorm(x, st, ops.ACT_LRELU, wp, None, S, 3, 1, mi, bs, want_partials=True)
        lst = mi.active_list()[0][:n_active].cpu().tolist()      # b << 24 | pd << 16 | ph << 8 | pw
        # compare on the active patches (what a kernel leaves at inactive voxels is unspecified)
        for k in range(0, n_active, 97):
            v = lst[k]
            b_, pd, ph, pw = v >> 24, (v >> 16) & 255, (v >> 8) & 255, v & 255
            pick = lambda t: t[b_, pd * 16:pd * 16 + 16, ph * 16:ph * 16 + 16, pw * 16:pw * 16 + 16].float()
            d = (pick(y) - pick(y2)).abs().max().item()
            assert d <= 4e-3 * pick(y2).abs().max().item() + 1e-3, (k, d)
    else:
        with pytest.raises(RuntimeError):                 # the launch itself still refuses (-7): the predicate is what the engine asks
            ops.conv3d_prenorm(x, st, ops.ACT_LRELU, wp, None, S, 3, 1, mi, bs, want_partials=True)
    s2 = ops.NormStats(C, DEV)
    part2.reduce(sums=s2.sums)
    assert torch.isfinite(s2.sums).all() and float(s2.sums.abs().sum()) > 0


@pytest.mark.parametrize("case", [(64, 64, (32, 32, 32), 2), (128, 64, (16, 32, 32), 4), (64, 128, (24, 16, 48), 6), (64, 64, (9, 8, 16), 64),
                                  (64, 32, (32, 32, 32), 2), (192, 96, (16, 16, 32), 8), (64, 32, (9, 8, 16), 64)])   # 32-wide cy tiles (Cout = 32, 96)
def test_conv_wgrad_k3_dense_8wave_dma_kernel(ops, case):
    """conv_wgk3.hip (round 5): the dense bf16 k3 s1 weight gradient with 64 x 64 channel tiles on the persistent 8-wave LDS-DMA kernel
    (both operands by DMA into swizzled unpadded rows, transposing fragment reads, the two waves of a SIMD in antiphase) against
    F.conv3d autograd on the same bf16-rounded operands: 2e-2 of max (the bf16 bound of every conv test here; the sums are fp32).  Cases:
    one tile / two cx tiles / two cy tiles with a ragged depth / a one-brick-per-plane grid with an odd depth and many samples (every
    brick is a border brick in h and w: the zero-filled halo rows).  The same launch in deterministic mode runs conv_wgrad.hip's
    ordered fold: the two kernels must agree to fp32 summation order."""
    cin, cout, S, B = case
    dtype = torch.bfloat16
    x = q(rnd(B, cin, *S, seed=401), dtype)
    dy = q(rnd(B, cout, *S, seed=402), dtype)
    from anatomask_amd import hip
    assert hip.lib()._lib.am_conv3d_wgrad_uses_k3(ops.CONV_FWD, hip.DT_BF16, 3, 1, B, *S, cin, cout, 0, 0) == 1, "the case must reach the new kernel"
    assert hip.lib()._lib.am_conv3d_wgrad_uses_k3(ops.CONV_FWD, hip.DT_BF16, 3, 1, B, *S, cin, cout, 0, 1) == 0
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, w, None, padding=1).backward(dy)
    dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), 3, 1)
    dw = torch.zeros(cout, cin, 3, 3, 3, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
    close(dw.cpu(), w.grad, TOL[dtype], "conv_wgk3 weight gradient")
    ops.DETERMINISTIC_WGRAD = True
    try:
        dwp2 = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), 3, 1)
    finally:
        ops.DETERMINISTIC_WGRAD = False
    close(dwp.cpu(), dwp2.cpu(), 1e-4, "conv_wgk3 vs conv_wgrad (deterministic fold)")
    # every tap separately (a wrong tap index or a shifted halo shows here even if the maximum hides it)
    ref = w.grad.permute(2, 3, 4, 0, 1).reshape(27, cout, cin)
    got = dwp.cpu()
    for t in range(27):
        assert (got[t] - ref[t]).abs().max().item() <= TOL[dtype] * ref[t].abs().max().item() + 1e-6, f"tap {t}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [32, 64, 128, 192])
def test_head_stencil_equals_conv_bn_proj_chain(ops, dtype, C):
    """am_head_fold + am_head_stencil: the decoder's eval-mode tail conv3x3x3(C -> C/2, no bias) -> BatchNorm3d(running statistics) ->
    Conv3d(C/2 -> 1, k1, bias) (P/decoder3D.py:20-22,51,61) as ONE C -> 1 stencil on the needed 16^3 patches, against the three-op chain
    in fp32 on the same (storage-rounded) input: rec on the needed patches (other patches untouched), the raw per-patch l2 against a
    volume (P/pretrain_AntoMask.py:423), border patches (zero padding) included.  fp32 storage: <= 2e-4 of max (summation order);
    bf16 storage: the input is exact in bf16 and the folded weights enter as hi + lo parts: <= 1e-3 of max."""
    B, S = 2, (32, 32, 48)
    cmid = C // 2
    r = q(rnd(B, C, *S, seed=301).clamp(0, 6), dtype)                       # a ReLU6 output
    w2 = rnd(cmid, C, 3, 3, 3, seed=302, scale=1.0 / np.sqrt(C * 27))
    gamma, beta = rnd(cmid, seed=303).abs() + 0.5, rnd(cmid, seed=304) * 0.2
    rmean, rvar = rnd(cmid, seed=305) * 0.3, rnd(cmid, seed=306).abs() + 0.3
    wp, bp = rnd(1, cmid, 1, 1, 1, seed=307, scale=0.3), rnd(1, seed=308)
    inp = rnd(B, *S, seed=309)
    chain = F.conv3d(F.batch_norm(F.conv3d(r, w2, None, padding=1), rmean, rvar, gamma, beta, training=False, eps=1e-5), wp, bp)[:, 0]
    f = tuple(v // 16 for v in S)
    need = mk_mask(B, f, 7, seed=11)                                        # 7 of 12 patches per sample, corner patches among them
    mi = ops.MaskInfo.from_bool(need, DEV)
    st = ops.NormStats(cmid, DEV)
    ops.norm_fold_running(st, gamma.to(DEV), beta.to(DEV), rmean.to(DEV), rvar.to(DEV), 1e-5)
    weff, beff = ops.head_fold(w2.to(DEV), st.scale, st.shift, wp.view(-1).to(DEV), bp.to(DEV))
    # the folded stencil itself, against its definition
    sc = gamma / torch.sqrt(rvar + 1e-5)
    weff_ref = torch.einsum("c,coxyz->xyzo", wp.view(-1) * sc, w2).reshape(27, C)
    assert (weff.cpu() - weff_ref).abs().max().item() <= 1e-6 * weff_ref.abs().max().item() + 1e-8
    x_cl = to_cl(r, dtype)
    assert ops.head_stencil_supported(x_cl)
    rec = torch.full((B, *S), 777.0, device=DEV)
    l2 = torch.full((B, f[0] * f[1] * f[2]), -1.0, device=DEV)
    ops.head_stencil(x_cl, weff, beff, mi, rec=rec, inp=inp.to(DEV), l2=l2)
    up = O.upsample_mask(need, S)[:, 0]
    tol = 2e-4 if dtype == torch.float32 else 1e-3
    scale = chain.abs().max().item()
    got = rec.cpu()
    assert (got - chain)[up].abs().max().item() <= tol * scale, ((got - chain)[up].abs().max().item(), scale)
    assert (got[~up] == 777.0).all(), "a patch outside the list was written"
    e = ((chain - inp) ** 2).reshape(B, f[0], 16, f[1], 16, f[2], 16).mean(dim=(2, 4, 6)).reshape(B, -1)
    nm = need.reshape(B, -1)
    l2c = l2.cpu()
    assert (l2c[nm] - e[nm]).abs().max().item() <= 2 * tol * e[nm].max().item()
    assert (l2c[~nm] == -1.0).all()
    # l2 alone (the teacher pass: rec never exists) gives the same numbers
    l2b = torch.zeros_like(l2)
    ops.head_stencil(x_cl, weff, beff, mi, inp=inp.to(DEV), l2=l2b)
    assert torch.equal(l2b.cpu()[nm], l2c[nm])


# ------------------------------------------------------------------ kernel branches only STUNet-L / STUNet-H reach
# (P/pretrain_AnatoMask_DDP.py:222-229: depth 2/3, dims 64..1024 / 96..1536, patches 160^3 / 192^3 -> 10^3 / 12^3 grids of
# one-voxel patches at level 4, 20- / 24-wide decoder grids, 96-channel level 0).  Every case: forward, data gradient and weight
# gradient against F.conv3d / F.conv_transpose3d autograd on the same bf16-rounded inputs.
def _conv_case(ops, dtype, cin, cout, k, s, so, B, sparse, bs_out, det=False, seed=50, keep_frac=0.4):
    si = tuple(v * s for v in so)
    x = q(rnd(B, cin, *si, seed=seed), dtype)
    w = q(rnd(cout, cin, k, k, k, seed=seed + 1, scale=1.0 / np.sqrt(cin * k ** 3)), dtype)
    bias = rnd(cout, seed=seed + 2)
    dy = q(rnd(B, cout, *so, seed=seed + 3), dtype)
    mask = mi = mo = mi_in = None
    bs_in = bs_out + (1 if s == 2 else 0)
    if sparse:
        f = tuple(v >> bs_out for v in so)
        assert all((fv << bs_out) == v for fv, v in zip(f, so))
        mask = mk_mask(B, f, max(1, int(round(keep_frac * f[0] * f[1] * f[2]))), seed=seed + 4)
        mi = ops.MaskInfo.from_bool(mask, DEV)
        mo, mi_in = O.upsample_mask(mask, so).float(), O.upsample_mask(mask, si).float()
        x, dy = x * mi_in, dy * mo
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, bias, stride=s, padding=k // 2)
    if sparse:
        yr = yr * mo
    yr.backward(dy)
    wd = w.to(DEV)
    y = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), ops.pack_weight(wd, dtype, False, False), bias.to(DEV), so, k, s,
                   in_mask=mi, in_bshift=bs_in, out_mask=mi, out_bshift=bs_out)
    close(from_cl(y), yr.detach(), TOL[dtype], "conv fwd", mo)
    dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), ops.pack_weight(wd, dtype, False, True), None, si, k, s,
                    in_mask=mi, in_bshift=bs_out, out_mask=mi, out_bshift=bs_in)
    close(from_cl(dx), xr.grad, TOL[dtype], "conv dgrad", mi_in)
    for d in ([False, True] if det else [False]):
        ops.DETERMINISTIC_WGRAD = d
        try:
            dwp = ops.conv3d_wgrad(ops.CONV_FWD, to_cl(x, dtype), to_cl(dy, dtype), k, s, x_mask=mi, x_bshift=bs_in, y_mask=mi, y_bshift=bs_out)
        finally:
            ops.DETERMINISTIC_WGRAD = False
        dw = torch.zeros_like(w, device=DEV)
        ops.unpack_grad(dwp, dw, transposed_conv=False, accumulate=False)
        close(dw.cpu(), wr.grad, TOL[dtype], f"conv wgrad (det={d})")


@pytest.mark.parametrize("case", [(64, 64, (2, 3, 2), 3, 2, 1), (64, 96, (1, 2, 3), 4, 1, 1), (64, 64, (3, 2, 2), 3, 2, 2), (48, 64, (2, 2, 3), 3, 3, 1)])
def test_conv_block_sparse_live_brick_grids(ops, case):
    """Block-sparse launches whose bricks lie inside the patches (8- and 16-wide patches) enumerate their live bricks from the
    active-patch list (conv_igemm.hip `ConvArgs.plist`): forward (stride 1 / 2) with bias and statistics rows, data gradient plain and
    accumulating, the k1 s2 shortcut's accumulating data gradient (one live class), non-cubic mask grids, NaN in every inactive voxel."""
    cin, cout, f, bs, B, stride = case
    dtype = torch.bfloat16
    so = tuple(v << bs for v in f)
    si = tuple(v * stride for v in so)
    bs_in = bs + (1 if stride == 2 else 0)
    mask = mk_mask(B, f, max(1, (f[0] * f[1] * f[2] * 2) // 5), seed=81)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    m_in, m_out = O.upsample_mask(mask, si).float(), O.upsample_mask(mask, so).float()
    x = q(rnd(B, cin, *si, seed=82), dtype) * m_in
    w = q(rnd(cout, cin, 3, 3, 3, seed=83, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=84)
    dy = q(rnd(B, cout, *so, seed=85), dtype) * m_out
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, bias, stride=stride, padding=1) * m_out
    yr.backward(dy)
    nan = lambda t, mm: torch.where(to_cl(mm.expand(B, t.shape[-1], *mm.shape[2:]), dtype) > 0, t, torch.full_like(t, float("nan")))
    wd = w.to(DEV)
    y, part = ops.conv3d(ops.CONV_FWD, nan(to_cl(x, dtype), m_in), ops.pack_weight(wd, dtype, False, False), bias.to(DEV), so, 3, stride,
                         in_mask=mi, in_bshift=bs_in, out_mask=mi, out_bshift=bs, want_partials=True,
                         out=torch.full((B, *so, cout), float("nan"), device=DEV, dtype=dtype))
    close(from_cl(y), yr.detach(), TOL[dtype], "live-brick conv fwd", m_out)
    assert torch.isnan(from_cl(y)[(m_out == 0).expand_as(yr)]).all()       # inactive voxels are never written
    ys = torch.nan_to_num(y.float()).cpu()
    rows = part.t[:part.rows].double().sum(0).cpu()
    want = torch.stack([ys.double().sum((0, 1, 2, 3)), (ys.double() ** 2).sum((0, 1, 2, 3))], 1)
    assert torch.allclose(rows, want, rtol=2e-4, atol=1e-3), (rows - want).abs().max()
    wb = ops.pack_weight(wd, dtype, False, True)
    dx = ops.conv3d(ops.CONV_DGRAD, nan(to_cl(dy, dtype), m_out), wb, None, si, 3, stride, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs_in)
    close(from_cl(dx), xr.grad, TOL[dtype], "live-brick conv dgrad", m_in)
    base = q(rnd(B, cin, *si, seed=86), dtype) * m_in
    dx2 = ops.conv3d(ops.CONV_DGRAD, nan(to_cl(dy, dtype), m_out), wb, None, si, 3, stride, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs_in,
                     out=nan(to_cl(base, dtype), m_in), accumulate=True)
    close(from_cl(dx2), xr.grad + base, 2 * TOL[dtype], "live-brick conv dgrad, accumulating", m_in)
    if stride == 2:                                          # the 1x1 stride-2 shortcut's data gradient, accumulated onto dx (engine.backward)
        w1 = q(rnd(cout, cin, 1, 1, 1, seed=87, scale=1.0 / np.sqrt(cin)), dtype)
        x1 = x.clone().requires_grad_(True)
        (F.conv3d(x1, w1, None, stride=2) * m_out).backward(dy)
        dx3 = ops.conv3d(ops.CONV_DGRAD, nan(to_cl(dy, dtype), m_out), ops.pack_weight(w1.to(DEV), dtype, False, True), None, si, 1, 2,
                         in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs_in, out=dx.clone(), accumulate=True)
        close(from_cl(dx3), xr.grad + x1.grad, 2 * TOL[dtype], "k1 s2 dgrad accumulated onto the k3 s2 dgrad", m_in)


@pytest.mark.parametrize("case", [(256, 256, (4, 4, 4), 1, 2, 1), (128, 192, (5, 6, 7), 0, 3, 1), (512, 256, (3, 4, 5), 1, 2, 1), (384, 128, (6, 5, 4), 0, 1, 1), (256, 64, (2, 2, 3), 1, 1, 1),
                                  (128, 128, (3, 3, 4), 2, 2, 1), (128, 128, (6, 6, 5), 2, 8, 1), (128, 384, (3, 2, 2), 2, 16, 1),
                                  (128, 256, (4, 3, 5), 0, 2, 2), (256, 128, (3, 3, 3), 1, 2, 2), (128, 64, (2, 3, 2), 2, 1, 2), (64, 128, (3, 3, 2), 2, 2, 2), (192, 64, (2, 2, 5), 1, 1, 2)])
def test_conv_gather_small_patches(ops, case):
    """Levels whose patches are 4^3 / 2^3 voxels or one voxel run on the voxel-list gather kernel (conv_gather.hip: rows of the implicit
    GEMM = the active voxels of the active-patch list, source fragments gathered per tap): forward (stride 1 and 2) and data gradient
    against F.conv3d of the zero-filled tensor (P/encoder3D.py:12-15), bias, the (sum, sum of squares) rows of the stored values, NaN in
    every inactive voxel; ragged last voxel tile, 128- and 256-channel stages, 64- and 128-channel tiles (the latter from 256 workgroups
    up: the B=8 case), several samples per tile."""
    cin, cout, f, bs, B, stride = case
    dtype = torch.bfloat16
    so = tuple(v << bs for v in f)
    si = tuple(v * stride for v in so)
    bs_in = bs + (1 if stride == 2 else 0)
    mask = mk_mask(B, f, max(1, (f[0] * f[1] * f[2] * 2) // 5), seed=71)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    m_in, m_out = O.upsample_mask(mask, si).float(), O.upsample_mask(mask, so).float()
    x = q(rnd(B, cin, *si, seed=72), dtype) * m_in
    w = q(rnd(cout, cin, 3, 3, 3, seed=73, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=74)
    dy = q(rnd(B, cout, *so, seed=75), dtype) * m_out
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, bias, stride=stride, padding=1) * m_out
    yr.backward(dy)
    nan = lambda t, mm: torch.where(to_cl(mm.expand(B, t.shape[-1], *mm.shape[2:]), dtype) > 0, t, torch.full_like(t, float("nan")))
    wd = w.to(DEV)
    y, part = ops.conv3d(ops.CONV_FWD, nan(to_cl(x, dtype), m_in), ops.pack_weight(wd, dtype, False, False), bias.to(DEV), so, 3, stride,
                         in_mask=mi, in_bshift=bs_in, out_mask=mi, out_bshift=bs, want_partials=True,
                         out=torch.full((B, *so, cout), float("nan"), device=DEV, dtype=dtype))
    n_vox = int(mask.sum()) << (3 * bs)
    assert part.rows == (n_vox + 127) // 128                 # one row per tile of 128 active voxels: the gather kernel took the launch
    close(from_cl(y), yr.detach(), TOL[dtype], "gather conv fwd", m_out)
    assert torch.isnan(from_cl(y)[(m_out == 0).expand_as(yr)]).all()       # inactive voxels are never written
    ys = torch.nan_to_num(y.float()).cpu()                   # the stored bf16 values
    rows = part.t[:part.rows].double().sum(0).cpu()
    want = torch.stack([ys.double().sum((0, 1, 2, 3)), (ys.double() ** 2).sum((0, 1, 2, 3))], 1)
    assert torch.allclose(rows, want, rtol=2e-4, atol=1e-3), (rows - want).abs().max()
    # data gradient (stride 2: the 8 output parities of every coarse voxel are 8 classes of workgroups), plain and accumulating
    wb = ops.pack_weight(wd, dtype, False, True)
    dx = ops.conv3d(ops.CONV_DGRAD, nan(to_cl(dy, dtype), m_out), wb, None, si, 3, stride, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs_in)
    close(from_cl(dx), xr.grad, TOL[dtype], "gather conv dgrad", m_in)
    base = q(rnd(B, cin, *si, seed=76), dtype) * m_in
    dx2 = ops.conv3d(ops.CONV_DGRAD, nan(to_cl(dy, dtype), m_out), wb, None, si, 3, stride, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs_in,
                     out=nan(to_cl(base, dtype), m_in), accumulate=True)
    close(from_cl(dx2), xr.grad + base, 2 * TOL[dtype], "gather conv dgrad, accumulating", m_in)


@pytest.mark.parametrize("case", [
    # cin, cout, k, s, output grid, B, sparse, out bshift
    (1536, 1536, 3, 1, (12, 12, 12), 1, True, 0),     # STUNet-H stage 4: 1728 wgrad tiles (slot count of conv_wgrad.hip launch()), one-voxel patches
    (1024, 1024, 3, 1, (10, 10, 10), 2, True, 0),     # STUNet-L stage 4 on its 10^3 grid (ragged 4x8x8 bricks)
    (1024, 1024, 3, 1, (5, 20, 20), 1, False, 0),     # STUNet-L decoder block 0 convs: dense, 20-wide (forward / data gradient on conv_gather.hip's dense rows; weight gradient: plane bricks, 256 tiles x 3 groups)
    (1536, 768, 3, 1, (3, 24, 24), 1, False, 0),      # STUNet-H decoder block 0 second conv: 24-wide
    (768, 1536, 3, 2, (12, 12, 12), 1, True, 0),      # STUNet-H stage 4 strided conv1 (24^3 two-voxel patches -> 12^3 one-voxel patches)
    (512, 1024, 1, 2, (10, 10, 10), 1, True, 0),      # STUNet-L stage 4 1x1 stride-2 shortcut
])
def test_conv_large_model_deep_levels(ops, case):
    cin, cout, k, s, so, B, sparse, bs = case
    _conv_case(ops, torch.bfloat16, cin, cout, k, s, so, B, sparse, bs, det=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    (96, 96, 3, 1, (16, 16, 32), 1, True, 4),         # STUNet-H level 0: 96-channel operands -> three 32-wide cy tiles in the weight gradient
    (96, 96, 3, 1, (8, 16, 32), 1, False, 0),         #   ... dense (plane bricks, MI = 2)
    (96, 192, 3, 2, (8, 8, 16), 2, True, 3),          # STUNet-H level 1 strided conv (full-resolution S2 staging, 96-channel X)
    (192, 96, 3, 1, (4, 8, 16), 2, False, 0),         # decoder output level of H: Cx 192 / Cy 96
    (64, 64, 3, 1, (10, 10, 10), 2, True, 0),         # one-voxel patches on a 10^3 grid, thin channels (brick straddles patches in every direction)
    (128, 128, 3, 1, (12, 12, 12), 1, True, 0),       # ... 12^3 (pick_tiling counts bricks per dimension)
    (64, 128, 3, 2, (10, 10, 10), 1, True, 0),        # stride 2 onto one-voxel patches
    (256, 256, 3, 1, (20, 20, 20), 1, True, 1),       # STUNet-L stage 3: two-voxel patches on a 20-wide grid
])
def test_conv_large_model_shapes(ops, dtype, case):
    cin, cout, k, s, so, B, sparse, bs = case
    _conv_case(ops, dtype, cin, cout, k, s, so, B, sparse, bs, det=(dtype == torch.bfloat16))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(64, 64, (5, 20, 20), 1), (96, 48, (3, 24, 24), 1), (128, 128, (5, 10, 10), 2), (64, 32, (6, 12, 12), 1), (32, 32, (10, 40, 40), 1),
                                  (256, 128, (6, 12, 12), 1), (128, 192, (10, 10, 10), 2), (384, 64, (3, 5, 20), 3)])     # (>= 128 input channels on ragged grids: the forward runs on conv_gather.hip)
def test_conv_transpose_large_model_grids(ops, dtype, case):
    """ConvTranspose3d on the q grids of STUNet-L 160^3 (10, 20, 40) and STUNet-H 192^3 (12, 24): conv_igemm.hip brick_shape picks the
    brick that pads the (h, w) plane less (4x8x8 on 20 / 24, 4x4x16 on 40) -- forward, data gradient, weight gradient."""
    cin, cout, si, B = case
    so = tuple(2 * v for v in si)
    x = q(rnd(B, cin, *si, seed=61), dtype)
    w = q(rnd(cin, cout, 4, 4, 4, seed=62, scale=1.0 / np.sqrt(cin * 8)), dtype)
    bias = rnd(cout, seed=63)
    dy = q(rnd(B, cout, *so, seed=64), dtype)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv_transpose3d(xr, wr, bias, stride=2, padding=1).backward(dy)
    yr = F.conv_transpose3d(x, w, bias, stride=2, padding=1)
    y = ops.conv3d(ops.CONVT_FWD, to_cl(x, dtype), ops.pack_weight(w.to(DEV), dtype, True, False), bias.to(DEV), so, 4, 2)
    close(from_cl(y), yr, TOL[dtype], "convT fwd")
    dx = ops.conv3d(ops.CONVT_DGRAD, to_cl(dy, dtype), ops.pack_weight(w.to(DEV), dtype, True, True), None, si, 4, 2)
    close(from_cl(dx), xr.grad, TOL[dtype], "convT dgrad")
    for det in ([False, True] if dtype == torch.bfloat16 else [False]):
        ops.DETERMINISTIC_WGRAD = det
        try:
            dwp = ops.conv3d_wgrad(ops.CONVT_FWD, to_cl(x, dtype), to_cl(dy, dtype), 4, 2)
        finally:
            ops.DETERMINISTIC_WGRAD = False
        dw = torch.zeros_like(w, device=DEV)
        ops.unpack_grad(dwp, dw, transposed_conv=True, accumulate=False)
        close(dw.cpu(), wr.grad, TOL[dtype], f"convT wgrad (det={det})")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ch", [(32, 32), (16, 16), (64, 64)])
def test_conv_transpose(ops, dtype, ch):
    cin, cout = ch
    B, si = 2, (4, 6, 8)
    so = tuple(2 * v for v in si)
    x = q(rnd(B, cin, *si, seed=1), dtype)
    w = q(rnd(cin, cout, 4, 4, 4, seed=2, scale=1.0 / np.sqrt(cin * 8)), dtype)
    bias = rnd(cout, seed=3)
    dy = q(rnd(B, cout, *so, seed=4), dtype)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, bias, stride=2, padding=1)
    yr.backward(dy)
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=True, for_dgrad=False)
    y = ops.conv3d(ops.CONVT_FWD, to_cl(x, dtype), wp, bias.to(DEV), so, 4, 2)
    close(from_cl(y), yr.detach(), TOL[dtype], "convT fwd")
    wpd = ops.pack_weight(w.to(DEV), dtype, transposed_conv=True, for_dgrad=True)
    dx = ops.conv3d(ops.CONVT_DGRAD, to_cl(dy, dtype), wpd, None, si, 4, 2)
    close(from_cl(dx), xr.grad, TOL[dtype], "convT dgrad")
    dwp = ops.conv3d_wgrad(ops.CONVT_FWD, to_cl(x, dtype), to_cl(dy, dtype), 4, 2)
    dw = torch.zeros_like(w, device=DEV)
    ops.unpack_grad(dwp, dw, transposed_conv=True, accumulate=False)
    close(dw.cpu(), wr.grad, TOL[dtype], "convT wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_transpose_512_voxel_bricks(ops, dtype):
    """Dense transposed convolutions with >= 4096 workgroups of the 4x4x16 plan take 4x8x16 q-bricks (conv_igemm.hip pick_tiling
    shape 3): a grid large enough to trigger them, with extents that are not brick multiples in any dimension."""
    cin, cout, B, si = 64, 64, 4, (15, 33, 64)
    so = tuple(2 * v for v in si)
    x = q(rnd(B, cin, *si, seed=1), dtype)
    w = q(rnd(cin, cout, 4, 4, 4, seed=2, scale=1.0 / np.sqrt(cin * 8)), dtype)
    bias = rnd(cout, seed=3)
    yr = F.conv_transpose3d(x, w, bias, stride=2, padding=1)
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=True, for_dgrad=False)
    y = ops.conv3d(ops.CONVT_FWD, to_cl(x, dtype), wp, bias.to(DEV), so, 4, 2)
    close(from_cl(y), yr, TOL[dtype], "convT fwd (512-voxel bricks)")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_odd_extent(ops, dtype):
    """Grids that are not multiples of the 4x8x8 brick (the reference recipe's 7x7x8 stage-4 grid)."""
    B, cin, cout, sp = 1, 32, 48, (7, 7, 8)
    x = q(rnd(B, cin, *sp, seed=1), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=2, scale=0.05), dtype)
    yr = F.conv3d(x, w, None, padding=1)
    wp = ops.pack_weight(w.to(DEV), dtype, False, False)
    y = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, sp, 3, 1)
    close(from_cl(y), yr, TOL[dtype], "conv odd extent")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,C", [(1, 16), (3, 16), (1, 32), (3, 32), (3, 64), (1, 96), (3, 96)])    # bf16 with C in {32, 64, 96}: matrix-core stem kernels
def test_stem_conv(ops, dtype, k, C):
    B, f = 2, (1, 2, 2)
    sp = tuple(v * 16 for v in f)                   # stage-0 tensor: 16^3 patches (block shift 4)
    mask = mk_mask(B, f, 2)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    x = rnd(B, 1, *sp, seed=1)
    w = rnd(C, 1, k, k, k, seed=2, scale=0.3).requires_grad_(True)
    b = rnd(C, seed=3).requires_grad_(True)
    dy = q(rnd(B, C, *sp, seed=4), dtype) * O.upsample_mask(mask, sp).float()
    yr = O.sparse_conv3d(x * O.upsample_mask(mask, sp).float(), w, b, 1, mask)
    yr.backward(dy)
    y, part = ops.stem_conv_fwd(x[:, 0].contiguous().to(DEV), w.detach().to(DEV), b.detach().to(DEV), mi, 4, dtype, want_partials=True)
    close(from_cl(y), yr.detach(), TOL[dtype], "stem fwd", O.upsample_mask(mask, sp).float())
    sums = torch.zeros(C, 2, device=DEV, dtype=torch.float64)            # statistics partials = sums of the STORED active values
    part.reduce(sums=sums)
    ya = torch.where(O.upsample_mask(mask, sp).expand(B, C, *sp), from_cl(y), torch.zeros(())).double()   # inactive voxels: don't-care bits
    n_act = float(O.upsample_mask(mask, sp).sum())
    e1 = (sums[:, 0].cpu() - ya.sum((0, 2, 3, 4))).abs().max().item()
    e2 = (sums[:, 1].cpu() - (ya * ya).sum((0, 2, 3, 4))).abs().max().item()
    # per-workgroup partials are fp32 sums of 512 (VALU kernel) / 4096 (matrix-core kernel: one row per patch) values: ~1e-7 * |values| * n
    assert part.rows in (B * 4 * 2 * 1 * f[0] * f[1] * f[2], mi.n_active)
    assert e1 <= 4e-6 * n_act * ya.abs().max().item(), (e1, n_act)
    assert e2 <= 4e-6 * n_act * (ya * ya).max().item(), (e2, n_act)
    dw = torch.zeros(C, k ** 3, device=DEV); db = torch.zeros(C, device=DEV)
    ops.stem_conv_wgrad(x[:, 0].contiguous().to(DEV), to_cl(dy, dtype), k, mi, 4, dw, db)
    # matrix-core kernels (bf16, C in {32, 64, 96}): the fp32 volume (and, forward, the fp32 weights) enter as hi + lo bf16 parts -- the
    # same 5e-4 as the VALU kernels against the UNROUNDED volume (a plain bf16 rounding of x moved this gradient by 4e-3, and the
    # stem weight's gradient of a full step from 0.43 to 0.64 of relative distance to fp32: profiles/r05_experiments.md section 7)
    close(dw.cpu().view_as(w), w.grad, 5e-4, "stem wgrad")
    close(db.cpu(), b.grad, 5e-4, "stem bgrad")
    if dtype == torch.bfloat16:                                          # forward: the only rounding left is the bf16 store of y
        act = O.upsample_mask(mask, sp).expand(B, C, *sp)
        ya32 = torch.where(act, from_cl(y).float(), torch.zeros(()))      # inactive voxels: don't-care bits
        yq = torch.where(act, q(yr.detach(), dtype), torch.zeros(()))
        assert (ya32 - yq).abs().max() <= 2.0 ** -8 * yr.detach().abs().max() and ((ya32 - yq).abs() > 1e-6 * yr.detach().abs().max()).float().mean() < 5e-2


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [16, 40, 96, 1280, 1536])      # fp32 rows of more than 256 16-byte chunks (C > 1024: STUNet-H) take 512-thread workgroups
def test_sparse_instance_norm_fwd_bwd(ops, dtype, C):
    B, f = 2, (2, 2, 3)
    sp = tuple(v * 4 for v in f)
    mask = torch.zeros(B, 1, *f, dtype=torch.bool)
    mask.view(B, -1)[0, :3] = True; mask.view(B, -1)[1, 2:9] = True          # unequal counts per sample
    mi = ops.MaskInfo.from_bool(mask, DEV)
    mf = O.upsample_mask(mask, sp).float()
    x = (q(rnd(B, C, *sp, seed=1), dtype) * mf).requires_grad_(True)
    res = (q(rnd(B, C, *sp, seed=5), dtype) * mf).requires_grad_(True)
    gam, bet = (1 + 0.2 * rnd(C, seed=2)).requires_grad_(True), (0.1 * rnd(C, seed=3)).requires_grad_(True)
    dout = q(rnd(B, C, *sp, seed=4), dtype) * mf
    yr = F.leaky_relu(O.sparse_instance_norm(x, gam, bet, 1e-5, mask) + res, 0.01)
    yr.backward(dout)
    st = ops.NormStats(C, DEV)
    cnt = torch.zeros(1, device=DEV, dtype=torch.float64)
    ops.mask_count(mi, 64, cnt)
    assert cnt.item() == mf[:, 0].sum().item()
    st.count_ptr = cnt
    xd = to_cl(x.detach(), dtype)
    ops.chan_stats(xd, mi, 2, st)
    ops.norm_finalize(st, gam.detach().to(DEV), bet.detach().to(DEV), 1e-5)
    y = ops.norm_apply(xd, st, ops.ACT_LRELU, mi, 2, res=to_cl(res.detach(), dtype))
    close(from_cl(y), yr.detach(), TOL[dtype], "IN fwd", mf)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dres = torch.empty_like(xd)
    # feed the oracle's own output so both sides take identical LeakyReLU gates
    dx = ops.norm_backward(to_cl(dout, dtype), to_cl(yr.detach(), dtype), xd, st, gam.detach().to(DEV), ops.ACT_LRELU, mi, 2, dg, db, dres=dres)
    close(from_cl(dx), x.grad, 2 * TOL[dtype], "IN dx", mf)
    close(from_cl(dres), res.grad, TOL[dtype], "IN dres", mf)
    close(dg.cpu(), gam.grad, 2 * TOL[dtype], "IN dgamma")
    close(db.cpu(), bet.grad, 2 * TOL[dtype], "IN dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [24, 1536])
def test_batchnorm_train_eval_relu6(ops, dtype, C):
    B, sp = 2, (6, 8, 10)
    x = q(rnd(B, C, *sp, seed=1) * 2 + 1.5, dtype).requires_grad_(True)
    p = {"bn.weight": (1 + 0.2 * rnd(C, seed=2)).requires_grad_(True), "bn.bias": (0.5 * rnd(C, seed=3)).requires_grad_(True),
         "bn.running_mean": 0.1 * rnd(C, seed=6), "bn.running_var": 1 + 0.1 * rnd(C, seed=7).abs(),
         "bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    nb = {}
    yr = F.relu6(O.batch_norm3d(p, "bn", x, True, nb))
    dout = q(rnd(B, C, *sp, seed=4), dtype)
    yr.backward(dout)
    st = ops.NormStats(C, DEV)
    st.count_host = float(B * sp[0] * sp[1] * sp[2])
    xd = to_cl(x.detach(), dtype)
    rm, rv = p["bn.running_mean"].clone().to(DEV), p["bn.running_var"].clone().to(DEV)
    ops.chan_stats(xd, None, 0, st)
    ops.norm_finalize(st, p["bn.weight"].detach().to(DEV), p["bn.bias"].detach().to(DEV), 1e-5, rm, rv, 0.1)
    y = ops.norm_apply(xd, st, ops.ACT_RELU6)
    close(from_cl(y), yr.detach(), TOL[dtype], "BN train fwd")
    close(rm.cpu(), nb["bn.running_mean"], 1e-5, "running_mean"); close(rv.cpu(), nb["bn.running_var"], 1e-5, "running_var")
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.norm_backward(to_cl(dout, dtype), to_cl(yr.detach(), dtype), xd, st, p["bn.weight"].detach().to(DEV), ops.ACT_RELU6, None, 0, dg, db)
    # (a saved bf16 output in (5.98, 6) rounds to 6.0 and closes the ReLU6 gate the fp32 reference holds open, and a recomputed
    # pre-activation within rounding of 0 or 6 may fall on the other side: with 1536 channels a few dozen of the 3 M elements do;
    # they are left out of the element-wise comparison)
    zr = O.batch_norm3d(p, "bn", x.detach(), True, {})                                # the reference's pre-activation
    edge = (((zr - 6.0).abs() > 0.03) & (zr.abs() > 1e-3)).float() if C > 256 else None   # (also the recomputed gate's knife edges at 0 and 6)
    close(from_cl(dx), x.grad, 2 * TOL[dtype], "BN dx", edge)
    close(dg.cpu(), p["bn.weight"].grad, 2 * TOL[dtype], "BN dgamma"); close(db.cpu(), p["bn.bias"].grad, 2 * TOL[dtype], "BN dbeta")
    # out=None: the ReLU6 gate is recomputed from x * scale + shift (what the engine does for norms without a residual)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx2 = ops.norm_backward(to_cl(dout, dtype), None, xd, st, p["bn.weight"].detach().to(DEV), ops.ACT_RELU6, None, 0, dg2, db2)
    close(from_cl(dx2), x.grad, 2 * TOL[dtype], "BN dx (recomputed gate)", edge)
    close(dg2.cpu(), p["bn.weight"].grad, 2 * TOL[dtype], "BN dgamma (recomputed gate)")
    close(db2.cpu(), p["bn.bias"].grad, 2 * TOL[dtype], "BN dbeta (recomputed gate)")
    # eval: running statistics
    ye = O.batch_norm3d(p, "bn", x.detach(), False, None)
    ops.norm_fold_running(st, p["bn.weight"].detach().to(DEV), p["bn.bias"].detach().to(DEV), p["bn.running_mean"].to(DEV),
                          p["bn.running_var"].to(DEV), 1e-5)
    close(from_cl(ops.norm_apply(xd, st, ops.ACT_NONE)), ye, TOL[dtype], "BN eval")


@pytest.mark.parametrize("dtype", DTYPES)
def test_densify_fill_fwd_bwd(ops, dtype):
    B, C, f = 2, 32, (2, 3, 2)
    sp = tuple(v * 2 for v in f)
    mask = mk_mask(B, f, 5)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    mb = O.upsample_mask(mask, sp)
    x = (q(rnd(B, C, *sp, seed=1), dtype) * mb.float()).requires_grad_(True)
    gam, bet = (1 + 0.2 * rnd(C, seed=2)).requires_grad_(True), (0.1 * rnd(C, seed=3)).requires_grad_(True)
    tok = (0.02 * rnd(1, C, 1, 1, 1, seed=8)).requires_grad_(True)
    yr = torch.where(mb, O.sparse_instance_norm(x, gam, bet, 1e-6, mask), tok.expand(B, C, *sp))
    dout = q(rnd(B, C, *sp, seed=4), dtype)
    yr.backward(dout)
    st = ops.NormStats(C, DEV)
    cnt = torch.zeros(1, device=DEV, dtype=torch.float64)
    ops.mask_count(mi, 8, cnt); st.count_ptr = cnt
    xd = to_cl(x.detach(), dtype)
    ops.chan_stats(xd, mi, 1, st)
    ops.norm_finalize(st, gam.detach().to(DEV), bet.detach().to(DEV), 1e-6)
    y = ops.norm_apply(xd, st, ops.ACT_NONE, mi, 1, fill=tok.detach().view(-1).to(DEV))
    close(from_cl(y), yr.detach(), TOL[dtype], "densify fwd")
    dg, db, dt = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.norm_backward(to_cl(dout, dtype), None, xd, st, gam.detach().to(DEV), ops.ACT_NONE, mi, 1, dg, db, dtoken=dt, fill=True)
    close(from_cl(dx), x.grad, 2 * TOL[dtype], "densify dx", mb.float())
    close(dt.cpu(), tok.grad.view(-1), 2 * TOL[dtype], "token grad")
    close(dg.cpu(), gam.grad, 2 * TOL[dtype], "densify dgamma")


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_shortcut_fused_apply(ops, dtype):
    """norm2 + (1x1 stem shortcut) + LeakyReLU of STUNet stage 0 fused in one pass."""
    B, C, f = 1, 16, (2, 2, 2)
    sp = tuple(v * 4 for v in f)
    mask = mk_mask(B, f, 4); mi = ops.MaskInfo.from_bool(mask, DEV); mf = O.upsample_mask(mask, sp).float()
    inp = rnd(B, 1, *sp, seed=9)
    x = q(rnd(B, C, *sp, seed=1), dtype) * mf
    gam, bet, w3, b3 = 1 + 0.2 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(C, 1, 1, 1, 1, seed=5), rnd(C, seed=6)
    yr = F.leaky_relu(O.sparse_instance_norm(x, gam, bet, 1e-5, mask) + O.sparse_conv3d(inp * mf, w3, b3, 1, mask), 0.01)
    st = ops.NormStats(C, DEV); cnt = torch.zeros(1, device=DEV, dtype=torch.float64); ops.mask_count(mi, 64, cnt); st.count_ptr = cnt
    xd = to_cl(x, dtype)
    ops.chan_stats(xd, mi, 2, st); ops.norm_finalize(st, gam.to(DEV), bet.to(DEV), 1e-5)
    y = ops.norm_apply(xd, st, ops.ACT_LRELU, mi, 2, stem=(inp[:, 0].contiguous().to(DEV), w3.view(-1).to(DEV), b3.to(DEV)))
    close(from_cl(y), yr, TOL[dtype], "stem-shortcut apply", mf)


@pytest.mark.parametrize("dtype", DTYPES)
def test_chan_sum_add_proj(ops, dtype):
    B, C, sp = 2, 32, (4, 6, 8)
    x = q(rnd(B, C, *sp, seed=1), dtype)
    xd = to_cl(x, dtype)
    out = torch.zeros(C, device=DEV)
    ops.chan_sum(xd, None, 0, out)
    close(out.cpu(), x.sum(dim=(0, 2, 3, 4)), 2 * TOL[dtype], "chan_sum")
    y2 = ops.add(xd, xd)
    close(from_cl(y2), 2 * x, TOL[dtype], "add")
    w, b = rnd(1, C, 1, 1, 1, seed=2).requires_grad_(True), rnd(1, seed=3).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    rr = F.conv3d(xr, w, b)
    drec = rnd(B, 1, *sp, seed=4)
    rr.backward(drec)
    rec = ops.proj_fwd(xd, w.detach().view(-1).to(DEV), b.detach().to(DEV))
    close(rec.cpu(), rr.detach()[:, 0], 1e-4, "proj fwd")
    dw, db = torch.zeros(C, device=DEV), torch.zeros(1, device=DEV)
    dx = ops.proj_bwd(xd, drec[:, 0].contiguous().to(DEV), w.detach().view(-1).to(DEV), dw, db)
    close(from_cl(dx), xr.grad, TOL[dtype], "proj dx")
    close(dw.cpu(), w.grad.view(-1), 5e-4, "proj dw"); close(db.cpu(), b.grad, 5e-4, "proj db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,sp,B", [(32, (8, 12, 16), 2), (96, (5, 6, 7), 3), (64, (16, 16, 16), 1)])
def test_projection_head_fused_with_its_batchnorm(ops, dtype, C, sp, B):
    """am_proj_fwd(pre_scale, pre_shift) + am_proj_norm_bwd: the last decoder block's train-mode BatchNorm applied and differentiated
    INSIDE the 1x1 projection (its output and the rank-1 gradient drec x w are never materialised) against torch autograd of
    conv1x1(batch_norm(x)) -- rec, dx and the gradients of proj.weight / proj.bias / BN weight / BN bias (P/decoder3D.py:22,51,61)."""
    x = q(rnd(B, C, *sp, seed=1) * 1.3 + 0.4, dtype)
    gam, bet = torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5, rnd(C, seed=3) * 0.3
    w, b = rnd(1, C, 1, 1, 1, seed=4) * 0.3, rnd(1, seed=5)
    drec = rnd(B, 1, *sp, seed=6)
    xr, gr, br, wr, pbr = (t.clone().requires_grad_(True) for t in (x, gam, bet, w, b))
    o = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    rr = F.conv3d(o, wr, pbr)
    rr.backward(drec)
    xd = to_cl(x, dtype)
    st = ops.NormStats(C, DEV)
    st.count_host = float(B * sp[0] * sp[1] * sp[2])
    ops.chan_stats(xd, None, 0, st)
    ops.norm_finalize(st, gam.to(DEV), bet.to(DEV), 1e-5)
    wd, bd = w.view(-1).to(DEV), b.to(DEV)
    rec = ops.proj_fwd(xd, wd, bd, pre=st)
    close(rec.cpu(), rr.detach()[:, 0], 2e-5 if dtype == torch.float32 else 1e-4, "fused head fwd")
    dg, dbt, dw, db = (torch.full((n,), 0.25, device=DEV) for n in (C, C, C, 1))      # accumulated INTO (0.25 already there)
    dx = ops.proj_norm_bwd(xd, st, drec[:, 0].contiguous().to(DEV), wd, gam.to(DEV), bet.to(DEV), dg, dbt, dw, db)
    close(from_cl(dx), xr.grad, TOL[dtype], "fused head dx")
    close(dg.cpu() - 0.25, gr.grad, 5e-4, "BN dgamma"); close(dbt.cpu() - 0.25, br.grad, 5e-4, "BN dbeta")
    close(dw.cpu() - 0.25, wr.grad.view(-1), 5e-4, "proj dw"); close(db.cpu() - 0.25, pbr.grad, 5e-4, "proj db")
    ws, _ = ops._bwd_workspaces(torch.device(DEV), C)
    assert bool((ws == 0).all()), "the shared zero workspace was left dirty"
    # and the stand-alone route (apply -> proj -> proj_bwd -> norm backward) agrees
    o2 = ops.norm_apply(xd, st, ops.ACT_NONE)
    rec2 = ops.proj_fwd(o2, wd, bd)
    close(rec.cpu(), rec2.cpu(), 1e-4 if dtype == torch.float32 else 2e-2, "fused vs stand-alone fwd")


# ------------------------------------------------------------------ loss / sampler / optimizer
@pytest.mark.parametrize("normalized", [True, False])
def test_patch_loss_fwd_bwd(ops, normalized):
    cfg = O.Config([8] * 6, [1] * 6, 8, (32, 48, 32))
    B = 2
    mask = mk_mask(B, cfg.fmap, cfg.len_keep)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    inp = rnd(B, 1, *cfg.input_size, seed=1) * 1.7 + 0.3
    rec = rnd(B, 1, *cfg.input_size, seed=2).requires_grad_(True)
    ip, rp = O.patchify(cfg, inp), O.patchify(cfg, rec)
    if normalized:
        loss, rl = O.forward_loss(ip, rp, mask)
        loss.backward()
    else:
        rl = O.teacher_patch_loss(ip, rp, mask)
    l2m, pm, pr, info = ops.patch_loss_fwd(inp[:, 0].contiguous().to(DEV), rec.detach()[:, 0].contiguous().to(DEV), mi, normalized)
    close(l2m.cpu(), rl.detach(), 1e-5, "per-patch l2")
    if normalized:
        assert abs(info[0].item() - loss.item()) <= 1e-5 * abs(loss.item())
        drec = ops.patch_loss_bwd(inp[:, 0].contiguous().to(DEV), rec.detach()[:, 0].contiguous().to(DEV), mi, pm, pr, info, None)
        close(drec.cpu(), rec.grad[:, 0], 1e-5, "drec")


def test_patch_loss_teacher_path_skips_visible_patches(ops):
    """want_loss=False (the teacher pass, pmean / prstd NULL): the kernel must not touch the VISIBLE patches at all -- their l2 is 0 by
    definition and their rec voxels may never have been written (the teacher's last decoder conv only produces the masked patches).
    NaN-filled visible patches must not leak into the result."""
    cfg = O.Config([8] * 6, [1] * 6, 8, (32, 48, 32))
    B = 2
    mask = mk_mask(B, cfg.fmap, cfg.len_keep)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    inp = rnd(B, 1, *cfg.input_size, seed=1) * 1.7 + 0.3
    rec = rnd(B, 1, *cfg.input_size, seed=2)
    rl = O.teacher_patch_loss(O.patchify(cfg, inp), O.patchify(cfg, rec), mask)
    vis = O.upsample_mask(mask, cfg.input_size)                  # True = visible
    rec_nan = torch.where(vis, torch.full_like(rec, float("nan")), rec)
    l2m, pm, pr, info = ops.patch_loss_fwd(inp[:, 0].contiguous().to(DEV), rec_nan[:, 0].contiguous().to(DEV), mi, False, want_loss=False)
    assert pm is None and pr is None and info is None
    assert torch.isfinite(l2m).all()
    close(l2m.cpu(), rl, 1e-5, "teacher per-patch l2 with unwritten visible patches")


def test_mask_sampler_golden_and_invariants(ops):
    f = load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    for tag in "abc":
        ep, tot, ll = (int(v) for v in f[f"gm_{tag}_ep"])
        m = ops.mask_sampler(torch.from_numpy(f[f"gm_{tag}_loss"]).to(DEV), torch.from_numpy(f[f"gm_{tag}_keys"]).to(DEV), cfg.len_keep, ll)
        assert np.array_equal(m.cpu().numpy().astype(bool).reshape(f[f"gm_{tag}_mask"].shape), f[f"gm_{tag}_mask"]), tag   # bit-exact
    # full-size invariants (192^3: L = 1728) + oracle equality on random draws
    cfg = O.Config([8] * 6, [1] * 6, 8, (192, 192, 192))
    B, L = 3, cfg.L
    g = torch.Generator().manual_seed(5)
    loss, keys = torch.rand(B, L, generator=g), torch.randn(B, L, generator=g)
    for ll in (0, 1, 259, L - cfg.len_keep):
        m = ops.mask_sampler(loss.to(DEV), keys.to(DEV), cfg.len_keep, ll).cpu().bool()
        assert (m.sum(1) == cfg.len_keep).all()
        if ll:
            hard = loss.argsort(1)[:, L - ll:]
            assert not m.gather(1, hard).any()
        assert torch.equal(m.view(B, 1, *cfg.fmap), O.generate_mask_from_keys(cfg, loss, keys, ll))
    # ragged / degenerate sizes
    for L2, keep in ((1, 1), (27, 11), (1000, 300)):
        m = ops.mask_sampler(torch.rand(2, L2).to(DEV), torch.rand(2, L2).to(DEV), keep, 0)
        assert (m.sum(1) == keep).all()


def test_adamw_ema_clip(ops):
    n = 4 * 50000
    p0, g = rnd(n, seed=1), rnd(n, seed=2) * 0.5
    P = {"w": p0.clone()}; E = {"w": p0.clone()}; state = {}
    pd, md, vd, ed = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), p0.clone().to(DEV)
    ss = torch.zeros(1, device=DEV, dtype=torch.float64); gn = torch.zeros(1, device=DEV)
    for step in range(1, 4):
        gs = {"w": (g * step).clone()}
        tot = O.clip_grad_norm(gs, 12.0)
        O.adamw_step(P, gs, state, step, 1e-3)
        O.ema_update(E, P, 0.99)
        gd = (g * step).to(DEV)
        ops.sumsq(gd, ss)
        ops.adamw_ema(pd, gd, md, vd, ed, n, 1e-3, (0.9, 0.999), 1e-8, 1e-5, step, ss, 12.0, 0.99, gn)
        assert abs(gn.item() - tot.item()) <= 1e-5 * tot.item()
        assert (pd.cpu() - P["w"]).abs().max().item() <= 2e-6
        assert (ed.cpu() - E["w"]).abs().max().item() <= 2e-6
    e2 = p0.clone().to(DEV)
    ops.ema(e2, pd, 0.5)
    assert (e2.cpu() - (0.5 * p0 + 0.5 * pd.cpu())).abs().max().item() <= 1e-6


def test_pack_unpack_roundtrip(ops):
    w = rnd(24, 16, 3, 3, 3, seed=1).to(DEV)
    for tc in (False, True):
        wp = ops.pack_weight(w, torch.float32, tc, False)
        cout, cin = (16, 24) if tc else (24, 16)
        assert wp.logical == (cout, cin) and wp.shape == (27, *ops.packed_dims(torch.float32, cout, cin))
        core = wp[:, :cout, :cin].contiguous()
        assert wp.abs().sum().item() == core.abs().sum().item()          # padding is zero
        back = torch.zeros_like(w)
        ops.unpack_grad(core, back, tc, False)
        assert torch.equal(back, w)
        ref = (w.permute(2, 3, 4, 1, 0) if tc else w.permute(2, 3, 4, 0, 1)).reshape(27, cout, cin)
        assert torch.equal(core, ref)
        wd = ops.pack_weight(w, torch.float32, tc, True)
        assert torch.equal(wd[:, :cin, :cout], ref.transpose(1, 2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sparse", [False, True])
def test_conv_epilogue_partials_feed_the_norm(ops, dtype, sparse):
    """Per-workgroup (sum, sumsq) partials written by the conv epilogue == a separate statistics pass over its output."""
    B, cin, cout, f = 2, 32, 48, (2, 2, 3)
    sp = tuple(v * 4 for v in f)
    mask = mk_mask(B, f, 5) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    x = q(rnd(B, cin, *sp, seed=1), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=2, scale=0.05), dtype)
    wp = ops.pack_weight(w.to(DEV), dtype, False, False)
    y, part = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, sp, 3, 1, in_mask=mi, in_bshift=2, out_mask=mi, out_bshift=2,
                         want_partials=True)
    st_a, st_b = ops.NormStats(cout, DEV), ops.NormStats(cout, DEV)
    part.reduce(sums=st_a.sums)
    ops.chan_stats(y, mi, 2, st_b)
    ref = st_b.sums.cpu().sum(0)                       # replicated accumulators
    assert (st_a.sums.cpu()[0] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    acc = torch.ones(cout, device=DEV)
    part.reduce(sum_accum=acc)
    assert (acc.cpu() - 1 - ref[:, 0].float()).abs().max().item() <= 1e-4 * ref[:, 0].abs().max().item() + 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout", [32, 64])
def test_conv_fused_epilogue(ops, dtype, cout):
    """y = act(conv * scale + shift + res): the eval-mode BatchNorm / skip add / ReLU6 folded into the conv store (teacher decoder)."""
    B, cin, S = 2, 40, (8, 8, 16)
    x = q(rnd(B, cin, *S, seed=21), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=22, scale=1.0 / np.sqrt(cin * 27)), dtype)
    sc, sh = rnd(cout, seed=23).abs() + 0.5, rnd(cout, seed=24)
    res = q(rnd(B, cout, *S, seed=25), dtype)
    conv = F.conv3d(x, w, None, padding=1)
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=False)
    want1 = torch.clamp(conv * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), 0, 6)
    y1 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_scale=sc.to(DEV), ep_shift=sh.to(DEV), ep_act=ops.ACT_RELU6)
    close(from_cl(y1), want1, TOL[dtype], "conv + bn + relu6")
    want2 = conv * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1) + res
    y2 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_scale=sc.to(DEV), ep_shift=sh.to(DEV), ep_res=to_cl(res, dtype))
    close(from_cl(y2), want2, TOL[dtype], "conv + bn + res")
    want3 = F.leaky_relu(conv + res, 0.01)
    y3 = ops.conv3d(ops.CONV_FWD, to_cl(x, dtype), wp, None, S, 3, 1, ep_res=to_cl(res, dtype), ep_act=ops.ACT_LRELU)
    close(from_cl(y3), want3, TOL[dtype], "conv + res + lrelu")


@pytest.mark.parametrize("rows,C", [(1000, 64), (37, 32), (4096, 32), (300, 512), (500, 96), (70, 256)])
def test_partials_reduce_paths(ops, rows, C):
    """[rows][C][2] per-workgroup partials -> per-channel double sums (+ float accumulate): strided v1 and coalesced v2 kernels."""
    part = ops.ConvPartials.__new__(ops.ConvPartials)
    part.rows, part.C = rows, C
    part.t = torch.randn(rows, C, 2, generator=torch.Generator().manual_seed(rows + C)).to(DEV)
    sums = torch.full((C, 2), 7.0, device=DEV, dtype=torch.float64)         # must be overwritten
    acc = torch.ones(C, device=DEV)
    part.reduce(sums=sums, sum_accum=acc)
    want = part.t.double().sum(0)
    assert torch.allclose(sums, want, rtol=0, atol=1e-9)
    assert torch.allclose(acc.double(), 1.0 + want[:, 0], rtol=0, atol=1e-4)


def test_abi_rejects_bad_arguments(ops):
    """Error convention of the C ABI: negative return codes for argument errors (surfaced as RuntimeError by the binding), no launch."""
    x = torch.zeros(1, 8, 8, 8, 12, device=DEV, dtype=torch.bfloat16)            # C % 8 != 0
    w = ops.pack_weight(torch.zeros(16, 16, 3, 3, 3, device=DEV), torch.bfloat16, False, False)
    w.logical = (16, 12)
    with pytest.raises(RuntimeError, match="am_conv3d failed with code -1"):
        ops.conv3d(ops.CONV_FWD, x, w, None, (8, 8, 8), 3, 1)
    x16 = torch.zeros(1, 8, 8, 8, 16, device=DEV, dtype=torch.bfloat16)
    w5 = ops.pack_weight(torch.zeros(16, 16, 3, 3, 3, device=DEV), torch.bfloat16, False, False)
    with pytest.raises(RuntimeError, match="am_conv3d failed with code -2"):      # ConvTranspose3d is k4 s2 only
        ops.conv3d(ops.CONVT_FWD, x16, w5, None, (16, 16, 16), 3, 2)
    with pytest.raises(RuntimeError, match="am_conv3d_wgrad failed with code -2"):
        ops.conv3d_wgrad(ops.CONV_FWD, x16, x16, 5, 1)
    with pytest.raises(RuntimeError, match="am_stem_conv_fwd failed with code -2"):   # the stem kernels need the 16^3 patch mask
        ops.stem_conv_fwd(torch.zeros(1, 16, 16, 16, device=DEV), torch.zeros(16, 1, 3, 3, 3, device=DEV), None, None, 4, torch.bfloat16)


# ------------------------------------------------------------------ resident-weight kernel for thin block-sparse layers (conv_rw.hip)
@pytest.mark.parametrize("case", [(32, 32), (16, 24), (8, 32)])
@pytest.mark.parametrize("sparse", [True, False])
def test_conv_rw_k3_s1_level0(ops, case, sparse):
    """STUNet stage-0 conv2 shape family (Cin, Cout <= 32, k3 s1, 16^3 patches, bf16): the persistent resident-weight kernel walking
    the active-patch list -- forward (+ statistics partials) and data gradient vs F.conv3d-then-mask (P/encoder3D.py:12-15)."""
    dtype = torch.bfloat16
    cin, cout = case
    B, f, bs = 2, (1, 2, 3), 4
    so = tuple(v << bs for v in f)               # (16, 32, 48)
    x = q(rnd(B, cin, *so, seed=41), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=42, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=43)
    dy = q(rnd(B, cout, *so, seed=44), dtype)
    mask = mk_mask(B, f, 3, seed=5) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    mo = O.upsample_mask(mask, so).float() if sparse else None
    if sparse:
        x, dy = x * mo, dy * mo
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, bias, padding=1)
    if sparse:
        yr = yr * mo
    yr.backward(dy)
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=False)
    xd = to_cl(x, dtype)
    if sparse:                                   # inactive voxels may hold anything: the kernel must never read them
        xd = torch.where(to_cl(mo.expand_as(x), dtype) > 0, xd, torch.full_like(xd, float("nan")))
    y, part = ops.conv3d(ops.CONV_FWD, xd, wp, bias.to(DEV), so, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, want_partials=True)
    assert part.rows <= 256                      # ONE row per persistent workgroup (the generic kernel leaves one per brick)
    close(from_cl(y), yr.detach(), TOL[dtype], "rw conv fwd", mo)
    st_a, st_b = ops.NormStats(cout, DEV), ops.NormStats(cout, DEV)
    part.reduce(sums=st_a.sums)
    ops.chan_stats(y, mi, bs if sparse else 0, st_b)
    ref = st_b.sums.cpu().sum(0)
    assert (st_a.sums.cpu()[0] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    wpd = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=True)
    dx = ops.conv3d(ops.CONV_DGRAD, to_cl(dy, dtype), wpd, None, so, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs)
    close(from_cl(dx), xr.grad, TOL[dtype], "rw conv dgrad", mo)


@pytest.mark.parametrize("case", [(32, 64), (16, 40), (32, 32)])
@pytest.mark.parametrize("sparse", [True, False])
def test_conv_rw_k3_s2_level1(ops, case, sparse):
    """STUNet stage-1 conv1 shape family (Cin <= 32 -> Cout <= 64, k3 stride 2, fine 16^3 -> coarse 8^3 patches, bf16): eight parity
    sub-lattice units per brick on the resident-weight kernel."""
    dtype = torch.bfloat16
    cin, cout = case
    B, f = 2, (2, 1, 3)
    si, so = tuple(v * 16 for v in f), tuple(v * 8 for v in f)
    x = q(rnd(B, cin, *si, seed=51), dtype)
    w = q(rnd(cout, cin, 3, 3, 3, seed=52, scale=1.0 / np.sqrt(cin * 27)), dtype)
    bias = rnd(cout, seed=53)
    mask = mk_mask(B, f, 3, seed=6) if sparse else None
    mi = ops.MaskInfo.from_bool(mask, DEV) if sparse else None
    if sparse:
        x = x * O.upsample_mask(mask, si).float()
    yr = F.conv3d(x, w, bias, stride=2, padding=1)
    mo = O.upsample_mask(mask, so).float() if sparse else None
    if sparse:
        yr = yr * mo
    wp = ops.pack_weight(w.to(DEV), dtype, transposed_conv=False, for_dgrad=False)
    xd = to_cl(x, dtype)
    if sparse:
        xd = torch.where(to_cl(O.upsample_mask(mask, si).float().expand_as(x), dtype) > 0, xd, torch.full_like(xd, float("nan")))
    y, part = ops.conv3d(ops.CONV_FWD, xd, wp, bias.to(DEV), so, 3, 2, in_mask=mi, in_bshift=4, out_mask=mi, out_bshift=3, want_partials=True)
    assert part.rows <= 256
    close(from_cl(y), yr, TOL[dtype], "rw conv s2 fwd", mo)
    st_a, st_b = ops.NormStats(cout, DEV), ops.NormStats(cout, DEV)
    part.reduce(sums=st_a.sums)
    ops.chan_stats(y, mi, 3 if sparse else 0, st_b)
    ref = st_b.sums.cpu().sum(0)
    assert (st_a.sums.cpu()[0] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_row_walk_kernels_beyond_the_linear_decode_limit(ops):
    """B * D * H * W * max(D, H, W) >= 2^32 (batch 16 at 128^3): the linear kernels' reciprocal-multiply voxel decode stops being exact
    there and they refuse (-4), but the row-walk kernels (active-patch list) have no such limit -- the norm passes of a block-sparse
    tensor must work (bench.py's default batch).  Reference: torch on the same device, a thin 8-channel tensor."""
    B, C, S, f = 16, 8, 128, 8
    g = torch.Generator().manual_seed(0)
    mask = torch.zeros(B, f ** 3, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(f ** 3, generator=g)[:205]] = True
    mask = mask.view(B, 1, f, f, f)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    x = torch.randn(B, S, S, S, C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).to(torch.bfloat16)
    mf = mask.to(DEV).repeat_interleave(16, 2).repeat_interleave(16, 3).repeat_interleave(16, 4)[:, 0, ..., None]     # (B,S,S,S,1)
    st = ops.NormStats(C, DEV)
    ops.chan_stats(x, mi, 4, st)
    st.count_host = float(B * 205 * 4096)
    gam, bet = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    ops.norm_finalize(st, gam, bet, 1e-5)
    y = ops.norm_apply(x, st, ops.ACT_LRELU, mi, 4)
    xa = x.float()[mf.expand_as(x)].view(-1, C)
    mean, var = xa.mean(0), xa.var(0, unbiased=False)
    want = torch.nn.functional.leaky_relu((x.float() - mean) * torch.rsqrt(var + 1e-5) * gam + bet, 0.01)
    err = torch.where(mf.expand_as(y), (y.float() - want).abs(), torch.zeros((), device=DEV)).max().item()     # inactive voxels: don't-care bits
    assert err <= 2e-2 * want.abs().max().item(), err
    with pytest.raises(RuntimeError, match="am_chan_sum failed with code -4"):      # a linear-walk kernel still refuses
        ops.chan_sum(x, mi, 4, torch.zeros(C, device=DEV))
