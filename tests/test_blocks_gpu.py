"""GPU, bf16 storage: ONE block of the network at production channel counts -- forward AND every gradient -- against the oracle's ideal
bf16-storage evaluation of the same block (`oracle.storage("bf16")`: fp32 math, every stored C > 1 tensor / gradient / MFMA weight copy
rounded to bf16).

Why this file exists: the op tests are tight in bf16 (2e-2 of max) and the whole-network bf16 tests can only
assert "as far from fp32 as an ideal bf16 evaluation is" (0.3-0.6 per tensor: LeakyReLU / ReLU6 gate flips accumulate with depth).  A
block is one norm + activation deep, gate flips cannot be the excuse there: a bf16-only bug that bends a block's gradient by 10 % fails
these bounds (relative L2 <= 3e-2, cosine >= 0.999 per tensor).

Blocks: BasicResBlock (P/STUNet_head.py:96-103 under P/encoder3D.py:301-329), block-sparse, stride 1 (identity shortcut) and stride 2
(1x1x1 stride-2 shortcut); UNetBlock (P/decoder3D.py:13-29: ConvT k4 s2, conv-BN-ReLU6-conv-BN, + skip).  They run through the engine's own
block functions (`engine._enc_block` / `_enc_block_backward`, `_dec_block` / `_dec_block_backward`), i.e. the code the training step runs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import anatomask_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
REL_L2, COS = 3e-2, 0.999


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def qb(t):
    return t.to(BF).float()


def to_cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().to(device=DEV, dtype=BF)


def from_cl(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def agree(got, want, what, keep=None):
    got, want = got.double(), want.double()
    if keep is not None:
        k = keep.expand_as(want) > 0
        got, want = torch.where(k, got, torch.zeros_like(got)), torch.where(k, want, torch.zeros_like(want))
    n = want.norm().item()
    assert n > 0, what
    rel = (got - want).norm().item() / n
    cos = (got * want).sum().item() / (got.norm().item() * n + 1e-300)
    assert rel <= REL_L2 and cos >= COS, f"{what}: rel-L2 {rel:.3e} (<= {REL_L2}), cos {cos:.6f} (>= {COS})"
    return rel, cos


def mk_mask(B, f, keep, seed):
    g = torch.Generator().manual_seed(seed)
    L = f[0] * f[1] * f[2]
    idx = torch.rand(B, L, generator=g).argsort(1)[:, :keep]
    return torch.zeros(B, L, dtype=torch.bool).scatter_(1, idx, True).view(B, 1, *f)


@pytest.mark.parametrize("case", [(1, 1, 64, 64, 1, (3, 3, 3)),      # stage 1, second block: 64 -> 64, stride 1, identity shortcut, 8^3 patches
                                  (2, 0, 64, 128, 2, (3, 3, 4)),     # stage 2, first block: 64 -> 128, stride 2, 1x1x1 stride-2 shortcut, 4^3 patches
                                  (3, 0, 128, 256, 2, (4, 4, 4))])   # stage 3, first block: 128 -> 256, stride 2, 2^3 patches (voxel-list gather kernels)
def test_basic_res_block_bf16_forward_and_gradients(case):
    from anatomask_amd import engine, ops
    s, b, cin, cout, stride, f = case
    B, bs = 2, 4 - s
    so = tuple(v << bs for v in f)
    si = tuple(v * stride for v in so)
    p = f"{engine.ENC}.{s}.{b}"
    first = b == 0
    mask = mk_mask(B, f, max(2, int(0.4 * f[0] * f[1] * f[2])), seed=100 + s)
    mo, mi_in = O.upsample_mask(mask, so).float(), O.upsample_mask(mask, si).float()
    sc = 1.0 / np.sqrt(27)
    P = {f"{p}.conv1.weight": qb(rnd(cout, cin, 3, 3, 3, seed=1, scale=sc / np.sqrt(cin))), f"{p}.conv1.bias": rnd(cout, seed=2, scale=0.1),
         f"{p}.norm1.weight": 1 + 0.2 * rnd(cout, seed=3), f"{p}.norm1.bias": 0.2 * rnd(cout, seed=4),
         f"{p}.conv2.weight": qb(rnd(cout, cout, 3, 3, 3, seed=5, scale=sc / np.sqrt(cout))), f"{p}.conv2.bias": rnd(cout, seed=6, scale=0.1),
         f"{p}.norm2.weight": 1 + 0.2 * rnd(cout, seed=7), f"{p}.norm2.bias": 0.2 * rnd(cout, seed=8)}
    if first:
        P[f"{p}.conv3.weight"] = qb(rnd(cout, cin, 1, 1, 1, seed=9, scale=1.0 / np.sqrt(cin)))
        P[f"{p}.conv3.bias"] = rnd(cout, seed=10, scale=0.1)
    x = qb(rnd(B, cin, *si, seed=11)) * mi_in
    gout = qb(rnd(B, cout, *so, seed=12)) * mo
    # ---- oracle: ideal bf16-storage evaluation of the block, autograd
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    with O.storage("bf16"):
        yr = O.basic_res_block(Pr, p, xr, stride, mask, has_sc=first)
        (yr * gout).sum().backward()
    # ---- HIP
    W = {k: v.to(DEV) for k, v in P.items()}
    G = {k: torch.zeros_like(v) for k, v in W.items()}
    pk = engine.PackCache(BF)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    counts = engine._counts(mi, range(5))
    out, rec = engine._enc_block(W, pk, None, mi, counts, so, s, b, to_cl(x))
    agree(from_cl(out), yr.detach(), "block output", mo)
    gx = engine._enc_block_backward(W, G, pk, None, mi, rec, to_cl(gout), None)
    engine._join_side(torch.device(DEV))
    torch.cuda.synchronize()
    agree(from_cl(gx), xr.grad, "gradient wrt the block input", mi_in)
    for k in P:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            # a conv bias under a norm: the analytic gradient is zero, what both sides hold is rounding noise of ~1e-3 of the weight gradients
            assert G[k].abs().max().item() <= 2e-2 * G[k.replace("bias", "weight")].abs().max().item(), k
            continue
        agree(G[k].cpu(), Pr[k].grad, f"gradient of {k}")


@pytest.mark.parametrize("case", [(2, 128, 64, (4, 4, 4), True, 2), (3, 64, 32, (8, 8, 8), False, 2),
                                  # round 5: the last block at a size where every launch takes the persistent LDS-DMA kernels -- transposed conv
                                  # (conv_k3_kernel<4,.,.,true>), 64 -> 64 with statistics, 64 -> 32 (32-channel tile), both data gradients, and the
                                  # 8-wave weight gradients with 64- and 32-wide cy tiles (asserted below)
                                  (3, 64, 32, (32, 32, 32), False, 4)])
def test_unet_block_bf16_forward_and_gradients(case):
    """UNetBlock i: ConvT(c -> c) k4 s2, conv3(c -> c)-BN-ReLU6, conv3(c -> c_out)-BN, + the next level's densified map (P/decoder3D.py:13-29,59)."""
    from anatomask_amd import engine, hip, ops
    i, c, c_out, si, with_skip, B = case
    if si[0] >= 32:
        L = hip.lib()._lib
        assert L.am_conv3d_wgrad_uses_k3(ops.CONV_FWD, hip.DT_BF16, 3, 1, B, *(2 * v for v in si), c, c, 0, 0) == 1
        assert L.am_conv3d_wgrad_uses_k3(ops.CONV_FWD, hip.DT_BF16, 3, 1, B, *(2 * v for v in si), c, c_out, 0, 0) == 1
    so = tuple(2 * v for v in si)
    q = f"{engine.DEC}.{i}"
    P = {f"{q}.up_sample.weight": qb(rnd(c, c, 4, 4, 4, seed=21, scale=1.0 / np.sqrt(8 * c))), f"{q}.up_sample.bias": rnd(c, seed=22, scale=0.1),
         f"{q}.conv.0.weight": qb(rnd(c, c, 3, 3, 3, seed=23, scale=1.0 / np.sqrt(27 * c))),
         f"{q}.conv.1.weight": 1 + 0.2 * rnd(c, seed=24), f"{q}.conv.1.bias": 0.2 * rnd(c, seed=25),
         f"{q}.conv.3.weight": qb(rnd(c_out, c, 3, 3, 3, seed=26, scale=1.0 / np.sqrt(27 * c))),
         f"{q}.conv.4.weight": 1 + 0.2 * rnd(c_out, seed=27), f"{q}.conv.4.bias": 0.2 * rnd(c_out, seed=28)}
    bufs = {f"{q}.conv.1.running_mean": torch.zeros(c), f"{q}.conv.1.running_var": torch.ones(c), f"{q}.conv.1.num_batches_tracked": torch.zeros((), dtype=torch.int64),
            f"{q}.conv.4.running_mean": torch.zeros(c_out), f"{q}.conv.4.running_var": torch.ones(c_out), f"{q}.conv.4.num_batches_tracked": torch.zeros((), dtype=torch.int64)}
    x = qb(rnd(B, c, *si, seed=31))
    nxt = qb(rnd(B, c_out, *so, seed=32)) if with_skip else None
    gout = qb(rnd(B, c_out, *so, seed=33))
    # ---- oracle (the body of O.decoder_forward's loop for one block, same storage points)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    allp = dict(Pr, **bufs)
    xr = x.clone().requires_grad_(True)
    with O.storage("bf16"):
        u = O._q(F.conv_transpose3d(xr, O._qw(Pr[f"{q}.up_sample.weight"], xr), Pr[f"{q}.up_sample.bias"], stride=2, padding=1))
        c1 = O._q(F.conv3d(u, O._qw(Pr[f"{q}.conv.0.weight"], u), None, padding=1))
        r = O._q(F.relu6(O.batch_norm3d(allp, f"{q}.conv.1", c1, True, None)))
        c2 = O._q(F.conv3d(r, O._qw(Pr[f"{q}.conv.3.weight"], r), None, padding=1))
        o = O.batch_norm3d(allp, f"{q}.conv.4", c2, True, None)
        o = O._q(o + nxt) if with_skip else O._q(o)            # (the block output is a stored tensor on both sides)
        (o * gout).sum().backward()
    # ---- HIP
    W = {k: v.to(DEV) for k, v in dict(P, **bufs).items()}
    G = {k: torch.zeros_like(v) for k, v in W.items() if k in P}
    pk = engine.PackCache(BF)
    out, rec = engine._dec_block(W, pk, i, to_cl(x), to_cl(nxt) if with_skip else None, True)
    agree(from_cl(out), o.detach(), "block output")
    gin = engine._dec_block_backward(W, G, pk, rec, to_cl(gout), None, None)
    engine._join_side(torch.device(DEV))
    torch.cuda.synchronize()
    agree(from_cl(gin), xr.grad, "gradient wrt the block input")
    for k in P:
        agree(G[k].cpu(), Pr[k].grad, f"gradient of {k}")


def test_stem_block_bf16_forward_and_gradients():
    """The Cin = 1 stem block (stage 0, first block: conv 1 -> 32, 1x1x1 shortcut from the VOLUME, 16^3 patches) on a smooth CT-like volume
    (oracle.smooth_volume: what the full-step tests feed).  The volume and the stem's fp32 weights are not bf16-storage tensors: the oracle's
    emulation leaves them alone, and the matrix-core stem kernels take them as hi + lo bf16 parts.  Round 5 found the full-step bound of the
    stem weight's gradient ON its limit because those kernels rounded the volume to bf16 once (8 bits: as much as the voxel-to-voxel
    differences a 3^3 filter sees); no block test covered this block."""
    from anatomask_amd import engine, ops
    s, b, cout, f, B = 0, 0, 32, (2, 2, 3), 2
    bs = 4
    so = tuple(v << bs for v in f)
    p = f"{engine.ENC}.{s}.{b}"
    mask = mk_mask(B, f, 5, seed=100)
    mo = O.upsample_mask(mask, so).float()
    P = {f"{p}.conv1.weight": rnd(cout, 1, 3, 3, 3, seed=1, scale=1.0 / np.sqrt(27)), f"{p}.conv1.bias": rnd(cout, seed=2, scale=0.1),
         f"{p}.norm1.weight": 1 + 0.2 * rnd(cout, seed=3), f"{p}.norm1.bias": 0.2 * rnd(cout, seed=4),
         f"{p}.conv2.weight": qb(rnd(cout, cout, 3, 3, 3, seed=5, scale=1.0 / np.sqrt(27 * cout))), f"{p}.conv2.bias": rnd(cout, seed=6, scale=0.1),
         f"{p}.norm2.weight": 1 + 0.2 * rnd(cout, seed=7), f"{p}.norm2.bias": 0.2 * rnd(cout, seed=8),
         f"{p}.conv3.weight": rnd(cout, 1, 1, 1, 1, seed=9), f"{p}.conv3.bias": rnd(cout, seed=10, scale=0.1)}
    x = O.smooth_volume(B, so, 9) * mo                                    # (B, 1, D, H, W) fp32, NOT bf16-representable
    assert (x - qb(x)).abs().max() > 1e-4
    gout = qb(rnd(B, cout, *so, seed=12)) * mo
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.storage("bf16"):
        yr = O.basic_res_block(Pr, p, x, 1, mask, has_sc=True)
        (yr * gout).sum().backward()
    W = {k: v.to(DEV) for k, v in P.items()}
    G = {k: torch.zeros_like(v) for k, v in W.items()}
    pk = engine.PackCache(BF)
    mi = ops.MaskInfo.from_bool(mask, DEV)
    counts = engine._counts(mi, range(5))
    inp = x[:, 0].contiguous().to(DEV)
    out, rec = engine._enc_block(W, pk, inp, mi, counts, so, s, b, None)
    agree(from_cl(out), yr.detach(), "stem block output", mo)
    assert engine._enc_block_backward(W, G, pk, inp, mi, rec, to_cl(gout), None) is None
    engine._join_side(torch.device(DEV))
    torch.cuda.synchronize()
    report = []
    for k in P:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            assert G[k].abs().max().item() <= 2e-2 * G[k.replace("bias", "weight")].abs().max().item(), k
            continue
        g, w = G[k].cpu().double(), Pr[k].grad.double()
        report.append((k.split(".")[-2] + "." + k.split(".")[-1], float((g - w).norm() / w.norm()), float((g * w).sum() / (g.norm() * w.norm()))))
    print("stem block, gradients vs the ideal emulation (rel-L2 / cos): " + ", ".join(f"{n} {r:.2e}/{c:.5f}" for n, r, c in report))
    # bound: the emulation itself sits 3.1e-2 .. 5.8e-2 from fp32 on this block and two rounding realisations of it (the volume scaled by
    # 1 + 2^-10) differ by 2.6e-2 .. 7.0e-2 per tensor (CPU, round 5: a smooth volume puts many norm1 outputs near the LeakyReLU gate);
    # measured HIP-vs-emulation 1.9e-2 .. 3.2e-2
    for n, r, c in report:
        assert r <= 6e-2 and c >= 0.998, (n, r, c)
    # what the test is for: the same launches fed the volume ROUNDED to bf16 (what the matrix-core stem kernels did to it in rounds 3-4)
    G2 = {k: torch.zeros_like(v) for k, v in W.items()}
    inp_r = qb(x)[:, 0].contiguous().to(DEV)
    out2, rec2 = engine._enc_block(W, pk, inp_r, mi, counts, so, s, b, None)
    engine._enc_block_backward(W, G2, pk, inp_r, mi, rec2, to_cl(gout), None)
    engine._join_side(torch.device(DEV))
    torch.cuda.synchronize()
    k1 = f"{p}.conv1.weight"
    w = Pr[k1].grad.double()
    r_round = float((G2[k1].cpu().double() - w).norm() / w.norm())
    print(f"stem weight gradient with the volume rounded to bf16 first: rel-L2 {r_round:.2e} (unrounded: {report[0][1]:.2e})")
    # measured 5.8e-2 against 2.1e-2: the stem weight's own bound sits between the two
    assert report[0][0] == "conv1.weight" and report[0][1] <= 4e-2 and r_round >= 1.5 * report[0][1], (report[0], r_round)
