"""CPU: the layer-zoo oracle (oracle/layers_oracle.py) against golden vectors produced by the reference's own classes
(tests/golden/layers_tiny.npz, made by tests/golden/make_layer_fixtures.py from P/encoder3D.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import layers_oracle as LO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layers_tiny.npz"))
ACTIVE = torch.from_numpy(G["active"])


def t(name):
    """a fixture array; the common input x and the cotangents g are regenerated from their seeds (tests/golden/make_layer_fixtures.py)."""
    if name in G.files:
        return torch.from_numpy(G[name])
    import zlib
    case, kind = name.rsplit(".", 1)
    if kind == "x":
        from oracle import layers_oracle as LO
        x = torch.from_numpy(np.random.RandomState(1).standard_normal((2, 16, 8, 8, 8)).astype(np.float32))
        return x * LO.up(torch.from_numpy(G["active"]), (8, 8, 8)).float()
    if kind == "g":
        return torch.from_numpy(np.random.RandomState(zlib.crc32(case.encode()) % 1000 + 7).standard_normal(G[case + ".y"].shape).astype(np.float32))
    raise KeyError(name)


def check(name, fn, params=(), tol=2e-6, cl=False):
    x = t(name + ".x").clone().requires_grad_(True)
    ps = {n: t(f"{name}.p.{n}").clone().requires_grad_(True) for n in params}
    y = fn(x, ps)
    assert torch.allclose(y, t(name + ".y"), rtol=1e-5, atol=tol), (name, (y - t(name + ".y")).abs().max())
    (y * t(name + ".g")).sum().backward()
    # gradients are compared on the ACTIVE voxels of x: at inactive ones the reference lets a gradient through (it never masks the input
    # of a layer, only the output) that the producing sparse layer's own output mask then discards -- don't-care values
    m = LO.up(ACTIVE, x.shape[1:4] if cl else x.shape[2:]).float()
    m = m.permute(0, 2, 3, 4, 1) if cl else m
    assert torch.allclose(x.grad * m, t(name + ".dx") * m, rtol=1e-4, atol=1e-5), (name, "dx", ((x.grad - t(name + ".dx")) * m).abs().max())
    for n, p in ps.items():
        want = t(f"{name}.d.{n}")
        assert torch.allclose(p.grad, want, rtol=1e-4, atol=2e-5 * max(1.0, want.abs().max().item())), (name, n, (p.grad - want).abs().max())


@pytest.mark.parametrize("name,k,s,p", [("maxpool_k2", 2, 2, 0), ("maxpool_k3s2p1", 3, 2, 1)])
def test_max_pool(name, k, s, p):
    check(name, lambda x, ps: LO.sparse_max_pool(x, ACTIVE, k, s, p))


@pytest.mark.parametrize("name,k,s,p,cip", [("avgpool_k2", 2, 2, 0, True), ("avgpool_k3s2p1", 3, 2, 1, True), ("avgpool_k3s2p1_nopad", 3, 2, 1, False)])
def test_avg_pool(name, k, s, p, cip):
    check(name, lambda x, ps: LO.sparse_avg_pool(x, ACTIVE, k, s, p, cip))


def test_batch_norm_train_and_eval():
    rm, rv = t("bn.init.running_mean").clone(), t("bn.init.running_var").clone()
    check("bn_train", lambda x, ps: LO.sparse_batch_norm(x, ACTIVE, ps["weight"], ps["bias"], rm, rv, True), ("weight", "bias"))
    assert torch.allclose(rm, t("bn_train.b.running_mean"), atol=1e-6) and torch.allclose(rv, t("bn_train.b.running_var"), atol=1e-6)
    check("bn_eval", lambda x, ps: LO.sparse_batch_norm(x, ACTIVE, ps["weight"], ps["bias"], rm, rv, False), ("weight", "bias"))


@pytest.mark.parametrize("name,groups", [("gn_g2", 2), ("gn_gC", 16)])
def test_group_norm(name, groups):
    check(name, lambda x, ps: LO.sparse_group_norm(x, ACTIVE, groups, ps["weight"], ps["bias"]), ("weight", "bias"))


def test_layer_norm_both_formats():
    check("ln_cf", lambda x, ps: LO.sparse_layer_norm(x, ACTIVE, ps["weight"], ps["bias"]), ("weight", "bias"))
    check("ln_cl", lambda x, ps: LO.sparse_layer_norm(x.permute(0, 4, 1, 2, 3), ACTIVE, ps["weight"], ps["bias"]).permute(0, 2, 3, 4, 1),
          ("weight", "bias"), cl=True)


def test_grn():
    check("grn", lambda x, ps: LO.sparse_grn(x, ACTIVE, ps["gamma"], ps["beta"]), ("gamma", "beta"), cl=True)


def test_adaptive_avg_pool():
    check("adaptive_avg", lambda x, ps: LO.sparse_adaptive_avg_pool(x, ACTIVE))


@pytest.mark.parametrize("name,stride,groups", [("dwconv7", 1, 16), ("dwconv3", 1, 16), ("conv3s2", 2, 1)])
def test_sparse_conv(name, stride, groups):
    check(name, lambda x, ps: LO.sparse_conv3d(x, ACTIVE, ps["weight"], ps["bias"], stride, groups), ("weight", "bias"), tol=1e-5)


def test_convnext_block():
    names = ("dwconv.weight", "dwconv.bias", "norm.weight", "norm.bias", "pwconv1.weight", "pwconv1.bias", "pwconv2.weight", "pwconv2.bias", "gamma")
    check("convnext", lambda x, ps: LO.convnext_block(x, ACTIVE, ps), names, tol=1e-5)


@pytest.mark.parametrize("name,k", [("dwconv7s2", 7), ("dwconv3s2", 3)])
def test_strided_depthwise_conv(name, k):
    check(name, lambda x, ps: LO.sparse_conv3d(x, ACTIVE, ps["weight"], ps["bias"], 2, 16), ("weight", "bias"), tol=1e-5)


def mednext_inputs():
    act = torch.from_numpy(G["mednext.active"])
    x = torch.from_numpy(np.random.RandomState(3).standard_normal((1, 1, 32, 32, 32)).astype(np.float32)) * LO.up(act, (32, 32, 32)).float()
    gs = [torch.from_numpy(np.random.RandomState(50 + i).standard_normal(G[f"mednext.map{i}"].shape).astype(np.float32)) for i in range(5)]
    return act, x, gs


def test_mednext_encoder_under_spark():
    """a tiny MedNeXt (P/MedNeXt_head.py) converted by the reference's SparseEncoder: 5 hierarchical maps and all 82 parameter gradients."""
    act, x, gs = mednext_inputs()
    p = {k[len("mednext.p."):]: t(k).clone().requires_grad_(True) for k in G.files if k.startswith("mednext.p.") and "dummy" not in k}
    maps = LO.mednext_encoder(x, act, p)
    for i, m in enumerate(maps):
        assert torch.allclose(m, t(f"mednext.map{i}"), rtol=1e-4, atol=2e-5), (i, (m - t(f"mednext.map{i}")).abs().max())
    sum((m * g).sum() for m, g in zip(maps, gs)).backward()
    for k, v in p.items():
        want = t("mednext.d." + k)
        assert torch.allclose(v.grad, want, rtol=1e-3, atol=2e-4 * max(1.0, want.abs().max().item())), (k, (v.grad - want).abs().max())
