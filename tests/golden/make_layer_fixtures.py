"""Golden vectors of the sparse layer zoo, produced by the REFERENCE's own classes (P/encoder3D.py) -- run in the build container only:

    python tests/golden/make_layer_fixtures.py        # -> tests/golden/layers_tiny.npz

timm (absent, used for DropPath only) is replaced by an in-memory stand-in; the reference source is imported where it lies and is not
copied.  Every case stores the seeded inputs, the parameters, the output and the gradients of sum(out * g)."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/nnunetv2/training/nnUNetTrainer/variants/pretrain"
tm, tml = types.ModuleType("timm"), types.ModuleType("timm.models.layers")
tml.DropPath = torch.nn.Identity
tml.trunc_normal_ = torch.nn.init.trunc_normal_
tmm = types.ModuleType("timm.models"); tmm.layers = tml; tm.models = tmm
sys.modules.update({"timm": tm, "timm.models": tmm, "timm.models.layers": tml})
sys.path.insert(0, REF)
import encoder3D as E  # noqa: E402

torch.manual_seed(0)
out = {}
B, f, S, C = 2, 2, 8, 16


def rnd(*shape, seed, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32)) * scale


active = torch.zeros(B, 1, f, f, f, dtype=torch.bool)
active.view(B, -1)[0, [0, 3, 5]] = True
active.view(B, -1)[1, [1, 2, 6, 7]] = True
out["active"] = active.numpy()
E._cur_active = active
mfull = active.repeat_interleave(S // f, 2).repeat_interleave(S // f, 3).repeat_interleave(S // f, 4).float()


def run(name, module, x, params=None, train=True):
    module.train(train)
    x = x.clone().requires_grad_(True)
    y = module(x)
    g = rnd(*y.shape, seed=hash(name) % 1000 + 7)
    (y * g).sum().backward()
    out[name + ".x"], out[name + ".y"], out[name + ".g"], out[name + ".dx"] = x.detach().numpy(), y.detach().numpy(), g.numpy(), x.grad.numpy()
    for n, p in module.named_parameters():
        out[f"{name}.p.{n}"] = p.detach().numpy()
        if p.grad is not None:
            out[f"{name}.d.{n}"] = p.grad.numpy()
    for n, b_ in module.named_buffers():
        out[f"{name}.b.{n}"] = b_.detach().numpy()


x = rnd(B, C, S, S, S, seed=1) * mfull
# pooling
run("maxpool_k2", E.SparseMaxPooling(2, 2), x)
run("maxpool_k3s2p1", E.SparseMaxPooling(3, 2, 1), x)
run("avgpool_k2", E.SparseAvgPooling(2, 2), x)
run("avgpool_k3s2p1", E.SparseAvgPooling(3, 2, 1), x)
run("avgpool_k3s2p1_nopad", E.SparseAvgPooling(3, 2, 1, count_include_pad=False), x)
# batch norm (two training calls: running statistics after the second), then eval
bn = E.SparseBatchNorm3d(C)
bn.weight.data = 1 + 0.3 * rnd(C, seed=21); bn.bias.data = 0.2 * rnd(C, seed=22)
out["bn.init.running_mean"], out["bn.init.running_var"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
run("bn_train", bn, x)
bn.zero_grad()
run("bn_eval", bn, x, train=False)
# group norm: 2 groups of 8, and one channel per group
gn = E.SparseGroupNorm(2, C); gn.weight.data = 1 + 0.3 * rnd(C, seed=23); gn.bias.data = 0.2 * rnd(C, seed=24)
run("gn_g2", gn, x)
gn1 = E.SparseGroupNorm(C, C); gn1.weight.data = 1 + 0.3 * rnd(C, seed=25); gn1.bias.data = 0.2 * rnd(C, seed=26)
run("gn_gC", gn1, x)
# layer norm, both data formats
ln = E.SparseConvNeXtLayerNorm(C, data_format="channels_first"); ln.weight.data = 1 + 0.3 * rnd(C, seed=27); ln.bias.data = 0.2 * rnd(C, seed=28)
run("ln_cf", ln, x)
lnl = E.SparseConvNeXtLayerNorm(C); lnl.weight.data = ln.weight.data.clone(); lnl.bias.data = ln.bias.data.clone()
run("ln_cl", lnl, x.permute(0, 2, 3, 4, 1).contiguous())
# GRN sparse branch: the reference takes the mask resolution from x.shape[2:5] of a channels-last tensor -> C must equal the spatial size
grn = E.SparseGRN(S); grn.gamma.data = 0.5 * rnd(1, S, seed=29); grn.beta.data = 0.2 * rnd(1, S, seed=30)
xg = (rnd(B, S, S, S, S, seed=2) * mfull).permute(0, 2, 3, 4, 1).contiguous()
run("grn", grn, xg)
# adaptive average pooling
run("adaptive_avg", E.SparseAdaptiveAvgPooling((1, 1, 1)), x)
# sparse convs: depthwise 7^3 / 3^3 and a dense 3^3 stride 2
dw = E.SparseConv3d(C, C, 7, padding=3, groups=C); run("dwconv7", dw, x)
dw3 = E.SparseConv3d(C, C, 3, padding=1, groups=C); run("dwconv3", dw3, x)
cv = E.SparseConv3d(C, 24, 3, stride=2, padding=1); run("conv3s2", cv, x)
# ConvNeXt block (the converter turns its dwconv into a SparseConv3d)
blk = E.SparseEncoder.dense_model_to_sparse(E.SparseConvNeXtBlock(C, layer_scale_init_value=0.5, ks=7))
blk.gamma.data = 0.5 + 0.2 * rnd(C, seed=31)
run("convnext", blk, x)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "layers_tiny.npz"), **out)
print("wrote", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")
