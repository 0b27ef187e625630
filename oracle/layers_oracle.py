"""CPU oracle of the sparse layer zoo -- TEST INFRASTRUCTURE ONLY: imported by tests/ and never by the product path.

Functional torch-CPU restatement of the mask-aware layers of P/encoder3D.py (= nnunetv2/training/nnUNetTrainer/variants/pretrain/
encoder3D.py); every function cites the lines it follows.  Pinned: tests/golden/layers_tiny.npz holds inputs / outputs / gradients
produced by the reference's own classes (tests/golden/make_layer_fixtures.py, run in the build container) and
tests/test_layers_oracle.py checks these functions against them.

Conventions: x is NCDHW fp32 unless stated, `active` is the bool patch mask (B,1,f,f,f) (encoder3D._cur_active), inactive voxels of
x are read as zeros (the reference's tensors hold zeros there by construction) and outputs are zero there.
"""
import torch
import torch.nn.functional as F


def up(active: torch.Tensor, size) -> torch.Tensor:
    """_get_active_ex_or_ii(returning_active_ex=True), encoder3D.py:7-10."""
    r = [s // a for s, a in zip(size, active.shape[-3:])]
    return active.repeat_interleave(r[0], 2).repeat_interleave(r[1], 3).repeat_interleave(r[2], 4)


def sparse_conv3d(x, active, weight, bias, stride=1, groups=1):
    """SparseConv3d = sp_conv_forward :12-15,27-28 ('same' padding k//2)."""
    k = weight.shape[-1]
    y = F.conv3d(x * up(active, x.shape[2:]), weight, bias, stride=stride, padding=k // 2, groups=groups)
    return y * up(active, y.shape[2:])


def sparse_max_pool(x, active, k, s, p=0):
    """SparseMaxPooling :31-32."""
    y = F.max_pool3d(x * up(active, x.shape[2:]), k, s, p)
    return y * up(active, y.shape[2:])


def sparse_avg_pool(x, active, k, s, p=0, count_include_pad=True):
    """SparseAvgPooling :35-36."""
    y = F.avg_pool3d(x * up(active, x.shape[2:]), k, s, p, count_include_pad=count_include_pad)
    return y * up(active, y.shape[2:])


def _on_active_rows(x, active, fn):
    """the gather / scatter of sp_bn_forward :17-25 and its siblings: fn maps the (N_active, C) matrix."""
    ii = up(active, x.shape[2:]).squeeze(1).nonzero(as_tuple=True)
    bhwdc = x.permute(0, 2, 3, 4, 1)
    out = torch.zeros_like(bhwdc)
    out[ii] = fn(bhwdc[ii])
    return out.permute(0, 4, 1, 2, 3)


def sparse_batch_norm(x, active, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """SparseBatchNorm3d :39-40: BatchNorm1d over the active rows (running buffers updated in place when training)."""
    return _on_active_rows(x, active, lambda nc: F.batch_norm(nc, running_mean, running_var, weight, bias, training, momentum, eps))


def sparse_group_norm(x, active, groups, weight, bias, eps=1e-6):
    """SparseGroupNorm :47-78: GroupNorm of the (N, C) matrix = per voxel over each channel group."""
    return _on_active_rows(x, active, lambda nc: F.group_norm(nc, groups, weight, bias, eps))


def sparse_layer_norm(x, active, weight, bias, eps=1e-6):
    """SparseConvNeXtLayerNorm :181-232 (x NCDHW here; the channels_last variant is the same numbers permuted)."""
    return _on_active_rows(x, active, lambda nc: F.layer_norm(nc, (nc.shape[1],), weight, bias, eps))


def sparse_grn(x_cl, active, gamma, beta):
    """SparseGRN sparse branch :116-127 on a channels-last (B,H,W,D,C) tensor: Gx = ||row||, Nx = Gx / (Gx.mean(-1) + 1e-6) over a size-1
    axis."""
    ii = up(active, x_cl.shape[1:4]).squeeze(1).nonzero(as_tuple=True)
    nc = x_cl[ii]
    Gx = torch.norm(nc, p=2, dim=1, keepdim=True)
    Nx = Gx / (Gx.mean(dim=-1, keepdim=True) + 1e-6)
    out = torch.zeros_like(x_cl)
    out[ii] = gamma * (nc * Nx) + beta if beta is not None else gamma * (nc * Nx)
    return out


def sparse_adaptive_avg_pool(x, active):
    """SparseAdaptiveAvgPooling :171-179."""
    m = up(active, x.shape[2:])
    return (x * m).sum(dim=(2, 3, 4), keepdim=True) / (m.sum(dim=(2, 3, 4), keepdim=True) + 1e-6)


def convnext_block(x, active, p, ks=7):
    """SparseConvNeXtBlock.forward :258-276 (drop_path = Identity); p: dict of the block's parameters by their state_dict names."""
    h = sparse_conv3d(x, active, p["dwconv.weight"], p["dwconv.bias"], groups=x.shape[1])
    h = sparse_layer_norm(h, active, p["norm.weight"], p["norm.bias"], 1e-6).permute(0, 2, 3, 4, 1)
    h = F.linear(h, p["pwconv1.weight"], p["pwconv1.bias"])
    h = F.gelu(h)
    h = F.linear(h, p["pwconv2.weight"], p["pwconv2.bias"])
    if p.get("gamma") is not None:
        h = p["gamma"] * h
    h = h.permute(0, 4, 1, 2, 3) * up(active, x.shape[2:])
    return x + h


def mednext_block(x, active, p, pre, stride=1, res_conv=False):
    """MedNeXtBlock.forward P/MedNeXt_head.py:301-309 / MedNeXtDownBlock.forward :343-351 under the converter (every Conv3d a SparseConv3d,
    the GroupNorm(num_groups = C) a SparseGroupNorm with eps 1e-5 -- nn.GroupNorm's default, which the converter passes on :318-320)."""
    C = x.shape[1]
    h = sparse_conv3d(x, active, p[pre + "conv1.weight"], p[pre + "conv1.bias"], stride=stride, groups=C)
    h = sparse_group_norm(h, active, C, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-5)
    h = F.gelu(sparse_conv3d(h, active, p[pre + "conv2.weight"], p[pre + "conv2.bias"]))
    h = sparse_conv3d(h, active, p[pre + "conv3.weight"], p[pre + "conv3.bias"])
    if stride == 1:
        return x + h                                        # do_res (:307-308)
    if res_conv:
        h = h + sparse_conv3d(x, active, p[pre + "res_conv.weight"], p[pre + "res_conv.bias"], stride=2)   # :347-349 (k1, padding 0 == k // 2)
    return h


def mednext_encoder(x, active, p):
    """MedNeXt.forward(x, hierarchical=True) P/MedNeXt_head.py:200-233 for block_counts = 1, do_res = do_res_up_down = True."""
    maps = []
    h = sparse_conv3d(x, active, p["stem.weight"], p["stem.bias"])
    for i in range(4):
        h = mednext_block(h, active, p, f"enc_block_{i}.0.")
        maps.append(h)
        h = mednext_block(h, active, p, f"down_{i}.", stride=2, res_conv=True)
    maps.append(mednext_block(h, active, p, "bottleneck.0."))
    return maps
