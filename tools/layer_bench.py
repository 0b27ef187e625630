"""Throughput of the sparse layer zoo kernels (csrc/layer_ops.hip) on a block-sparse 128^3 x 32-channel bf16 volume (B=4, mask 0.6, 16^3
patches -- the stage-0 tensor of a ConvNeXt/MedNeXt-style encoder under SparK) and on its 64^3 x 64 stage: HIP-event time, algorithmic
HBM bytes (active voxels, each tensor once) / time vs the 8 TB/s peak, and VALU TFLOP/s for the depthwise convolutions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M, ops, sparse_layers as SL  # noqa: E402
from anatomask_amd.hip import lib  # noqa: E402

dev = "cuda:0"
dt = torch.bfloat16


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def line(name, ms, nbytes, flops=None):
    s = f"{name:34s} {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.0f} GB/s  {nbytes / ms / 1e6 / 8000:6.1%} of HBM peak"
    if flops:
        s += f"  {flops / ms / 1e9:7.1f} TFLOP/s (VALU fp32 peak 157)"
    print(s, flush=True)


for (S, C, B) in [(128, 32, 4), (64, 64, 4)]:
    f = 8
    g = torch.Generator().manual_seed(0)
    act = torch.zeros(B, f ** 3, dtype=torch.bool)
    for b in range(B):
        act[b, torch.randperm(f ** 3, generator=g)[:round(f ** 3 * 0.4)]] = True
    M._cur_active = act.view(B, 1, f, f, f).to(dev)
    mi = SL.current_mask(dev)
    bs = SL._bshift(mi, S)
    mp, fd, fh, fw = ops._mk(mi)
    al = ops._al(mi)
    nact = al[1] * (1 << (3 * bs))
    tb = nact * C * 2                                          # bytes of one pass over the active voxels of a tensor
    x = torch.randn(B, S, S, S, C, device=dev).to(dt)
    dy = torch.randn(B, S, S, S, C, device=dev).to(dt)
    y = torch.empty_like(x)
    ga, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    dg, db = torch.zeros(64, C, device=dev), torch.zeros(64, C, device=dev)
    L, st = lib(), ops._stream()
    print(f"--- {S}^3 x {C} ch, B={B}, {nact} active voxels ({tb / 1e6:.0f} MB per tensor pass)")
    line("LayerNorm fwd", timed(lambda: L.voxel_norm_fwd(1, 0, x.data_ptr(), y.data_ptr(), B, S, S, S, C, 1, ga.data_ptr(), be.data_ptr(), 1e-6, mp, bs, *al, st)), 2 * tb)
    line("LayerNorm bwd", timed(lambda: L.voxel_norm_bwd(1, 0, x.data_ptr(), dy.data_ptr(), y.data_ptr(), B, S, S, S, C, 1, ga.data_ptr(), 1e-6, dg.data_ptr(), db.data_ptr(), mp, bs, *al, st)), 3 * tb)
    line("GRN fwd", timed(lambda: L.voxel_norm_fwd(1, 1, x.data_ptr(), y.data_ptr(), B, S, S, S, C, 1, ga.data_ptr(), be.data_ptr(), 0.0, mp, bs, *al, st)), 2 * tb)
    line("GELU fwd", timed(lambda: L.gelu(1, x.data_ptr(), None, y.data_ptr(), B, S, S, S, C, mp, bs, *al, st)), 2 * tb)
    line("GELU bwd", timed(lambda: L.gelu(1, x.data_ptr(), dy.data_ptr(), y.data_ptr(), B, S, S, S, C, mp, bs, *al, st)), 3 * tb)
    line("layer scale + residual fwd", timed(lambda: L.scale_residual(1, 0, x.data_ptr(), dy.data_ptr(), ga.data_ptr(), y.data_ptr(), None, B, S, S, S, C, mp, bs, *al, st)), 3 * tb)
    yo = torch.empty(B, S // 2, S // 2, S // 2, C, device=dev, dtype=dt)
    idx = torch.empty(B, S // 2, S // 2, S // 2, C, device=dev, dtype=torch.int32)
    line("max pool k2 s2 fwd (+argmax)", timed(lambda: L.pool3d_fwd(1, 0, x.data_ptr(), yo.data_ptr(), idx.data_ptr(), B, S, S, S, C, 2, 2, 0, 1, 1, S // 2, S // 2, S // 2, mp, bs, bs - 1, fd, fh, fw, *al, st)), tb + tb // 8 * 3)
    line("max pool k2 s2 bwd", timed(lambda: L.pool3d_bwd(1, 0, yo.data_ptr(), idx.data_ptr(), y.data_ptr(), B, S, S, S, C, 2, 2, 0, 1, 1, S // 2, S // 2, S // 2, mp, bs, bs - 1, fd, fh, fw, *al, st)), tb + tb // 8 * 3)
    for k in (3, 7):
        w = torch.randn(C, k ** 3, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        dw, dbb = torch.zeros(C, k ** 3, device=dev), torch.zeros(C, device=dev)
        fl = 2.0 * nact * C * k ** 3
        line(f"depthwise {k}^3 fwd", timed(lambda: L.dwconv3d(1, 0, x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, S, S, S, C, k, mp, bs, fd, fh, fw, st)), 2 * tb, fl)
        line(f"depthwise {k}^3 wgrad", timed(lambda: L.dwconv3d_wgrad(1, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), dbb.data_ptr(), B, S, S, S, C, k, mp, bs, fd, fh, fw, st)), 2 * tb, fl)
