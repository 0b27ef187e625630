"""Census of the convolution launches of ONE training step (STUNet-B 128^3, B=8, bf16): every am_conv3d / am_conv3d_wgrad call with its
shape, HIP-event time (side stream off, so nothing overlaps) and TFLOP/s, grouped by shape -- where the MFMA time of the step goes and
which launches run far from the matrix-core peak.     python tools/conv_census.py [batch]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
SIZE, PATCH, MR = os.environ.get("AM_CENSUS_SIZE", "B"), int(os.environ.get("AM_CENSUS_PATCH", "128")), float(os.environ.get("AM_CENSUS_MASK", "0.6"))
ACT = 1.0 - MR                                   # active fraction: block-sparse launches are credited with the active voxels' flops
kw = M.STUNET_CONFIGS[SIZE]                      # AM_CENSUS_SIZE=L AM_CENSUS_PATCH=160 AM_CENSUS_MASK=0.7 python tools/conv_census.py 4
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (PATCH,) * 3, MR, compute_dtype=torch.bfloat16,
                      recompute=bool(int(os.environ.get("AM_CENSUS_RECOMPUTE", "0")))).to(dev)   # (recompute: the re-run forward launches are listed too)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, PATCH, PATCH, PATCH, device=dev)
engine._USE_SIDE = False
for _ in range(2):
    tr.step(x, epoch=500)
torch.cuda.synchronize()

rec = []
_conv, _wg = ops.conv3d, ops.conv3d_wgrad
MODES = {ops.CONV_FWD: "fwd", ops.CONV_DGRAD: "dgrad", ops.CONVT_FWD: "convT", ops.CONVT_DGRAD: "convT-dgrad"}


def conv3d(mode, x, w, bias, out_spatial, ksize, stride, in_mask=None, in_bshift=0, out_mask=None, out_bshift=0, **kwa):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _conv(mode, x, w, bias, out_spatial, ksize, stride, in_mask, in_bshift, out_mask, out_bshift, **kwa)
    e1.record()
    Bq, Di, Hi, Wi, Cin = x.shape
    Cout = w.logical[0]
    Do, Ho, Wo = out_spatial
    if mode in (ops.CONV_FWD, ops.CONVT_DGRAD):
        vox, taps = Bq * Do * Ho * Wo, (ksize ** 3 if mode == ops.CONV_FWD else 8 * 8)       # convT-dgrad = k4 s2 conv of dy: 64 taps per output voxel
    elif mode == ops.CONV_DGRAD:
        vox, taps = Bq * Do * Ho * Wo, ksize ** 3 / (stride ** 3)
    else:
        vox, taps = Bq * Do * Ho * Wo, 8
    sparse = (in_mask or out_mask) is not None
    rec.append((f"{MODES[mode]} k{ksize}s{stride} {Cin}->{Cout} @{Do}{'s' if sparse else ''}", 2.0 * vox * taps * Cin * Cout * (ACT if sparse else 1.0), e0, e1))
    return r


def conv3d_wgrad(mode, x, dy, ksize, stride, x_mask=None, x_bshift=0, y_mask=None, y_bshift=0, **kwa):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _wg(mode, x, dy, ksize, stride, x_mask, x_bshift, y_mask, y_bshift, **kwa)
    e1.record()
    Bq, Dx, Hx, Wx, Cx = x.shape
    _, Dy, Hy, Wy, Cy = dy.shape
    sparse = (x_mask or y_mask) is not None
    if mode == ops.CONV_FWD:
        fl = 2.0 * Bq * Dy * Hy * Wy * ksize ** 3 * Cx * Cy
    else:
        fl = 2.0 * Bq * Dy * Hy * Wy * 8 * Cx * Cy
    rec.append((f"wgrad {'convT' if mode != ops.CONV_FWD else ''} k{ksize}s{stride} {Cx}->{Cy} @{Dy}{'s' if sparse else ''}", fl * (ACT if sparse else 1.0), e0, e1))
    return r


ops.conv3d, ops.conv3d_wgrad = conv3d, conv3d_wgrad
engine.ops.conv3d, engine.ops.conv3d_wgrad = conv3d, conv3d_wgrad
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
tr.step(x, epoch=500)
e1.record()
torch.cuda.synchronize()
print(f"step (side stream off, instrumented): {e0.elapsed_time(e1):.1f} ms, {len(rec)} conv launches")
if os.environ.get("AM_CENSUS_EACH"):               # every launch of the shapes that contain this substring, in issue order
    for name, fl, a, b in rec:
        if os.environ["AM_CENSUS_EACH"] in name:
            print(f"   {name:44s} {a.elapsed_time(b):8.3f} ms {fl / a.elapsed_time(b) / 1e9:8.0f} TFLOP/s")
agg = OrderedDict()
for name, fl, a, b in rec:
    t = a.elapsed_time(b)
    n, T, F = agg.get(name, (0, 0.0, 0.0))
    agg[name] = (n + 1, T + t, F + fl)
tot = sum(v[1] for v in agg.values())
print(f"{'launch (s = block-sparse, FLOPs x' + format(ACT, '.1f') + ')':48s} {'n':>3s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>8s}")
for name, (n, T, F) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:48s} {n:3d} {T:8.3f} {100 * T / tot:6.1f} {F / T / 1e9:8.0f}")
print(f"{'total':48s} {len(rec):3d} {tot:8.3f}        {sum(v[2] for v in agg.values()) / tot / 1e9:8.0f}")
