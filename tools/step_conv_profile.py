"""Per-call timing of every conv3d / conv3d_wgrad launch inside one AnatoMask step (STUNet-B 128^3, bf16, B=4):
wraps the ops with HIP-event timing (synchronous, so the sum is a little above the pipelined step)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for _ in range(2):
    tr.step(x, epoch=500)
rec = collections.OrderedDict()
MODE = {ops.CONV_FWD: "fwd", ops.CONV_DGRAD: "dgrad", ops.CONVT_FWD: "Tfwd", ops.CONVT_DGRAD: "Tdgrad"}
_c, _w = ops.conv3d, ops.conv3d_wgrad


def timed(fn, key, flops):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1)
    a = rec.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += flops
    return r


def frac(m):
    return float(m.t.float().mean()) if m is not None else 1.0


def conv3d(mode, x, wp, bias, out_spatial, ksize, stride, in_mask=None, in_bshift=0, out_mask=None, out_bshift=0, **k):
    Cout, Cin = wp.logical
    Do, Ho, Wo = out_spatial
    taps = ksize ** 3 if mode in (ops.CONV_FWD, ops.CONV_DGRAD) else 8
    nvox = x.shape[0] * Do * Ho * Wo
    if mode == ops.CONV_DGRAD and stride == 2:
        nvox = nvox // 8
    fl = 2.0 * nvox * Cin * Cout * taps * frac(out_mask or in_mask)
    key = f"{MODE[mode]:6s} k{ksize}s{stride} {Cin:3d}->{Cout:3d} out{Do:3d} {'sparse' if (in_mask or out_mask) else 'dense '}"
    return timed(lambda: _c(mode, x, wp, bias, out_spatial, ksize, stride, in_mask, in_bshift, out_mask, out_bshift, **k), key, fl)


def conv3d_wgrad(mode, x, dy, ksize, stride, x_mask=None, x_bshift=0, y_mask=None, y_bshift=0, **kwa):
    taps = ksize ** 3
    nv = dy.numel() // dy.shape[-1] if mode == ops.CONV_FWD else x.numel() // x.shape[-1]
    fl = 2.0 * nv * x.shape[-1] * dy.shape[-1] * (taps if mode == ops.CONV_FWD else 8) * frac(x_mask or y_mask)
    key = f"wgrad{'T' if mode != ops.CONV_FWD else ' '} k{ksize}s{stride} {x.shape[-1]:3d}x{dy.shape[-1]:3d} dy{dy.shape[1]:3d} {'sparse' if (x_mask or y_mask) else 'dense '}"
    return timed(lambda: _w(mode, x, dy, ksize, stride, x_mask, x_bshift, y_mask, y_bshift, **kwa), key, fl)


ops.conv3d, ops.conv3d_wgrad = conv3d, conv3d_wgrad
from anatomask_amd import engine  # noqa: E402
tr.step(x, epoch=500)
torch.cuda.synchronize()
tot = sum(v[1] for v in rec.values())
print(f"B={B}  total conv time {tot:.2f} ms")
for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print(f"{k}  n={v[0]:2d}  {v[1]:7.3f} ms  {v[2] / v[1] / 1e9 if v[1] else 0:7.0f} TF(active)")
