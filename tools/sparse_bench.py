"""Sparse vs dense timing of the encoder conv shapes (STUNet-B, 128^3, mask 0.6, B=4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = 4
torch.manual_seed(0)
keep = torch.rand(B, 8, 8, 8, device=dev).flatten(1).argsort(1)[:, :205]
mk = torch.zeros(B, 512, dtype=torch.uint8, device=dev).scatter_(1, keep, 1).view(B, 8, 8, 8)
mi = ops.MaskInfo(mk)


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for ci, co, S, bs in [(32, 32, 128, 4), (64, 64, 64, 3), (128, 128, 32, 2), (256, 256, 16, 1), (512, 512, 8, 0)]:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, S, S, S, co, device=dev).to(torch.bfloat16)
    fl = 2.0 * B * S ** 3 * ci * co * 27
    td = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y))
    ts = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y))
    tsp = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y, want_partials=True))
    wd = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1))
    ws = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1, x_mask=mi, x_bshift=bs, y_mask=mi, y_bshift=bs))
    print(f"{ci}->{co} @{S}^3: fwd dense {td:.3f} ms ({fl/td/1e9:.0f} TF) sparse {ts:.3f} ms (x{td/ts:.2f}; ideal x2.5) +partials {tsp:.3f} | "
          f"wgrad dense {wd:.3f} ms ({fl/wd/1e9:.0f} TF) sparse {ws:.3f} ms (x{wd/ws:.2f})", flush=True)
