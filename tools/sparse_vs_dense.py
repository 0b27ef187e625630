"""The level-0 64->32 @128^3 convolution dense against block-sparse (mask 0.6): how far the sparse launch is from 40 % of the dense one,
and what the input-mask lookups cost (out_mask only = inactive bricks skipped, halo rows read unmasked).  python tools/sparse_vs_dense.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
keep = torch.rand(B, 8, 8, 8, device=dev).flatten(1).argsort(1)[:, :205]
mk = torch.zeros(B, 512, dtype=torch.uint8, device=dev).scatter_(1, keep, 1).view(B, 8, 8, 8)
mi = ops.MaskInfo(mk)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for ci, co, S, bs in [(64, 32, 128, 4), (64, 64, 128, 4), (64, 64, 64, 3)]:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    wp = ops.pack_weight(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02, torch.bfloat16, False, False)
    y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * B * S ** 3 * ci * co * 27
    td = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y))
    ts = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y))
    to = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out_mask=mi, out_bshift=bs, out=y))
    print(f"{ci}->{co} @{S}^3 B={B}: dense {td:.3f} ms ({fl / td / 1e9:.0f} TF) | sparse {ts:.3f} ms = {ts / td:.2f} x dense ({0.4 * fl / ts / 1e9:.0f} TF on the active 40 %)"
          f" | out_mask only {to:.3f} ms = {to / td:.2f} x", flush=True)
