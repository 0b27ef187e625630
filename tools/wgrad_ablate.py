"""Ablation timing of conv_wgrad (AM_WG_DBG bits: 1 no atomic flush, 2 no contraction, 4 no global loads) on the decoder shapes, B=4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))      # tools-only library with the -DAM_ABLATE switches
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = 4


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for cx, cy, S in [(64, 64, 128), (128, 128, 64)]:
    x = torch.randn(B, S, S, S, cx, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, S, S, S, cy, device=dev).to(torch.bfloat16)
    fl = 2.0 * B * S ** 3 * cx * cy * 27
    out = []
    for rep in range(2):
        for dbg in [int(v) for v in os.environ.get('AM_ABLATE', '0,1,2,4,6').split(',')]:
            os.environ["AM_WG_DBG"] = str(dbg)
            t = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1))
            out.append(f"dbg{dbg}: {t:.3f} ms {fl / t / 1e9:.0f} TF")
    os.environ["AM_WG_DBG"] = "0"
    print(f"wgrad {cx}x{cy} @{S}^3: " + " | ".join(out), flush=True)
