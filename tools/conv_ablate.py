"""Ablation timing of conv_igemm (AM_CV_DBG bits: 1 no stores, 2 no source loads, 4 no weight loads) on the step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))      # tools-only library with the -DAM_ABLATE switches
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = 4
shapes = [(64, 64, 128), (64, 32, 128), (128, 128, 64), (256, 256, 32), (512, 512, 16)]


def timed(fn, iters=10):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for ci, co, S in shapes:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * B * S ** 3 * ci * co * 27
    out = []
    DBG = [int(v) for v in os.environ.get("AM_ABLATE", "0").split(",")]
    for rep in range(2):
        for dbg in DBG:
            os.environ["AM_CV_DBG"] = str(dbg)
            t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y), iters=20)
            out.append(f"dbg{dbg}: {t:.3f} ms {fl / t / 1e9:.0f} TF")
    os.environ["AM_CV_DBG"] = "0"
    print(f"conv {ci}->{co} @{S}^3 B={B}: " + " | ".join(out), flush=True)
