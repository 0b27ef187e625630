"""Ablation timing (AM_CV_DBG) of the level-0 / level-1 block-sparse encoder convs, B=4, mask 0.6."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))      # tools-only library with the -DAM_ABLATE switches
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
    return e0.elapsed_time(e1) / iters * 1e3


for ci, co, S, bs in [(32, 32, 128, 4), (64, 64, 64, 3)]:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
    out = []
    for rep in range(2):
        for dbg in [int(v) for v in os.environ.get("AM_ABLATE", "0,1,2,4,16,32,55").split(",")]:
            os.environ["AM_CV_DBG"] = str(dbg)
            t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y))
            out.append(f"dbg{dbg}: {t:.0f}")
        os.environ["AM_CV_DBG"] = "0"
        t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y, want_partials=True))
        out.append(f"+partials: {t:.0f}")
        t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out_mask=mi, out_bshift=bs, out=y))
        out.append(f"no in_mask: {t:.0f}")
    print(f"sparse conv {ci}->{co} @{S}^3 (us): " + " | ".join(out), flush=True)
