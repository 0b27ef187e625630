"""Ablation timing of the resident-weight conv kernel (conv_rw.hip) on the level-0 / level-1 encoder shapes, B=8, mask 0.6.
AM_CV_DBG bits (tools build only): 1 no stores, 2 no source loads, 4 no MFMA phase, 8 no LDS staging writes, 16 no staging-plan
arithmetic, 32 no epilogue; AM_CV_NORW=1: the generic kernel instead."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))      # tools-only library with the -DAM_ABLATE switches
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = 8
mk = ops.mask_sampler(torch.zeros(B, 512, device=dev), torch.rand(B, 512, device=dev), 205, 0)
mi = ops.MaskInfo(mk.view(B, 8, 8, 8), n_active=B * 205)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, cin, cout, si, so, stride, ibs, obs in (("level-0 conv2 32->32 s1 @128^3", 32, 32, 128, 128, 1, 4, 4),
                                                   ("level-1 conv1 32->64 s2 -> 64^3", 32, 64, 128, 64, 2, 4, 3)):
    x = torch.randn(B, si, si, si, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty(B, so, so, so, cout, device=dev, dtype=torch.bfloat16)
    out = []
    for dbg in (0, 7, 7 + 8, 7 + 16, 7 + 32, 7 + 8 + 16, 63):
        os.environ["AM_CV_DBG"] = str(dbg)
        t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (so,) * 3, 3, stride, in_mask=mi, in_bshift=ibs, out_mask=mi, out_bshift=obs, out=y))
        out.append(f"dbg{dbg}: {t:.0f}")
    os.environ["AM_CV_DBG"] = "0"
    os.environ["AM_CV_NORW"] = "1"
    t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (so,) * 3, 3, stride, in_mask=mi, in_bshift=ibs, out_mask=mi, out_bshift=obs, out=y))
    del os.environ["AM_CV_NORW"]
    print(name, " us | ".join(out), f"| generic kernel: {t:.0f}")
