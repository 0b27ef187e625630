"""Ablation timing of the block-sparse weight gradients (AM_WG_DBG bits: 1 no atomic flush, 2 no contraction, 4 no global loads,
8 no LDS staging writes) on the student encoder's shapes (STUNet-B 128^3, mask 0.6).   python tools/wgrad_sparse_ablate.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))      # tools-only library with the -DAM_ABLATE switches
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
keep = torch.rand(B, 8, 8, 8, device=dev).flatten(1).argsort(1)[:, :205]
mk = torch.zeros(B, 512, dtype=torch.uint8, device=dev).scatter_(1, keep, 1).view(B, 8, 8, 8)
mi = ops.MaskInfo(mk)


def up(m, f):
    return m.repeat_interleave(f, 1).repeat_interleave(f, 2).repeat_interleave(f, 3).unsqueeze(-1)


def timed(fn, iters=15):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for cx, cy, S, bs, stride in [(32, 64, 128, 4, 2), (64, 128, 64, 3, 2), (32, 32, 128, 4, 1), (64, 64, 64, 3, 1), (128, 128, 32, 2, 1)]:
    So = S // stride
    x = (torch.randn(B, S, S, S, cx, device=dev) * up(mk, S // 8)).to(torch.bfloat16)
    dy = (torch.randn(B, So, So, So, cy, device=dev) * up(mk, So // 8)).to(torch.bfloat16)
    out = []
    for dbg in [int(v) for v in os.environ.get("AM_ABLATE", "0,1,2,4,8,12,14,15").split(",")]:
        os.environ["AM_WG_DBG"] = str(dbg)
        t = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, stride, x_mask=mi, x_bshift=bs, y_mask=mi, y_bshift=bs - (1 if stride == 2 else 0)))
        out.append(f"dbg{dbg}: {t * 1e3:.0f} us")
    os.environ["AM_WG_DBG"] = "0"
    print(f"wgrad k3 s{stride} {cx}->{cy} @{S} sparse: " + " | ".join(out), flush=True)
