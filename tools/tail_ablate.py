"""Ablation timing of the conv_igemm launches that run far below the dense rate (AM_CV_DBG bits: 1 no stores, 2 no source loads,
4 no weight loads, 16 no barriers in the group loop, 512 no statistics epilogue): transposed conv, 32-channel output, block-sparse
stride-2 forward / data gradient.     python tools/tail_ablate.py [B]"""
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
DBG = [int(v) for v in os.environ.get("AM_ABLATE", "0,1,2,4,3,7").split(",")]


def timed(fn, iters=12):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def sweep(name, fn):
    out = []
    for dbg in DBG:
        os.environ["AM_CV_DBG"] = str(dbg)
        out.append(f"dbg{dbg}: {timed(fn) * 1e3:.0f} us")
    os.environ["AM_CV_DBG"] = "0"
    print(f"{name}: " + " | ".join(out), flush=True)


# transposed conv 64->64 @64 -> 128 (dense)
x = torch.randn(B, 64, 64, 64, 64, device=dev).to(torch.bfloat16)
wp = ops.pack_weight(torch.randn(64, 64, 4, 4, 4, device=dev) * 0.02, torch.bfloat16, True, False)
y = torch.empty(B, 128, 128, 128, 64, device=dev, dtype=torch.bfloat16)
sweep("convT 64->64 @64->128", lambda: ops.conv3d(ops.CONVT_FWD, x, wp, None, (128,) * 3, 4, 2, out=y))
del x, y
# conv 64->32 @128 (dense)
x = torch.randn(B, 128, 128, 128, 64, device=dev).to(torch.bfloat16)
wp = ops.pack_weight(torch.randn(32, 64, 3, 3, 3, device=dev) * 0.02, torch.bfloat16, False, False)
y = torch.empty(B, 128, 128, 128, 32, device=dev, dtype=torch.bfloat16)
sweep("conv 64->32 @128", lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (128,) * 3, 3, 1, out=y))
del x, y
# block-sparse stride-2 forward / data gradient 32->64 @128->64, 64->128 @64->32
for ci, co, S, bs in [(32, 64, 128, 4), (64, 128, 64, 3)]:
    So = S // 2
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, So, So, So, co, device=dev).to(torch.bfloat16)
    y = torch.empty(B, So, So, So, co, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(B, S, S, S, ci, device=dev, dtype=torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wf, wb = ops.pack_weight(w, torch.bfloat16, False, False), ops.pack_weight(w, torch.bfloat16, False, True)
    sweep(f"sparse fwd k3s2 {ci}->{co} @{S}", lambda: ops.conv3d(ops.CONV_FWD, x, wf, None, (So,) * 3, 3, 2, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs - 1, out=y))
    sweep(f"sparse dgrad k3s2 {co}->{ci} @{S}", lambda: ops.conv3d(ops.CONV_DGRAD, dy, wb, None, (S,) * 3, 3, 2, in_mask=mi, in_bshift=bs - 1, out_mask=mi, out_bshift=bs, out=dx))
    del x, dy, y, dx
