"""Timing of the multi-unit conv modes (ConvTranspose fwd / dgrad, stride-2 conv fwd / dgrad) on the STUNet-B shapes, B=4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(os.environ.get("AM_CB_BATCH", "4"))


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for C, S in [(64, 64), (128, 32), (256, 16)]:
    x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, 2 * S, 2 * S, 2 * S, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, C, 4, 4, 4, device=dev) * 0.02
    wf, wd = ops.pack_weight(w, torch.bfloat16, True, False), ops.pack_weight(w, torch.bfloat16, True, True)
    fl = 2.0 * B * (2 * S) ** 3 * C * C * 8
    t1 = timed(lambda: ops.conv3d(ops.CONVT_FWD, x, wf, None, (2 * S,) * 3, 4, 2))
    t2 = timed(lambda: ops.conv3d(ops.CONVT_DGRAD, dy, wd, None, (S,) * 3, 4, 2))
    print(f"ConvT {C}->{C} {S}^3->{2*S}^3: fwd {t1:.3f} ms ({fl/t1/1e9:.0f} TF)  dgrad {t2:.3f} ms ({fl/t2/1e9:.0f} TF)", flush=True)
for ci, co, S in [(32, 64, 128), (64, 128, 64), (128, 256, 32)]:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, S // 2, S // 2, S // 2, co, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wf, wd = ops.pack_weight(w, torch.bfloat16, False, False), ops.pack_weight(w, torch.bfloat16, False, True)
    fl = 2.0 * B * (S // 2) ** 3 * ci * co * 27
    t1 = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wf, None, (S // 2,) * 3, 3, 2))
    t2 = timed(lambda: ops.conv3d(ops.CONV_DGRAD, dy, wd, None, (S,) * 3, 3, 2))
    print(f"conv k3 s2 {ci}->{co} {S}^3->{S//2}^3 (dense): fwd {t1:.3f} ms ({fl/t1/1e9:.0f} TF)  dgrad {t2:.3f} ms ({fl/t2/1e9:.0f} TF)", flush=True)
