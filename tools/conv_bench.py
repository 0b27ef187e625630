"""Micro-benchmark of the conv kernels on the STUNet-B/128^3 shapes (for rocprofv3 counter runs).
usage: python tools/conv_bench.py [fwd|wgrad|all] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
B, C, S = int(os.environ.get("AM_CB_BATCH", "2")), 64, 128      # AM_CB_BATCH=16: the launch bench.py times for `roofline` (its default batch)
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
dy = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.02
wp = ops.pack_weight(w, torch.bfloat16, False, False)
y = torch.empty_like(x)


def timed(fn, name, flops):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / iters
    print(f"{name}: {t:.3f} ms  {flops / t / 1e9:.1f} TFLOP/s")


fl = 2.0 * B * S ** 3 * C * C * 27
if what in ("fwd", "all"):
    # AM_CB_STATS=1: with the statistics epilogue (the launch the training step runs: the student's decoder conv feeds a BatchNorm)
    st = bool(int(os.environ.get("AM_CB_STATS", "0")))
    timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, want_partials=st), "conv fwd 64->64 @128^3" + (" + statistics" if st else ""), fl)
if what in ("wgrad", "all"):
    timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1), "conv wgrad 64x64 @128^3", fl)
