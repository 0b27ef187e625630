"""Cost of the store-epilogue variants of conv_igemm on the decoder conv 64->64 @128^3 (B=16): plain, statistics (train-mode norm
follows), fused eval-mode BatchNorm + ReLU6 (the teacher's decoder), interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B, C, S = int(os.environ.get("AM_CB_BATCH", "16")), 64, 128
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
wp = ops.pack_weight(torch.randn(C, C, 3, 3, 3, device=dev) * 0.02, torch.bfloat16, False, False)
y = torch.empty_like(x)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev)
fl = 2.0 * B * S ** 3 * C * C * 27


def timed(fn, iters=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


V = {
    "plain": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y),
    "statistics": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, want_partials=True),
    "scale+shift+relu6": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, ep_scale=sc, ep_shift=sh, ep_act=ops.ACT_RELU6),
    "relu6 only": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, ep_act=ops.ACT_RELU6),
    "scale+shift+res": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, ep_scale=sc, ep_shift=sh, ep_res=x),
    "accumulate": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, accumulate=True),
    "scale+shift only": lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, ep_scale=sc, ep_shift=sh),
}
for rep in range(2):
    for k, fn in V.items():
        t = timed(fn)
        print(f"{k:20s} {t:.3f} ms {fl / t / 1e9:.0f} TFLOP/s", flush=True)
