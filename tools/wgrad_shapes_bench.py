"""Weight-gradient rate of the step's convolution shapes (bf16, B from argv): python tools/wgrad_shapes_bench.py [B]
Run through tools/with_lib.py to compare builds on one box (tools/ab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (Cx, Cy, S of x, kind)
shapes = [(64, 64, 128, "k3"), (64, 32, 128, "k3"), (32, 64, 128, "k3"), (128, 128, 64, "k3"), (256, 256, 32, "k3"), (512, 512, 16, "k3"),
          (64, 128, 128, "k3s2"), (128, 256, 64, "k3s2"), (128, 64, 64, "convT"), (256, 128, 32, "convT"), (512, 256, 16, "convT")]


def timed(fn, iters=15):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


out = []
for cx, cy, S, kind in shapes:
    x = torch.randn(B, S, S, S, cx, device=dev).to(torch.bfloat16)
    if kind == "k3":
        dy = torch.randn(B, S, S, S, cy, device=dev).to(torch.bfloat16)
        t = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1))
        fl = 2.0 * B * S ** 3 * cx * cy * 27
    elif kind == "k3s2":
        dy = torch.randn(B, S // 2, S // 2, S // 2, cy, device=dev).to(torch.bfloat16)
        t = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 2))
        fl = 2.0 * B * (S // 2) ** 3 * cx * cy * 27
    else:
        dy = torch.randn(B, 2 * S, 2 * S, 2 * S, cy, device=dev).to(torch.bfloat16)
        t = timed(lambda: ops.conv3d_wgrad(ops.CONVT_FWD, x, dy, 4, 2))
        fl = 2.0 * B * (2 * S) ** 3 * cx * cy * 8
    out.append(f"{kind} {cx}->{cy}@{S}: {t:.3f} ms {fl / t / 1e9:.0f} TF")
    del x, dy
print(" | ".join(out), flush=True)
