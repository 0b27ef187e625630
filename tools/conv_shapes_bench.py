"""Forward / data-gradient rate of the step's convolution shapes (bf16, B from argv): python tools/conv_shapes_bench.py [B]
Run through tools/with_lib.py to compare builds on one box (tools/ab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [(64, 64, 128, 3, "fwd"), (64, 32, 128, 3, "fwd"), (32, 64, 128, 3, "dgrad"), (128, 128, 64, 3, "fwd"), (256, 256, 32, 3, "fwd"), (512, 512, 16, 3, "fwd"),
          (64, 64, 64, 4, "convT"), (128, 128, 32, 4, "convT")]


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
for ci, co, S, k, kind in shapes:
    if kind == "convT":
        x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
        w = torch.randn(ci, co, 4, 4, 4, device=dev) * 0.02
        wp = ops.pack_weight(w, torch.bfloat16, True, False)
        y = torch.empty(B, 2 * S, 2 * S, 2 * S, co, device=dev, dtype=torch.bfloat16)
        t = timed(lambda: ops.conv3d(ops.CONVT_FWD, x, wp, None, (2 * S,) * 3, 4, 2, out=y))
        fl = 2.0 * B * (2 * S) ** 3 * ci * co * 8
    else:
        x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
        # "dgrad ci->co": the data gradient of the conv co -> ci (its weight is (ci, co, 3, 3, 3)), input = dy with ci channels
        w = torch.randn(*((ci, co) if kind == "dgrad" else (co, ci)), 3, 3, 3, device=dev) * 0.02
        wp = ops.pack_weight(w, torch.bfloat16, False, kind == "dgrad")
        y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
        mode = ops.CONV_DGRAD if kind == "dgrad" else ops.CONV_FWD
        t = timed(lambda: ops.conv3d(mode, x, wp, None, (S, S, S), 3, 1, out=y))
        fl = 2.0 * B * S ** 3 * ci * co * 27
    out.append(f"{kind} {ci}->{co}@{S}: {t:.3f} ms {fl / t / 1e9:.0f} TF")
    del x, y
print(" | ".join(out), flush=True)
