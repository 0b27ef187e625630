"""Isolated time of the stem kernels (Cin = 1, C = 32, 128^3, 40 % active, bf16): forward (matrix cores) and the two weight gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
f, C, S = 8, 32, 128
g = torch.Generator().manual_seed(0)
act = torch.zeros(B, f ** 3, dtype=torch.bool)
for b in range(B):
    act[b, torch.randperm(f ** 3, generator=g)[:205]] = True
mi = ops.MaskInfo(act.view(B, f, f, f).to(dev).to(torch.uint8).contiguous(), n_active=B * 205)
x = torch.randn(B, S, S, S, device=dev)
dy = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, 1, 3, 3, 3, device=dev) * 0.1
bias = torch.zeros(C, device=dev)
dw3, dw1, db = torch.zeros(C, 27, device=dev), torch.zeros(C, 1, device=dev), torch.zeros(C, device=dev)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


nact = B * 205 * 4096
for name, fn, nbytes in [("stem conv k3 fwd", lambda: ops.stem_conv_fwd(x, w, bias, mi, 4, torch.bfloat16), nact * (4 + C * 2)),
                         ("stem wgrad k3", lambda: ops.stem_conv_wgrad(x, dy, 3, mi, 4, dw3, db), nact * (4 + C * 2)),
                         ("stem wgrad k1", lambda: ops.stem_conv_wgrad(x, dy, 1, mi, 4, dw1, None), nact * (4 + C * 2))]:
    t = timed(fn)
    print(f"{name:18s} {t * 1e3:8.1f} us  {nbytes / t / 1e6:7.0f} GB/s  {nbytes / t / 1e6 / 8000:6.1%} of HBM peak", flush=True)
