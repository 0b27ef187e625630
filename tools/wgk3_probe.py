"""One dense k3 s1 weight-gradient launch (bf16) timed alone: python tools/wgk3_probe.py [Cx Cy S B iters]
(tools build + AM_WGK3_DBG = 1 no flush | 2 no MFMAs | 4 no DMA for timing ablations; AM_WG_NOK3=1 = conv_wgrad.hip)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

cx, cy, S, B, it = (int(v) for v in (sys.argv[1:6] + ["64", "64", "128", "16", "15"][len(sys.argv) - 1:]))
dev = "cuda:0"
x = torch.randn(B, S, S, S, cx, device=dev).to(torch.bfloat16)
dy = torch.randn(B, S, S, S, cy, device=dev).to(torch.bfloat16)
for _ in range(4):
    ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it):
    ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1)
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / it
print(f"wgrad k3 {cx}->{cy}@{S} B={B} dbg={os.environ.get('AM_WGK3_DBG', '0')} nok3={os.environ.get('AM_WG_NOK3', '0')}: {t:.3f} ms {2.0 * B * S ** 3 * cx * cy * 27 / t / 1e9:.0f} TF", flush=True)
