"""One dense ConvTranspose3d k4 s2 launch (bf16) timed alone: python tools/ct_probe.py [C S B]   (tools build: AM_K3_DBG = 1 no stores | 2 no brick DMA
| 4 no weight DMA | 8 no MFMAs; AM_CV_NOK3T=1 = conv_igemm.hip)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

C, S, B = (int(v) for v in (sys.argv[1:4] + ["64", "64", "16"][len(sys.argv) - 1:]))
dev = "cuda:0"
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, 4, 4, 4, device=dev) * 0.02
wf = ops.pack_weight(w, torch.bfloat16, True, False)
bias = torch.zeros(C, device=dev)
y = torch.empty(B, 2 * S, 2 * S, 2 * S, C, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    ops.conv3d(ops.CONVT_FWD, x, wf, bias, (2 * S,) * 3, 4, 2, out=y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(15):
    ops.conv3d(ops.CONVT_FWD, x, wf, bias, (2 * S,) * 3, 4, 2, out=y)
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 15
print(f"ConvT {C}->{C} {S}^3->{2*S}^3 B={B} dbg={os.environ.get('AM_K3_DBG', '0')} nok3t={os.environ.get('AM_CV_NOK3T', '0')}: {t:.3f} ms {2.0 * B * (2 * S) ** 3 * C * C * 8 / t / 1e9:.0f} TF", flush=True)
