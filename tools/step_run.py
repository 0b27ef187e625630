"""N fused steps of STUNet-B 128^3 bf16 for a profiler to look at: python tools/step_run.py [batch] [steps] [side_stream 0/1]
(side stream off: every kernel runs alone on the chip, so rocprofv3's per-kernel durations are ISOLATED times)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
engine._USE_SIDE = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
tr.set_epoch(500)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for _ in range(3):
    tr.step(x, epoch=500)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    out = tr.step(x, epoch=500)
e1.record()
torch.cuda.synchronize()
print(f"B={B} side_stream={engine._USE_SIDE}: {e0.elapsed_time(e1) / steps:.2f} ms/step, loss {out['loss'].item():.5f}")
