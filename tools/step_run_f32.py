"""N fused steps of the reference RECIPE (STUNet-B, 112x112x128, batch 4, fp32 storage) for a profiler: python tools/step_run_f32.py [split 0/1] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

split = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (112, 112, 128), 0.6, compute_dtype=torch.float32).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1, distributed=False, f32_split=split)
tr.set_epoch(500)
x = torch.randn(4, 1, 112, 112, 128, device=dev)
for _ in range(2):
    tr.step(x, epoch=500)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    out = tr.step(x, epoch=500)
e1.record()
torch.cuda.synchronize()
print(f"fp32 storage split={split}: {e0.elapsed_time(e1) / steps:.2f} ms/step, loss {out['loss'].item():.5f}")
