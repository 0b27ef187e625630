"""Per-kernel time of the student sparse-encoder forward (the north-star figure), STUNet-B 128^3 bf16, mask 0.6 -- or, from the
environment, AM_ENC_SIZE=L AM_ENC_PATCH=160 AM_ENC_MASK=0.7 (BASELINE configs[3]) / H 192 0.6.   python tools/encoder_profile.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M, ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SIZE, PATCH, MR = os.environ.get("AM_ENC_SIZE", "B"), int(os.environ.get("AM_ENC_PATCH", "128")), float(os.environ.get("AM_ENC_MASK", "0.6"))
ALGO = {"S": 36.1e6, "B": 1105.3e6, "L": 6135e6, "H": 30622e6}[SIZE]      # algorithmic bytes per volume (bench.py ENC_FWD_ALGO)
kw = M.STUNET_CONFIGS[SIZE]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (PATCH,) * 3, MR, compute_dtype=torch.bfloat16).to(dev)
model._ensure_flat() if hasattr(model, "_ensure_flat") else None
x = torch.randn(B, PATCH, PATCH, PATCH, device=dev)
L = model.spec.fmap[0] * model.spec.fmap[1] * model.spec.fmap[2]
mk = ops.mask_sampler(torch.zeros(B, L, device=dev), torch.rand(B, L, device=dev), model.len_keep, 0)
mi = ops.MaskInfo(mk.view(B, *model.spec.fmap), n_active=B * model.len_keep)


def run():
    engine.forward(model.spec, model._W, model._pack, x, mi, True, None, encoder_only=True)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"encoder forward STUNet-{SIZE} {PATCH}^3 mask {MR} B={B}: {t:.3f} ms  -> {ALGO * B / t / 1e6:.0f} GB/s algorithmic ({ALGO * B / t / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
