"""Same-process A/B of the side stream's priority (weight gradients) against the main chain: python tools/side_priority.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
tr.set_epoch(500)
x = torch.randn(B, 1, 128, 128, 128, device=dev)


def timed(n=12):
    for _ in range(3):
        tr.step(x, epoch=500)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tr.step(x, epoch=500)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


least, greatest = torch.cuda.Stream.priority_range()
for rep in range(2):
    for name, mk in (("side default (0)", lambda: torch.cuda.Stream(device=dev)), (f"side least ({least})", lambda: torch.cuda.Stream(device=dev, priority=least)),
                     (f"side greatest ({greatest})", lambda: torch.cuda.Stream(device=dev, priority=greatest))):
        engine._SIDE.clear()
        engine._SIDE[0] = mk()
        print(f"{name:24s} {timed():8.2f} ms/step", flush=True)
    # main chain on a HIGH-priority stream, side default
    engine._SIDE.clear()
    hi = torch.cuda.Stream(device=dev, priority=greatest)
    with torch.cuda.stream(hi):
        print(f"{'main greatest, side 0':24s} {timed():8.2f} ms/step", flush=True)
    engine._SIDE.clear()
    engine._SIDE[0] = torch.cuda.Stream(device=dev, priority=least)
    with torch.cuda.stream(hi):
        print(f"{'main greatest, side least':24s} {timed():8.2f} ms/step", flush=True)
