"""Step time of the launch-bound configurations: STUNet-S 48^3 B=2 (BASELINE configs[0]) and STUNet-B 128^3 B=1,
eager launches vs the captured hipGraph replay (AnatoMaskTrainer.graphed_step), with the host time per step (no synchronisation)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

dev = torch.device("cuda:0")


def run(model_key, size, B, graphed):
    kw = M.STUNET_CONFIGS[model_key]
    torch.manual_seed(0)
    model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (size,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
    tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
    x = torch.randn(B, 1, size, size, size, device=dev)
    step = (lambda: tr.graphed_step(x, epoch=500)) if graphed else (lambda: tr.step(x, epoch=500))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 30
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(n):
        step()
    host = (time.perf_counter() - t0) / n * 1e3          # until the last launch returned: includes blocking on a full launch queue
    cpu = (time.thread_time() - c0) / n * 1e3             # CPU time of the launching thread alone (what Python + ctypes + the runtime cost)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return wall, host, cpu


for key, size, B in [("S", 48, 2), ("B", 128, 1), ("B", 128, 4), ("B", 128, 8)]:
    e = run(key, size, B, False)
    line = f"STUNet-{key} {size}^3 B={B}: eager {e[0]:.2f} ms/step (last launch returned after {e[1]:.2f} ms, launching thread CPU time {e[2]:.2f} ms)"
    if hasattr(AnatoMaskTrainer, "graphed_step"):
        g = run(key, size, B, True)
        line += f" | hipGraph replay {g[0]:.2f} ms/step (returned after {g[1]:.2f} ms, CPU {g[2]:.2f} ms)"
    print(line, flush=True)
