"""Interleaved A/B of a knob on the full step in ONE process (robust to noisy neighbours on the GPU):
usage: python tools/step_ab.py KNOB=v1,v2,... [batch]
  KNOB = engine.<attr> / ops.<attr> flips a module attribute (e.g. engine._USE_SIDE=1,0, ops.FUSED_BWD_TAILS=1,0);
  any other KNOB is set as an environment variable (only the -DAM_ABLATE tools build reads any)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

knob, vals = sys.argv[1].split("=")
vals = vals.split(",")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for _ in range(3):
    tr.step(x, epoch=500)
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        if knob.startswith(("engine.", "ops.")):
            import importlib
            mod = importlib.import_module("anatomask_amd." + knob.split(".")[0])
            attr = knob.split(".", 1)[1]
            setattr(mod, attr, type(getattr(mod, attr))(int(v)))
        else:
            os.environ[knob] = v
        tr.step(x, epoch=500)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.step(x, epoch=500)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 5 * 1e3)
for v in vals:
    print(f"{knob}={v}: " + " ".join(f"{t:.1f}" for t in res[v]) + f"  ms/step (min {min(res[v]):.1f})")
