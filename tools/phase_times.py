"""HIP-event times of the phases of one fused step (STUNet-B 128^3 bf16): teacher forward, student forward, backward, optimizer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for _ in range(3):
    tr.step(x, epoch=500)
marks = []
_f, _b = engine.forward, engine.backward


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def fwd(*a, **k):
    e0 = ev(); r = _f(*a, **k); marks.append(("teacher fwd" if not k.get("train", a[5] if len(a) > 5 else False) else "student fwd", e0, ev())); return r


def bwd(*a, **k):
    e0 = ev(); r = _b(*a, **k); marks.append(("backward", e0, ev())); return r


engine.forward, engine.backward = fwd, bwd
tot = {}
for _ in range(5):
    marks.clear()
    s0 = ev(); tr.step(x, epoch=500); s1 = ev()
    torch.cuda.synchronize()
    for n, a, b in marks:
        tot.setdefault(n, []).append(a.elapsed_time(b))
    tot.setdefault("step", []).append(s0.elapsed_time(s1))
for n, v in tot.items():
    print(f"{n:12s} {min(v):7.2f} ms (min of {len(v)})")
print(f"rest (sampler, loss, optimizer, EMA, repack) {min(tot['step']) - sum(min(v) for k, v in tot.items() if k != 'step'):.2f} ms")
