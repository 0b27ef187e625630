"""Per-call timing of the norm / streaming ops inside one AnatoMask step (STUNet-B 128^3, bf16, B=4)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for _ in range(2):
    tr.step(x, epoch=500)
rec = []


def wrap(name, fn, shape_of):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record(); e1.synchronize()
        t = shape_of(*a, **k)
        rec.append((name, tuple(t.shape), e0.elapsed_time(e1) * 1e3, k.get("mask") is not None or (len(a) > 3 and isinstance(a[3], ops.MaskInfo))))
        return r
    return w


ops.norm_apply = wrap("norm_apply", ops.norm_apply, lambda *a, **k: a[0])
_nb = ops.norm_backward
if os.environ.get("NO_DXSUM"):
    def _nb2(*a, **k):
        k.pop("dxsum", None)
        return _nb(*a, **k)
    ops.norm_backward = wrap("norm_backward", _nb2, lambda *a, **k: a[2])
else:
    ops.norm_backward = wrap("norm_backward", _nb, lambda *a, **k: a[2])
ops.chan_stats = wrap("chan_stats", ops.chan_stats, lambda *a, **k: a[0])
tr.step(x, epoch=500)
torch.cuda.synchronize()
tot = collections.defaultdict(float)
for n, s, t, m in rec:
    nb = 1
    for d in s:
        nb *= d
    print(f"{n:14s} {str(s):28s} {t:8.1f} us   {nb * 2 / 1e6:8.1f} MB/tensor")
    tot[n] += t
print({k: round(v / 1e3, 2) for k, v in tot.items()})
