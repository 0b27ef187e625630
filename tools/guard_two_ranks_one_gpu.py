"""Two ranks of the real trainer on ONE GPU (gloo over device tensors): a NaN volume on rank 1 ONLY at step 3 must latch the non-finite
guard (am_adamw_ema) on BOTH ranks at step 3 -- the NaN gradient survives the all-reduce -- and leave both ranks' weights, Adam moments,
teacher and BatchNorm buffers bit-identical to what they were after step 2 (P/pretrain_AntoMask.py:441-446: every step is checked).
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/guard_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.pretrain import first_nonfinite_step  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
kw = M.STUNET_CONFIGS["S"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=1000, seed=7 + rank)
x = torch.randn(2, 1, 48, 48, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))


def snap():
    t = tr.teacher.ema
    return [v.detach().clone() for v in (model._flat, model._bflat, model._iflat, tr.m, tr.v, t._flat, t._bflat, t._iflat)]


for it in range(1, 6):
    if it == 3:
        before = snap()
    xb = x.clone()
    if it == 3 and rank == 1:
        xb[1, 0, 20:24, 20:24, 20:24] = float("nan")
    tr.step(xb, epoch=500)
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(before, snap()))
bad = first_nonfinite_step(tr, 2)
print(f"rank {rank}: guard {tr.guard.tolist()} first non-finite step {bad}; state as before step 3: {same}", flush=True)
assert same and bad == 3 and tr.guard.tolist()[:3] == [1, 3, 5]
dist.destroy_process_group()
