"""Two ranks of the real trainer on ONE GPU (gloo over device tensors -- RCCL refuses two ranks on one device): checks that the
overlapped per-group gradient exchange keeps the ranks' weights bit-identical and equal to the average-of-gradients step.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ddp_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
kw = M.STUNET_CONFIGS["S"]
torch.manual_seed(rank)                                   # DIFFERENT init per rank: the start-up broadcast must fix it
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (64, 64, 64), 0.6, compute_dtype=torch.float32).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=1000, seed=7)          # same sampler seed: same masks on both ranks
x = torch.randn(2, 1, 64, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
for it in range(3):
    out = tr.step(x, epoch=500)
torch.cuda.synchronize()
flat = model._flat.detach().clone()
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
same = torch.equal(other[0], other[1])
ema_same_l = [torch.empty_like(tr.teacher.ema._flat) for _ in range(2)]
dist.all_gather(ema_same_l, tr.teacher.ema._flat.detach().clone())
print(f"rank {rank}: loss {out['loss'].item():.5f} grad_norm {out['grad_norm'].item():.5f} weights identical across ranks: {same}; "
      f"teacher identical: {torch.equal(ema_same_l[0], ema_same_l[1])}", flush=True)
assert same and torch.equal(ema_same_l[0], ema_same_l[1])
dist.destroy_process_group()
