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
import copy  # noqa: E402
ref_model = copy.deepcopy(model)                          # the weights BOTH ranks start from (after the trainer's start-up broadcast)
for it in range(3):
    out = tr.step(x, epoch=500)
    if it == 0:
        # ---- the exchanged gradient must be the MEAN of the two ranks' local gradients: recompute both local gradients on this rank
        # through the module API (loss.backward() on the start weights, with each rank's own input and sampled mask) and compare
        # with what the trainer's flat buffer holds after the all-reduce (SUM) times the factor the optimizer folds in (1 / world)
        xs = [torch.empty_like(x) for _ in range(2)]
        ms = [torch.empty_like(out["mask"]) for _ in range(2)]
        dist.all_gather(xs, x); dist.all_gather(ms, out["mask"].contiguous())
        ref_model.train()
        mean = None
        for xr, mr in zip(xs, ms):
            ref_model.zero_grad(set_to_none=True)
            inp, rec = ref_model(xr, active_b1ff=mr.view(2, 1, *model.spec.fmap).bool())
            loss, _ = ref_model.forward_loss(inp, rec, mr.view(2, 1, *model.spec.fmap).bool())
            loss.backward()
            g = {k: p.grad.detach().clone() for k, p in ref_model.named_parameters() if p.grad is not None}
            mean = g if mean is None else {k: 0.5 * (mean[k] + g[k]) for k in g}
        gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in mean.values())))
        worst = 0.0
        for k, gm in mean.items():                           # (tensors with an analytically zero gradient -- conv biases under a norm -- hold
            o, n = model._offs[k], gm.numel()                #  rounding noise only: the error is measured against max(|g|, 1e-3 |all g|))
            got = model._gflat[o:o + n].view_as(gm) * tr.grad_scale
            worst = max(worst, float((got - gm).norm() / max(float(gm.norm()), 1e-3 * gn)))
        print(f"rank {rank}: exchanged gradient vs mean of the local gradients: worst tensor rel {worst:.2e}; "
              f"grad_norm {out['grad_norm'].item():.6f} vs {gn:.6f}; mean-of-gradients ok: {worst < 2e-3 and abs(out['grad_norm'].item() - gn) < 1e-3 * gn}", flush=True)
        assert worst < 2e-3 and abs(out["grad_norm"].item() - gn) < 1e-3 * gn
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
