"""Two ranks of the GENERIC trainer (anatomask_amd/generic_trainer.py, MedNeXt-shaped backbone) on ONE GPU, gloo over device tensors:
different init and different data per rank -> after 3 steps students and EMA teachers bit-identical across the ranks, and the
exchanged gradient of step 1 = the mean of the two ranks' local gradients.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/generic_two_ranks_one_gpu.py"""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.generic_trainer import GenericTrainer  # noqa: E402
from tests.helpers import tiny_mednext  # noqa: E402

rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.manual_seed(rank)                                   # DIFFERENT init per rank: the start-up broadcast must fix it
dense = tiny_mednext()
dense.get_downsample_ratio = lambda: 16
dense.get_feature_map_channels = lambda: [8, 16, 32, 64, 128]
enc = M.SparseEncoder(dense, input_size=(64, 64, 64))
dec = M.LightDecoder(enc.downsample_ratio, sbn=False, width=128, out_channel=1)
model = M.SparK(sparse_encoder=enc, dense_decoder=dec, mask_ratio=0.5, densify_norm="in", compute_dtype=torch.float32).train().to(dev)
with torch.no_grad():                                     # (trunc_normal(0.02) projections leave the loss flat: give the step something to do)
    for p in model.parameters():
        if p.dim() > 1:
            p.mul_(3.0)
tr = GenericTrainer(model, lr=1e-3, total_epochs=1000, seed=7)
GenericTrainer.BUCKET_BYTES = 256 << 10                   # several collectives per step on this small model
x = torch.randn(2, 1, 64, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
ref_model = copy.deepcopy(model)
for it in range(3):
    out = tr.step(x, epoch=500)
    if it == 0:
        assert len(tr.exchange_log) > 1 and max(tr.exchange_log) <= 256 << 10, tr.exchange_log
        xs = [torch.empty_like(x) for _ in range(2)]
        ms = [torch.empty_like(out["mask"]) for _ in range(2)]
        dist.all_gather(xs, x); dist.all_gather(ms, out["mask"].contiguous())
        ref_model.train()
        mean = None
        for xr, mr in zip(xs, ms):
            ref_model.zero_grad(set_to_none=True)
            inp, rec = ref_model(xr, active_b1ff=mr)
            loss, _ = ref_model.forward_loss(inp, rec, mr)
            loss.backward()
            g = {k: p.grad.detach().clone() for k, p in ref_model.named_parameters() if p.grad is not None}
            mean = g if mean is None else {k: 0.5 * (mean[k] + g[k]) for k in g}
        gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in mean.values())))
        worst = 0.0
        for (k, _), gv in zip(tr._params, tr._gviews):
            if k not in mean:
                continue
            gm = mean[k]
            worst = max(worst, float((gv * tr.grad_scale - gm).norm() / max(float(gm.norm()), 1e-3 * gn)))
        ok = worst < 2e-3 and abs(out["grad_norm"].item() - gn) < 1e-3 * gn
        print(f"rank {rank}: exchanged gradient vs mean of the local gradients: worst tensor rel {worst:.2e}; grad_norm {out['grad_norm'].item():.6f} "
              f"vs {gn:.6f}; mean-of-gradients ok: {ok}", flush=True)
        assert ok
torch.cuda.synchronize()
both = [torch.empty_like(tr.flat) for _ in range(2)]
dist.all_gather(both, tr.flat.detach().clone())
tboth = [torch.empty_like(tr.tflat) for _ in range(2)]
dist.all_gather(tboth, tr.tflat.detach().clone())
same, tsame = torch.equal(both[0], both[1]), torch.equal(tboth[0], tboth[1])
print(f"rank {rank}: loss {out['loss'].item():.5f} weights identical across ranks: {same}; teacher identical: {tsame}", flush=True)
assert same and tsame
dist.destroy_process_group()
