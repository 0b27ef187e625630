"""SyncBatchNorm decoder (P/decoder3D.py:42-43, on in P/pretrain_DDP.py:225) with two ranks on ONE GPU (gloo over device tensors):
each rank forwards + backwards its own 2 volumes with LightDecoder(sbn=True); rank 0 also runs the SAME 4 volumes as one batch
through the plain BatchNorm model.  Statistics over the global batch => identical reconstructions, BN buffers, input gradients, and
(after DDP's averaging: mean of the per-rank LOCAL sums) parameter gradients = the big-batch gradients.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 tools/syncbn_two_ranks.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402

rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dims, width, size = [8, 16, 32, 64, 128, 128], 128, (32, 32, 48)
torch.manual_seed(0)
sync = M.build_spark(dims, [1] * 6, width, size, 0.6, sbn=True).to(dev).train()
torch.manual_seed(0)
plain = M.build_spark(dims, [1] * 6, width, size, 0.6, sbn=False).to(dev).train()
g = torch.Generator().manual_seed(5)
chs = [width // 2 ** i for i in range(4)]
sizes = [tuple(v // 16 * 2 ** i for v in size) for i in range(4)]
tall = [torch.randn(4, chs[i], *sizes[i], generator=g) for i in range(4)]     # (the pooled sparse InstanceNorm of the encoder is per LOCAL
# batch in the reference, so a 2 + 2 split of the full model is not a 4-batch: the decoder is compared on given inputs)
tin = [t[2 * rank:2 * rank + 2].to(dev).requires_grad_(True) for t in tall]
rec = sync.dense_decoder(tin)
loss = (rec ** 2).sum() / (4 * rec[0].numel())                  # each rank's share of the big-batch mean
loss.backward()
dec_names = [n for n, _ in sync.named_parameters() if n.startswith("dense_decoder")]
gr = torch.cat([dict(sync.named_parameters())[n].grad.flatten() for n in dec_names])
dist.all_reduce(gr)                                             # DDP: sum of the per-rank gradients of (sum of per-rank shares)
bufs = torch.cat([b.flatten().float() for n, b in sync.named_buffers()])
ok = True
if rank == 0:
    tb = [t.to(dev).requires_grad_(True) for t in tall]
    recb = plain.dense_decoder(tb)
    lb = (recb ** 2).mean()
    lb.backward()
    gb = torch.cat([dict(plain.named_parameters())[n].grad.flatten() for n in dec_names])
    bb = torch.cat([b.flatten().float() for n, b in plain.named_buffers()])
    e_rec = ((rec - recb[:2]).abs().max() / recb.abs().max()).item()
    e_g = ((gr - gb).norm() / gb.norm()).item()
    e_b = ((bufs - bb).abs().max() / bb.abs().max()).item()
    e_in = max(((tin[i].grad - tb[i].grad[:2]).norm() / tb[i].grad[:2].norm()).item() for i in range(4))
    print(f"syncbn vs big batch: rec {e_rec:.2e} decoder-gradients {e_g:.2e} input-gradients {e_in:.2e} buffers {e_b:.2e}", flush=True)
    ok = e_rec < 2e-4 and e_g < 2e-3 and e_in < 2e-3 and e_b < 1e-5
flag = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
print(f"rank {rank}: syncbn ok: {bool(flag.item())}", flush=True)
assert flag.item() > 0
dist.destroy_process_group()
