"""Two ranks of the real trainer on ONE GPU (gloo over device tensors), bit-reproducible mode, the SAME data and sampler seed on both
ranks: the all-reduced SUM is exactly 2 g and the folded 1/world makes it g again, so after N steps every rank must hold exactly the
bits of a single-process run (the pytest process computes those).  Prints the SHA-256 of the weights, the teacher and AdamW's moments.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 tools/ddp_same_data_det.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402


def run(distributed: bool):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(dev)
    tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=5, deterministic_wgrad=True, distributed=distributed)
    g = torch.Generator(device=dev).manual_seed(123)
    for _ in range(3):
        tr.step(torch.randn(2, 1, 48, 48, 48, device=dev, generator=g), epoch=50)
    torch.cuda.synchronize()
    n = model._live_end
    h = hashlib.sha256()
    for t in (model._flat, tr.teacher.ema._flat, tr.m[:n], tr.v[:n], model._bflat):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    print(f"rank {rank} sha256 {run(True)}", flush=True)
    dist.destroy_process_group()
