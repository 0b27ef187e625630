"""What do the 28 small unpack_grad launches behind the side stream's weight gradients cost the STEP?  (Isolated they take 14 us each; inside
the step rocprofv3 shows 277 us each: they wait for a free wave slot beside the main stream's persistent kernels.)  Timing only: the variant
without them computes wrong gradients.   python tools/unpack_cost.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
tr.set_epoch(500)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
real_unpack, real_zeros = ops.unpack_grad, torch.zeros


def timed(n=8):
    for _ in range(2):
        tr.step(x, epoch=500)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tr.step(x, epoch=500)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(2):
    ops.unpack_grad = real_unpack
    t_real = timed()
    ops.unpack_grad = lambda *a, **k: None
    t_skip = timed()
    print(f"B={B} rep {rep}: with unpack_grad {t_real:.2f} ms/step, without {t_skip:.2f}", flush=True)
ops.unpack_grad = real_unpack
