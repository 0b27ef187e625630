"""How reproducible is the fused step run to run?  Two trainers from the same seed take the same 3 steps; reports how many weights
differ bitwise, with the default fp32-atomic weight-gradient reduction and with deterministic_wgrad=True."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M, ops  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS[os.environ.get("AM_DP_SIZE", "B")]


def run(det, dtype):
    ops.DETERMINISTIC_WGRAD = False
    torch.manual_seed(0)
    model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (64, 64, 64), 0.6, compute_dtype=dtype).to(dev)
    tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=100, seed=3, deterministic_wgrad=det)
    x = torch.randn(4, 1, 64, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    g1 = None
    for i in range(3):
        o = tr.step(x, epoch=50)
        if i == 0:
            g1 = model._gflat[:model._live_end].clone()
    return model._flat[:model._live_end].clone(), g1, o["loss"].item(), model


for dtype in (torch.float32, torch.bfloat16):
    for det in (False, True):
        (w1, g1, l1, m), (w2, g2, l2, _) = run(det, dtype), run(det, dtype)
        nd_w, nd_g = int((w1 != w2).sum()), int((g1 != g2).sum())
        # which tensors' first-step gradients differ
        bad = [k for k in m._pnames if k not in m._dead and not torch.equal(g1[m._offs[k]:m._offs[k] + m._W[k].numel()], g2[m._offs[k]:m._offs[k] + m._W[k].numel()])]
        print(f"{str(dtype)[6:]:9s} deterministic_wgrad={det}: first-step gradient elements differing {nd_g} / {g1.numel()} in {len(bad)} tensors "
              f"{[b.replace('sparse_encoder.sp_cnn.conv_blocks_context', 'enc').replace('dense_decoder', 'dec') for b in bad[:6]]}; weights after 3 steps differing {nd_w}; loss {l1:.6f} / {l2:.6f}", flush=True)
ops.DETERMINISTIC_WGRAD = False
