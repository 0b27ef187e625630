"""Dense transposed convs on the small ragged q grids of the STUNet-L / H decoders: conv_gather.hip (rows = all voxels of the coarse grid,
8 output-parity classes of workgroups) against conv_igemm's bricks (tools build: AM_GA_MAXBS=-1 switches the gather kernel off).
    python tools/convt_small_grids.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=15):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for name, B, c, S in [("STUNet-L 1024->1024 @10->20", 4, 1024, 10), ("STUNet-L 512->512 @20->40", 4, 512, 20), ("STUNet-L 256->256 @40->80", 4, 256, 40),
                      ("STUNet-H 1536->1536 @12->24", 2, 1536, 12), ("STUNet-H 768->768 @24->48", 2, 768, 24)]:
    x = torch.randn(B, S, S, S, c, device=dev).to(torch.bfloat16)
    wp = ops.pack_weight(torch.randn(c, c, 4, 4, 4, device=dev) * 0.02, torch.bfloat16, True, False)
    bias = torch.randn(c, device=dev)
    out, ys = [], []
    for rep in range(2):
        for tag, v in (("bricks", "-1"), ("gather", "2")):
            os.environ["AM_GA_MAXBS"] = v
            y = torch.empty(B, 2 * S, 2 * S, 2 * S, c, device=dev, dtype=torch.bfloat16)
            t = timed(lambda: ops.conv3d(ops.CONVT_FWD, x, wp, bias, (2 * S,) * 3, 4, 2, out=y))
            ys.append(y)
            out.append(f"{tag} {t * 1e3:.0f} us {2.0 * B * (2 * S) ** 3 * c * c * 8 / t / 1e9:.0f} TF")
    err = (ys[0].float() - ys[1].float()).abs().max().item() / ys[0].float().abs().max().item()
    print(f"{name}: " + " | ".join(out) + f" | max rel diff {err:.1e}", flush=True)
    # its data gradient: a k4 s2 conv of dy (4^3 window of the fine grid per coarse voxel)
    dy = torch.randn(B, 2 * S, 2 * S, 2 * S, c, device=dev).to(torch.bfloat16)
    wb = ops.pack_weight(torch.randn(c, c, 4, 4, 4, device=dev) * 0.02, torch.bfloat16, True, True)
    out, ys = [], []
    for rep in range(2):
        for tag, v in (("bricks", "-1"), ("gather", "2")):
            os.environ["AM_GA_MAXBS"] = v
            dx = torch.empty(B, S, S, S, c, device=dev, dtype=torch.bfloat16)
            t = timed(lambda: ops.conv3d(ops.CONVT_DGRAD, dy, wb, None, (S,) * 3, 4, 2, out=dx))
            ys.append(dx)
            out.append(f"{tag} {t * 1e3:.0f} us {2.0 * B * (2 * S) ** 3 * c * c * 8 / t / 1e9:.0f} TF")
    err = (ys[0].float() - ys[1].float()).abs().max().item() / ys[0].float().abs().max().item()
    print(f"   data gradient: " + " | ".join(out) + f" | max rel diff {err:.1e}", flush=True)
    del x, ys, dy
os.environ.pop("AM_GA_MAXBS", None)
