"""Brick-walk slots of the weight gradient (tools build: AM_WG_ROUNDS = whole rounds of resident workgroups the launch aims for) on the
deep block-sparse levels, where the per-workgroup flush ([taps][64][64] float atomics) outweighs the contraction.
    python tools/wgrad_rounds.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
keep = torch.rand(B, 8, 8, 8, device=dev).flatten(1).argsort(1)[:, :205]
mk = torch.zeros(B, 512, dtype=torch.uint8, device=dev).scatter_(1, keep, 1).view(B, 8, 8, 8)
mi = ops.MaskInfo(mk)


def timed(fn, iters=15):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for cx, cy, S, bs, stride, sparse in [(128, 128, 32, 2, 1, True), (256, 256, 16, 1, 1, True), (512, 512, 8, 0, 1, True), (128, 256, 32, 2, 2, True), (256, 512, 16, 1, 2, True),
                                      (256, 256, 16, 1, 1, False), (512, 512, 8, 0, 1, False), (512, 512, 16, 0, 1, False), (256, 256, 32, 0, 1, False), (256, 128, 32, 0, 1, False),
                                      (128, 128, 64, 0, 1, False), (128, 64, 64, 0, 1, False), (64, 64, 128, 0, 1, False), (512, 256, 16, 0, 1, False)][int(os.environ.get('AM_FIRST', '0')):]:
    So = S // stride
    x = torch.randn(B, S, S, S, cx, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, So, So, So, cy, device=dev).to(torch.bfloat16)
    out = []
    for rep in range(2):
        for r in os.environ.get("AM_ROUNDS", "0,1,2,3,4").split(","):
            if r == "0":
                os.environ.pop("AM_WG_ROUNDS", None)
            else:
                os.environ["AM_WG_ROUNDS"] = r
            kw = dict(x_mask=mi, x_bshift=bs, y_mask=mi, y_bshift=bs - (1 if stride == 2 else 0)) if sparse else {}
            t = timed(lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, stride, **kw))
            out.append(f"R{r}: {t * 1e3:.0f} us")
    os.environ.pop("AM_WG_ROUNDS", None)
    print(f"wgrad k3 s{stride} {cx}->{cy} @{S} {'sparse' if sparse else 'dense'}: " + " | ".join(out), flush=True)
