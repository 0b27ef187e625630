"""The voxel-list gather kernel (conv_gather.hip) against the brick kernel on the deep block-sparse levels, tools build:
AM_GA_MAXBS = -1 (off: conv_igemm bricks), 1 (patches up to 2^3: the default), 2 (also 4^3); AM_GA_WIDE = 0 / 1 (64- / 128-channel tiles).
    python tools/gather_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
__import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=20):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


# (model, B, mask grid f, keep fraction, [(channels, level grid, bshift)])
for name, B, f, keep, levels in [("STUNet-B 128^3 m0.6 B=16", 16, 8, 0.4, [(128, 32, 2), (256, 16, 1), (512, 8, 0)]),
                                 ("STUNet-L 160^3 m0.7 B=4", 4, 10, 0.3, [(256, 40, 2), (512, 20, 1), (1024, 10, 0)]),
                                 ("STUNet-H 192^3 m0.6 B=2", 2, 12, 0.4, [(384, 48, 2), (768, 24, 1), (1536, 12, 0)])]:
    torch.manual_seed(0)
    L = f ** 3
    idx = torch.rand(B, L, device=dev).argsort(1)[:, :int(round(L * keep))]
    mk = torch.zeros(B, L, dtype=torch.uint8, device=dev).scatter_(1, idx, 1).view(B, f, f, f)
    mi = ops.MaskInfo(mk, n_active=int(mk.sum()))
    for c, S, bs in levels:
        x = torch.randn(B, S, S, S, c, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        wp = ops.pack_weight(torch.randn(c, c, 3, 3, 3, device=dev) * 0.02, torch.bfloat16, False, False)
        fl = 2.0 * int(mk.sum()) * (1 << (3 * bs)) * 27 * c * c
        out = []
        for rep in range(2):
            for tag, env in (("bricks", {"AM_GA_MAXBS": "-1"}), ("gather64", {"AM_GA_MAXBS": "2", "AM_GA_WIDE": "0"}), ("gather128", {"AM_GA_MAXBS": "2", "AM_GA_WIDE": "1"})):
                os.environ.update(env)
                t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S,) * 3, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y, want_partials=True))
                out.append(f"{tag} {t * 1e3:.0f} us {fl / t / 1e9:.0f} TF")
        print(f"{name} {c}->{c} @{S} patch {1 << bs}: " + " | ".join(out), flush=True)
        del x, y
for k in ("AM_GA_MAXBS", "AM_GA_WIDE"):
    os.environ.pop(k, None)
