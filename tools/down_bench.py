"""The block-sparse downsampling convolutions of the student encoder (STUNet-B 128^3, mask 0.6): k3 s2 and k1 s2, forward / data
gradient / weight gradient, with the masks the engine passes.  ms, TFLOP/s on active voxels, GB/s on active bytes.
    python tools/down_bench.py [B]          (through tools/with_lib.py to compare builds: tools/ab.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
keep = torch.rand(B, 8, 8, 8, device=dev).flatten(1).argsort(1)[:, :205]
mk = torch.zeros(B, 512, dtype=torch.uint8, device=dev).scatter_(1, keep, 1).view(B, 8, 8, 8)
mi = ops.MaskInfo(mk)
ACT = 205 / 512


def timed(fn, iters=15):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def up(m, f):                                    # patch mask -> voxel mask (for zeroing inactive voxels like the engine's tensors)
    return m.repeat_interleave(f, 1).repeat_interleave(f, 2).repeat_interleave(f, 3).unsqueeze(-1)


only = os.environ.get("AM_DOWN_ONLY", "")
for ci, co, S, bs in [(32, 64, 128, 4), (64, 128, 64, 3), (128, 256, 32, 2), (256, 512, 16, 1)]:
    So = S // 2
    x = (torch.randn(B, S, S, S, ci, device=dev) * up(mk, S // 8)).to(torch.bfloat16)
    dy = (torch.randn(B, So, So, So, co, device=dev) * up(mk, So // 8)).to(torch.bfloat16)
    y = torch.empty(B, So, So, So, co, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(B, S, S, S, ci, device=dev, dtype=torch.bfloat16)
    byts = (x.numel() + y.numel()) * 2 * ACT
    line = [f"{ci}->{co} @{S}->{So} ({byts / 1e6:.0f} MB active)"]
    for k in (3, 1):
        w = torch.randn(co, ci, k, k, k, device=dev) * 0.02
        wf, wb = ops.pack_weight(w, torch.bfloat16, False, False), ops.pack_weight(w, torch.bfloat16, False, True)
        fl = 2.0 * B * So ** 3 * ci * co * k ** 3 * ACT
        for name, fn in (("fwd", lambda: ops.conv3d(ops.CONV_FWD, x, wf, None, (So,) * 3, k, 2, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs - 1, out=y)),
                         ("dgrad", lambda: ops.conv3d(ops.CONV_DGRAD, dy, wb, None, (S,) * 3, k, 2, in_mask=mi, in_bshift=bs - 1, out_mask=mi, out_bshift=bs, out=dx)),
                         ("wgrad", lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, k, 2, x_mask=mi, x_bshift=bs, y_mask=mi, y_bshift=bs - 1))):
            if only and only not in f"k{k}{name}":
                continue
            t = timed(fn)
            line.append(f"k{k} {name} {t * 1e3:.0f} us {fl / t / 1e9:.0f} TF {byts / t / 1e6:.0f} GB/s")
    print(" | ".join(line), flush=True)

# the stride-1 block-sparse layers of the same levels (weight gradient, forward, data gradient)
for c, S, bs in [(32, 128, 4), (64, 64, 3), (128, 32, 2), (256, 16, 1), (512, 8, 0)]:
    x = (torch.randn(B, S, S, S, c, device=dev) * up(mk, S // 8)).to(torch.bfloat16)
    dy = (torch.randn(B, S, S, S, c, device=dev) * up(mk, S // 8)).to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.02
    wf, wb = ops.pack_weight(w, torch.bfloat16, False, False), ops.pack_weight(w, torch.bfloat16, False, True)
    fl = 2.0 * B * S ** 3 * c * c * 27 * ACT
    line = [f"{c}->{c} @{S} s1"]
    for name, fn in (("fwd", lambda: ops.conv3d(ops.CONV_FWD, x, wf, None, (S,) * 3, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y)),
                     ("dgrad", lambda: ops.conv3d(ops.CONV_DGRAD, dy, wb, None, (S,) * 3, 3, 1, in_mask=mi, in_bshift=bs, out_mask=mi, out_bshift=bs, out=y)),
                     ("wgrad", lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1, x_mask=mi, x_bshift=bs, y_mask=mi, y_bshift=bs))):
        if only and only not in f"k3{name}":
            continue
        t = timed(fn)
        line.append(f"k3 {name} {t * 1e3:.0f} us {fl / t / 1e9:.0f} TF")
    print(" | ".join(line), flush=True)
