"""Correctness + rate of the persistent LDS-DMA k3 s1 kernel (conv_k3.hip) against torch's conv3d: python tools/k3_check.py [quick]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
FAILED = []
torch.manual_seed(0)


def ref_conv(x, w, bias, dgrad):
    xc = x.float().permute(0, 4, 1, 2, 3)
    if dgrad:   # data gradient of y = conv(x', w): dx' = conv_transpose(dy, w)
        y = F.conv_transpose3d(xc, w, None, stride=1, padding=1)
    else:
        y = F.conv3d(xc, w, bias, padding=1)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def check(B, D, H, W, ci, co, dgrad=False, bias=True, stats=False, fused=False, iters=0):
    x = torch.randn(B, D, H, W, ci, device=dev).to(torch.bfloat16)
    wshape = (ci, co, 3, 3, 3) if dgrad else (co, ci, 3, 3, 3)
    w = (torch.randn(*wshape, device=dev) * (1.0 / (27 * ci) ** 0.5)).to(torch.bfloat16).float()
    wp = ops.pack_weight(w, torch.bfloat16, False, dgrad)
    b = torch.randn(co, device=dev) if (bias and not dgrad) else None
    kw = {}
    res = None
    if fused:
        sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
        res = torch.randn(B, D, H, W, co, device=dev).to(torch.bfloat16)
        kw = dict(ep_scale=sc, ep_shift=sh, ep_res=res, ep_act=ops.ACT_RELU6)
    out = ops.conv3d(ops.CONV_DGRAD if dgrad else ops.CONV_FWD, x, wp, b, (D, H, W), 3, 1, want_partials=stats, **kw)
    part = None
    if stats:
        out, part = out
    ref = ref_conv(x, w, b, dgrad)
    if fused:
        ref = torch.clamp(ref * sc + sh + res.float(), 0.0, 6.0)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    msg = f"B{B} {D}x{H}x{W} {ci}->{co} {'dgrad' if dgrad else 'fwd'}{' fused' if fused else ''}: rel-max err {err:.2e}"
    if part is not None:
        s = part.t[:part.rows].double().sum(0)
        o = out.double().reshape(-1, co)
        e1 = ((s[:, 0] - o.sum(0)).abs().max() / o.sum(0).abs().max()).item()
        e2 = ((s[:, 1] - (o * o).sum(0)).abs().max() / (o * o).sum(0).abs().max()).item()
        msg += f"  stats rows {part.rows}: sum {e1:.1e} sumsq {e2:.1e}"
    if iters:
        y = torch.empty_like(out)
        fn = lambda: ops.conv3d(ops.CONV_DGRAD if dgrad else ops.CONV_FWD, x, wp, b, (D, H, W), 3, 1, out=y, **kw)
        for _ in range(3):
            fn()
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1_.record(); e1_.synchronize()
        t = e0.elapsed_time(e1_) / iters
        msg += f"  {t:.3f} ms {2.0 * B * D * H * W * ci * co * 27 / t / 1e9:.0f} TF"
    bad = ~torch.isfinite(out.float()) | ((out.float() - ref).abs() > 0.05 * ref.abs().max())
    if bad.any():
        idx = bad.nonzero()
        msg += f"  BAD {bad.sum().item()} of {bad.numel()}: first {idx[0].tolist()} last {idx[-1].tolist()}; d {sorted(set(idx[:, 1].tolist()))[:20]} h%4 {sorted(set((idx[:, 2] % 4).tolist()))} c//8 {sorted(set((idx[:, 4] // 8).tolist()))}"
        FAILED.append(msg)
    print(msg, flush=True)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
check(1, 64, 64, 64, 64, 64)
check(1, 64, 64, 64, 64, 64, dgrad=True)
check(1, 64, 64, 80, 64, 64, stats=True)
check(1, 64, 64, 64, 32, 64, dgrad=True)
check(1, 64, 32, 64, 128, 128, stats=True)
check(2, 64, 64, 64, 64, 64, fused=True)
check(1, 72, 36, 64, 64, 64, stats=True, bias=False)
if not quick:
    check(8, 128, 128, 128, 64, 64, iters=10)
    check(8, 128, 128, 128, 64, 64, dgrad=True, iters=10)
    check(8, 64, 64, 64, 128, 128, iters=10)
    check(8, 32, 32, 32, 256, 256, iters=10)
    check(8, 128, 128, 128, 32, 64, dgrad=True, iters=10)

assert not FAILED, FAILED
