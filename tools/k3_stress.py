"""Race screen of conv_k3_kernel (a new synchronisation structure: LDS-DMA across raw barriers, counted vmcnt, deferred stores): every
shape is launched N times on the same inputs while a second stream keeps the chip busy with other work (uneven load), and every output
and statistics row must equal the first launch's BIT FOR BIT; the first launch is checked against torch's conv3d.
python tools/k3_stress.py [N]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda:0"
torch.manual_seed(0)
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
for (B, D, H, W, ci, co, dgrad, stats, fused) in [(2, 64, 64, 64, 64, 64, False, True, False), (1, 64, 64, 128, 32, 64, True, False, False),
                                                   (2, 32, 32, 64, 128, 128, False, True, False), (1, 40, 48, 48, 96, 192, False, False, False),
                                                   (4, 128, 128, 128, 64, 64, False, True, False), (2, 64, 64, 80, 64, 64, False, False, True),
                                                   (1, 16, 32, 64, 512, 512, False, False, False)]:
    x = torch.randn(B, D, H, W, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(*((ci, co) if dgrad else (co, ci)), 3, 3, 3, device=dev) / (27 * ci) ** 0.5).to(torch.bfloat16).float()
    wp = ops.pack_weight(w, torch.bfloat16, False, dgrad)
    kw = {}
    if fused:
        kw = dict(ep_scale=torch.rand(co, device=dev) + 0.5, ep_shift=torch.randn(co, device=dev), ep_res=torch.randn(B, D, H, W, co, device=dev).to(torch.bfloat16), ep_act=ops.ACT_RELU6)
    mode = ops.CONV_DGRAD if dgrad else ops.CONV_FWD
    first = first_rows = None
    for it in range(N):
        if it % 3 == 0:                                 # uneven load: a bandwidth-bound copy on another stream, started at varying offsets
            with torch.cuda.stream(side):
                noise_b[: (it + 1) << 18].copy_(noise_a[: (it + 1) << 18])
        out = ops.conv3d(mode, x, wp, None, (D, H, W), 3, 1, want_partials=stats, **kw)
        rows = None
        if stats:
            out, part = out
            rows = part.t[:part.rows].clone()
        if first is None:
            first, first_rows = out.clone(), rows
            xc = x.float().permute(0, 4, 1, 2, 3)
            ref = (F.conv_transpose3d(xc, w, None, padding=1) if dgrad else F.conv3d(xc, w, None, padding=1)).permute(0, 2, 3, 4, 1)
            if fused:
                ref = torch.clamp(ref * kw["ep_scale"] + kw["ep_shift"] + kw["ep_res"].float(), 0, 6)
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-2, err
        else:
            if not torch.equal(out, first) or (rows is not None and not torch.equal(rows, first_rows)):
                bad += 1
                d = (out.float() - first.float()).abs()
                print(f"  MISMATCH at launch {it}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"B{B} {D}x{H}x{W} {ci}->{co} {'dgrad' if dgrad else 'fwd'}{' +stats' if stats else ''}{' fused' if fused else ''}: {N} launches identical to the first (rel err vs torch {err:.2e})" if not bad else "FAILED", flush=True)
    del x, first, out
# round 5: the 32-channel output tile and the transposed instantiation (two / four slabs; two channel tiles), and the 8-wave weight gradient
for (B, D, H, W, ci, co, kind) in [(4, 32, 32, 64, 64, 32, "k3"), (4, 32, 32, 32, 64, 64, "ct"), (8, 16, 32, 32, 128, 64, "ct"), (4, 32, 16, 32, 64, 128, "ct"),
                                   (2, 64, 64, 64, 64, 64, "wg"), (2, 64, 64, 64, 64, 32, "wg"), (4, 32, 32, 32, 192, 96, "wg")]:   # (32-wide cy tiles)
    x = torch.randn(B, D, H, W, ci, device=dev).to(torch.bfloat16)
    if kind == "k3":
        w = (torch.randn(co, ci, 3, 3, 3, device=dev) / (27 * ci) ** 0.5).to(torch.bfloat16).float()
        wp = ops.pack_weight(w, torch.bfloat16, False, False)
        run = lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (D, H, W), 3, 1)
        ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), w, None, padding=1).permute(0, 2, 3, 4, 1)
    elif kind == "ct":
        w = (torch.randn(ci, co, 4, 4, 4, device=dev) / (8 * ci) ** 0.5).to(torch.bfloat16).float()
        wp = ops.pack_weight(w, torch.bfloat16, True, False)
        run = lambda: ops.conv3d(ops.CONVT_FWD, x, wp, None, (2 * D, 2 * H, 2 * W), 4, 2)
        ref = F.conv_transpose3d(x.float().permute(0, 4, 1, 2, 3), w, None, stride=2, padding=1).permute(0, 2, 3, 4, 1)
    else:
        dy = torch.randn(B, D, H, W, co, device=dev).to(torch.bfloat16)
        run = lambda: ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1)
        ref = None
    first = None
    for it in range(N):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise_b[: (it + 1) << 18].copy_(noise_a[: (it + 1) << 18])
        out = run()
        if first is None:
            first = out.clone()
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item() if ref is not None else 0.0
            assert err < 2e-2, err
        elif kind != "wg":                               # (the weight gradient accumulates with fp32 atomics: equal to 1e-5, not bit for bit)
            if not torch.equal(out, first):
                bad += 1
                print(f"  MISMATCH at launch {it}: {int(((out.float() - first.float()).abs() > 0).sum())} elements differ", flush=True)
        elif ((out - first).abs().max() / first.abs().max()).item() > 1e-4:
            bad += 1
            print(f"  MISMATCH (weight gradient) at launch {it}", flush=True)
    torch.cuda.synchronize()
    print(f"B{B} {D}x{H}x{W} {ci}->{co} {kind}: {N} launches consistent (rel err vs torch {err:.2e})" if not bad else "FAILED", flush=True)
    del x, first, out
print("race screen:", "clean" if bad == 0 else f"{bad} mismatching launches")
sys.exit(1 if bad else 0)
