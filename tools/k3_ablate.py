"""Timing ablation of conv_k3_kernel (tools build only: python tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/k3_ablate.py [B]).
AM_K3_DBG bits: 1 no stores, 2 no brick DMA, 4 no weight DMA, 8 no MFMAs.  Results are wrong on purpose; only the time is read."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [(64, 64, 128), (128, 128, 64)]
variants = [("shipped", 0), ("no stores", 1), ("no brick DMA", 2), ("no weight DMA", 4), ("no DMA at all", 6), ("no MFMA", 8), ("no DMA, no stores", 7), ("only the skeleton", 15)]


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for ci, co, S in shapes:
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty(B, S, S, S, co, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * B * S ** 3 * ci * co * 27
    for rep in range(2):
        row = []
        for name, bits in variants:
            os.environ["AM_K3_DBG"] = str(bits)
            t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y))
            row.append(f"{name}: {t:.3f} ms ({fl / t / 1e9:.0f})")
        os.environ["AM_K3_DBG"] = "0"
        t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, want_partials=True))
        row.append(f"with statistics (+alloc): {t:.3f} ms ({fl / t / 1e9:.0f})")
        print(f"{ci}->{co} @{S}^3 B={B}: " + " | ".join(row), flush=True)
    del x, y

# in-kernel stamps (AM_K3_DBG = 16): where a wave's cycles go
for dbg, ci, co, S in [(d, *sh) for sh in shapes for d in ("16", "48")]:
    os.environ["AM_K3_DBG"] = dbg
    x = torch.randn(B, S, S, S, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    for _ in range(3):
        out, part = ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, want_partials=True)
    torch.cuda.synchronize()
    rows = part.t[:part.rows].view(torch.int32).reshape(part.rows, -1)[:, :6].cpu().double()
    rows = rows.reshape(-1, 8, 6)
    nrun = (B * S ** 3 // 512) * (ci // 32) * 9 * (co // 64) / rows.shape[0]
    for half, name in ((slice(0, 4), "X"), (slice(4, 8), "Y")):
        m = rows[:, half, :].reshape(-1, 6).median(0).values
        clk = (m[0] + m[1] + m[2] + m[3] + m[4]) / (m[5] * 10.0) if m[5] > 0 else 0.0   # s_memrealtime ticks at 100 MHz
        print(f"[dbg {dbg}] {ci}->{co} @{S}^3 {name} waves, cycles per run: L {m[0] / nrun:.0f}  barrier after L {m[1] / nrun:.0f}  M {m[2] / nrun:.0f}  "
              f"barrier after M {m[3] / nrun:.0f}  | epilogue total {m[4] / (nrun / 9 / (ci // 32)):.0f} per unit | clock {clk:.2f} GHz", flush=True)
os.environ["AM_K3_DBG"] = "0"
