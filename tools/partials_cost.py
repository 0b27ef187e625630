"""Cost of the statistics epilogue of conv_igemm (per-workgroup per-channel sum / sum of squares for the norm that follows): the decoder
conv 64->64 @128^3, B=8, with and without partials, interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import build as _build  # noqa: E402
if os.environ.get("AM_PC_ABLATE"):
    __import__("anatomask_amd.hip", fromlist=["hip"]).use_library(_build.build(verbose=False, ablate=True))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B, C, S = 8, 64, 128
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
wp = ops.pack_weight(torch.randn(C, C, 3, 3, 3, device=dev) * 0.02, torch.bfloat16, False, False)
y = torch.empty_like(x)
fl = 2.0 * B * S ** 3 * C * C * 27


def timed(fn, iters=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for dbg in [int(v) for v in os.environ.get("AM_PC_ABLATE", "0").split(",")]:
    os.environ["AM_CV_DBG"] = str(dbg)
    t = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, want_partials=True))
    print(f"AM_CV_DBG={dbg}: with partials {t:.3f} ms {fl / t / 1e9:.0f} TF", flush=True)
os.environ["AM_CV_DBG"] = "0"
for rep in range(3):
    a = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y))
    b = timed(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (S, S, S), 3, 1, out=y, want_partials=True))
    print(f"plain {a:.3f} ms {fl / a / 1e9:.0f} TF | with partials {b:.3f} ms {fl / b / 1e9:.0f} TF  (+{(b / a - 1) * 100:.1f} %)", flush=True)
