"""Can an HBM-bound pass run UNDER the persistent 8-wave weight gradient?  Times, at the bench batch, the 64->64 @128^3 weight gradient alone,
a streaming pass alone (norm_apply, 56 VGPRs: fits beside two 220-register waves per SIMD; norm_backward: 98-142 VGPRs, does not), and both
at once on two streams:  python tools/overlap_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
S, C = 128, 64
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
dy = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
y = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
da = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
out = torch.empty_like(y)
st = ops.NormStats(C, dev)
ops.chan_stats(y, None, 0, st)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
st.scale.fill_(1.0); st.shift.zero_(); st.mean.zero_(); st.rstd.fill_(1.0); st.count_host = float(B * S ** 3)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dx = torch.empty_like(y)
side = torch.cuda.Stream()


def wg():
    ops.conv3d_wgrad(ops.CONV_FWD, x, dy, 3, 1)


def apply3():
    for _ in range(3):
        ops.norm_apply(y, st, ops.ACT_RELU6, out=out)


def bwd():
    ops.norm_backward(da, None, y, st, gamma, ops.ACT_RELU6, None, 0, dg, db, dx=dx)


def timed(fn_main, fn_side=None, n=6):
    for _ in range(2):
        if fn_side:
            with torch.cuda.stream(side):
                fn_side()
        fn_main()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        if fn_side:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn_side()
        fn_main()
        if fn_side:
            torch.cuda.current_stream().wait_stream(side)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


t_wg, t_ap, t_bw = timed(wg), timed(apply3), timed(bwd)
print(f"B={B}: weight gradient alone {t_wg:.2f} ms | 3 x norm_apply alone {t_ap:.2f} ms | norm_backward (reduce + apply) alone {t_bw:.2f} ms")
print(f"  weight gradient (side) + 3 x norm_apply (main): {timed(apply3, wg):.2f} ms  (sum {t_wg + t_ap:.2f}, max {max(t_wg, t_ap):.2f})")
print(f"  weight gradient (side) + norm_backward (main):  {timed(bwd, wg):.2f} ms  (sum {t_wg + t_bw:.2f}, max {max(t_wg, t_bw):.2f})")
