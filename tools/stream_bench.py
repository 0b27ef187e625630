"""Achieved HBM GB/s of the streaming kernels on the decoder level-3 tensor [2,128,128,128,64] bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B, S, C = 2, 128, 64
x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
d = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
dx = torch.empty_like(x)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
st = ops.NormStats(C, dev)
st.count_host = float(B * S ** 3)
nb = x.numel() * 2


def timed(fn, name, nbytes, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / iters
    print(f"{name:28s} {t*1e3:8.1f} us  {nbytes / t / 1e6:8.1f} GB/s")


timed(lambda: ops.chan_stats(x, None, 0, st), "chan_stats (1 read)", nb)
ops.norm_finalize(st, gam, bet, 1e-5)
timed(lambda: ops.norm_apply(x, st, ops.ACT_RELU6, out=y), "norm_apply (1r 1w)", 2 * nb)
timed(lambda: ops.norm_apply(x, st, ops.ACT_NONE, res=d, out=y), "norm_apply+res (2r 1w)", 3 * nb)
sc = ops.NormBwdScratch(C, dev)
bsum = torch.zeros(ops.NREP, C, 3, device=dev, dtype=torch.float64)
L = __import__("anatomask_amd.hip", fromlist=["lib"]).lib()
s = torch.cuda.current_stream().cuda_stream
timed(lambda: L.norm_bwd_reduce(1, d.data_ptr(), y.data_ptr(), x.data_ptr(), B, S, S, S, C, None, 0, 1, 1, 1, st.mean.data_ptr(),
                                st.rstd.data_ptr(), ops.ACT_RELU6, 0, bsum.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), None, 0, None, 0.0, None, None, None, None, None, None, None, None, s), "bwd_reduce (3 reads)", 3 * nb)
timed(lambda: ops.norm_backward(d, y, x, st, gam, ops.ACT_RELU6, None, 0, None, None, dx=dx, scratch=sc), "bwd reduce+apply (6r 1w)", 7 * nb)
out = torch.zeros(C, device=dev)
timed(lambda: ops.chan_sum(x, None, 0, out), "chan_sum (1 read)", nb)
a = torch.empty_like(x)
timed(lambda: a.copy_(x), "torch copy (1r 1w)", 2 * nb)

# ---- block-sparse tensors (encoder level 0: [8,128,128,128,32] bf16, 16^3 patches): active-patch row walk
del x, d, y, dx, a
Bs, Cs = 8, 32
xs = torch.randn(Bs, S, S, S, Cs, device=dev).to(torch.bfloat16)
ys = torch.empty_like(xs)
sts = ops.NormStats(Cs, dev)
ops.norm_fold_running(sts, torch.ones(Cs, device=dev), torch.zeros(Cs, device=dev), torch.zeros(Cs, device=dev), torch.ones(Cs, device=dev), 1e-5)
for frac in (0.4, 1.0):
    L, keep = 512, round(512 * frac)
    mk = ops.mask_sampler(torch.zeros(Bs, L, device=dev), torch.rand(Bs, L, device=dev), keep, 0)
    mi = ops.MaskInfo(mk.view(Bs, 8, 8, 8), n_active=Bs * keep)
    nbs = Bs * keep * 4096 * Cs * 2
    timed(lambda: ops.norm_apply(xs, sts, ops.ACT_LRELU, mi, 4, out=ys), f"sparse {frac:.0%} rows apply (1r 1w)", 2 * nbs)
    timed(lambda: ops.chan_stats(xs, mi, 4, sts), f"sparse {frac:.0%} rows stats (1r)", nbs)
    mi_nolist = ops.MaskInfo(mk.view(Bs, 8, 8, 8), n_active=0)
    mi_nolist._list = torch.zeros(1, device=dev, dtype=torch.int32)       # n_active 0 -> linear kernel with per-voxel mask lookups
    timed(lambda: ops.norm_apply(xs, sts, ops.ACT_LRELU, mi_nolist, 4, out=ys), f"sparse {frac:.0%} linear apply", 2 * nbs)
