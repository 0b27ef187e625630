"""Isolated rates of the dense / block-sparse norm kernels on the tensors of the STUNet-B step (B=16, bf16), same process A/B of the
workgroup-count targets (ablation build: python tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/norm_bench.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def dense(S, C, act):
    x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
    d = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = ops.NormStats(C, dev)
    st.count_host = float(B * S ** 3)
    ops.chan_stats(x, None, 0, st); ops.norm_finalize(st, gam, bet, 1e-5)
    nb = x.numel() * 2
    sc = ops.NormBwdScratch(C, dev)
    L = __import__("anatomask_amd.hip", fromlist=["lib"]).lib()
    s = torch.cuda.current_stream().cuda_stream
    ws_b, ws_x = ops._bwd_workspaces(x.device, C)

    def red():
        L.norm_bwd_reduce(1, d.data_ptr(), None, x.data_ptr(), B, S, S, S, C, None, 0, 1, 1, 1, st.mean.data_ptr(), st.rstd.data_ptr(), act, 0,
                          ws_b.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), None, 0, None, float(st.count_host), gam.data_ptr(),
                          sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), None, None, None, None, s)

    def app():
        L.norm_bwd_apply(1, d.data_ptr(), None, x.data_ptr(), B, S, S, S, C, None, 0, 1, 1, 1, st.mean.data_ptr(), st.rstd.data_ptr(),
                         sc.k[0].data_ptr(), sc.k[1].data_ptr(), sc.k[2].data_ptr(), act, dx.data_ptr(), None, None, ws_x.data_ptr(),
                         st.scale.data_ptr(), st.shift.data_ptr(), None, 0, 1, s)
    out = {}
    for tgt in os.environ.get("TARGETS", "1024").split(","):
        os.environ["AM_RED_TARGET"] = tgt; os.environ["AM_MAP_TARGET"] = str(2 * int(tgt))
        t1, t2, t3 = timed(red), timed(app), timed(lambda: ops.norm_apply(x, st, act, out=y))
        print(f"[{B},{S}^3,{C}] act {act} targets {tgt:>5s}/{2 * int(tgt):<5d}: bwd_reduce {t1 * 1e3:7.1f} us {2 * nb / t1 / 1e9:5.2f} TB/s | bwd_apply {t2 * 1e3:7.1f} us "
              f"{3 * nb / t2 / 1e9:5.2f} TB/s | apply {t3 * 1e3:7.1f} us {2 * nb / t3 / 1e9:5.2f} TB/s", flush=True)


for S, C, act in ((128, 64, ops.ACT_RELU6), (128, 32, ops.ACT_RELU6), (64, 128, ops.ACT_RELU6), (64, 64, ops.ACT_NONE), (32, 256, ops.ACT_RELU6)):
    dense(S, C, act)
    torch.cuda.empty_cache()
