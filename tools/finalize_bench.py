"""am_partials_finalize on the row counts a step produces (rows x C floats x 2): time and effective read bandwidth."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import ops  # noqa: E402
from anatomask_amd.hip import lib  # noqa: E402

dev = "cuda:0"
for rows, C in [(131072, 64), (65536, 64), (65536, 32), (16384, 128), (2048, 256), (3280, 32), (256, 512)]:
    part = torch.randn(rows, C, 2, device=dev)
    st = ops.NormStats(C, dev)
    st.count_host = float(rows * 256)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ws = ops._stats_workspace(torch.device(dev), C)

    def run():
        lib().partials_finalize(part.data_ptr(), rows, C, ws.data_ptr(), None, st.count_host, g.data_ptr(), b.data_ptr(), 1e-5, st.mean.data_ptr(),
                                st.rstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), None, None, 0.1, None, None, ops._stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 20
    ref = part.double().sum(0)
    err = (st.mean.double() * st.count_host - ref[:, 0]).abs().max().item() / ref[:, 0].abs().max().item()
    print(f"rows {rows:7d} C {C:4d}: {t * 1e3:7.1f} us  {part.numel() * 4 / t / 1e6:7.0f} GB/s  (sum check rel {err:.1e}; workspace left zero: {bool((ws == 0).all())})", flush=True)
