"""CPU calibration of the end-to-end parity tolerances (a script, not a test module; run: python tests/calibrate_tolerances.py).

Evaluates the oracle's student forward + backward on the tiny fixture config three ways -- fp64, fp32, and fp32 with every
tensor the HIP path STORES in bf16 rounded to bf16 (conv operands and outputs, norm / activation outputs, their gradients) --
and prints, per parameter tensor, the relative L2 error and cosine of fp32-vs-fp64 and bf16-storage-vs-fp64.  The bounds
asserted in tests/test_e2e_gpu.py are these floors times a small factor (stated there).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import anatomask_oracle as O  # noqa: E402


class _R(torch.autograd.Function):
    """round to bf16 in forward AND round the incoming gradient (what bf16 activation / gradient storage does)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def grads(cfg, W, x, mask, dtype=torch.float32, bf16=False):
    Wd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in W.items()}
    saved = (F.conv3d, F.conv_transpose3d, O.sparse_instance_norm, O.batch_norm3d)
    if bf16:
        c3, ct = F.conv3d, F.conv_transpose3d
        r = _R.apply
        single = lambda t: t.shape[1] == 1                                        # noqa: E731  C == 1 tensors stay fp32 in the HIP path
        F.conv3d = lambda x_, w, b=None, **kw: (lambda y: y if single(y) else r(y))(c3(x_ if single(x_) else r(x_), w if single(x_) else r(w), b, **kw))
        F.conv_transpose3d = lambda x_, w, b=None, **kw: r(ct(r(x_), r(w), b, **kw))
        sin, bn = O.sparse_instance_norm, O.batch_norm3d
        O.sparse_instance_norm = lambda *a, **k: r(sin(*a, **k))
        O.batch_norm3d = lambda *a, **k: r(bn(*a, **k))
    try:
        loss, _, g, _ = O.student_loss_and_grads(cfg, Wd, x.to(dtype), mask)
    finally:
        F.conv3d, F.conv_transpose3d, O.sparse_instance_norm, O.batch_norm3d = saved
    return float(loss), g


def compare(name, ga, gb):
    rows = []
    for k in ga:
        if ga[k] is None or float(gb[k].norm()) < 1e-9:
            continue
        a, b = ga[k].double().flatten(), gb[k].double().flatten()
        rows.append((k, float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm() + 1e-300)), float(a.norm() / b.norm()), a.numel()))
    errs = np.array([r[1] for r in rows])
    print(f"--- {name}: per-tensor rel L2 error  median {np.median(errs):.2e}  p90 {np.percentile(errs, 90):.2e}  max {errs.max():.2e}")
    for r in sorted(rows, key=lambda r: -r[1])[:8]:
        print(f"    {r[0]:70s} n={r[4]:7d} rel {r[1]:.2e} cos {r[2]:.6f} norm ratio {r[3]:.4f}")
    return rows


def main():
    torch.set_num_threads(8)
    cfg = O.Config([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 48, 64), 0.6)
    x = torch.from_numpy(np.random.RandomState(1234).standard_normal((2, 1, *cfg.input_size)).astype(np.float32))
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(7))
    for tag, W in (("closed-form sin weights (round 1)", O.closed_form_state(cfg)), ("seeded reference-initialiser weights", O.seeded_state(cfg, 0))):
        print(f"==== {tag}")
        l64, g64 = grads(cfg, W, x, mask, torch.float64)
        l32, g32 = grads(cfg, W, x, mask, torch.float32)
        lbf, gbf = grads(cfg, W, x, mask, torch.float32, bf16=True)
        print(f"loss fp64 {l64:.8f}  fp32 {l32:.8f} (rel {abs(l32 - l64) / l64:.1e})  bf16-storage {lbf:.8f} (rel {abs(lbf - l64) / l64:.1e})")
        compare("fp32 vs fp64", g32, g64)
        compare("bf16 storage vs fp64", gbf, g64)


if __name__ == "__main__":
    main()
