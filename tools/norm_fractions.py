"""HBM fractions of the NORM families of a step (hygiene #12): algorithmic bytes of every norm_apply / norm_backward call of ONE
step (logged at the Python level: tensor shape x active fraction x element size, per kernel family), divided by the family's time in the committed
kernel-stats CSVs, in-step (side stream on) and isolated (side stream off).

    python tools/norm_fractions.py log  > gpurun_out/<tag>/norm_bytes.json          # on the GPU box: one step, bytes per family
    python tools/norm_fractions.py table gpurun_out/<tag>/norm_bytes.json profiles/<tag>_step_kernel_stats_b16.csv profiles/<tag>_step_kernel_stats_b16_isolated.csv
Bytes counted per call (bf16 storage, active voxels only):  norm_apply: 1 read + 1 write (+ 1 read with a residual);
norm_bwd_reduce: 2 reads (dy, x) (+ 1: the saved output, when a residual forces it);  norm_bwd_apply: 2 reads + 1 write (+ 1 read saved output, + 1 write shortcut gradient)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def log():
    import torch
    from anatomask_amd import modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer
    B = 16
    dev = torch.device("cuda:0")
    kw = M.STUNET_CONFIGS["B"]
    torch.manual_seed(0)
    model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
    tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1, distributed=False)
    tr.set_epoch(500)
    x = torch.randn(B, 1, 128, 128, 128, device=dev)
    tr.step(x, epoch=500)
    acc = {}

    def add(fam, nbytes):
        a = acc.setdefault(fam, [0, 0.0]); a[0] += 1; a[1] += nbytes

    def frac(t, mask, bshift):
        if mask is None:
            return 1.0
        n = mask.n_active if mask.n_active is not None else int(mask.t.count_nonzero().item())
        return n * (1 << (3 * bshift)) / (t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3])
    o_apply, o_bwd = ops.norm_apply, ops.norm_backward

    def norm_apply(x_, st, act, mask=None, bshift=0, res=None, stem=None, fill=None, out=None):
        nb = x_.numel() * x_.element_size() * (frac(x_, mask, bshift) if fill is None else 1.0)
        add("norm_apply_rows" if (mask is not None and fill is None) else "norm_apply", nb * (2 + (1 if res is not None else 0)))
        return o_apply(x_, st, act, mask, bshift, res, stem, fill, out)

    def norm_backward(dout, out, x_, st, gamma, act, mask, bshift, dgamma, dbeta, dtoken=None, fill=False, dx=None, dres=None, scratch=None,
                      dbeta2=None, dxsum=None, reduced=None):
        nb = x_.numel() * x_.element_size() * frac(x_, mask, bshift)
        sfx = "_rows" if (mask is not None and not fill) else ""
        if reduced is None:
            add("norm_bwd_reduce" + sfx, nb * (2 + (1 if out is not None else 0)))
        add("norm_bwd_apply" + sfx, nb * (3 + (1 if out is not None else 0) + (1 if dres is not None else 0)))
        return o_bwd(dout, out, x_, st, gamma, act, mask, bshift, dgamma, dbeta, dtoken, fill, dx, dres, scratch, dbeta2, dxsum, reduced)
    ops.norm_apply, ops.norm_backward = norm_apply, norm_backward
    try:
        tr.step(x, epoch=500)
        torch.cuda.synchronize()
    finally:
        ops.norm_apply, ops.norm_backward = o_apply, o_bwd
    print(json.dumps({k: {"calls": v[0], "bytes": v[1]} for k, v in acc.items()}))


def table(jpath, csv_step, csv_iso):
    acc = json.load(open(jpath))
    fam_kernels = {"norm_apply": ["norm_apply_kernel"], "norm_apply_rows": ["norm_apply_rows_kernel"], "norm_bwd_reduce": ["norm_bwd_reduce_kernel"],
                   "norm_bwd_reduce_rows": ["norm_bwd_reduce_rows_kernel"], "norm_bwd_apply": ["norm_bwd_apply_kernel"], "norm_bwd_apply_rows": ["norm_bwd_apply_rows_kernel"]}

    def ms(path, subs):
        rows = list(csv.DictReader(open(path)))
        steps = [int(r["Calls"]) for r in rows if "adamw_ema_kernel" in r["Name"]][0]
        sel = [r for r in rows if "proj_" not in r["Name"] and any(s + "<" in r["Name"] or s + "(" in r["Name"] for s in subs)]
        return sum(float(r["TotalDurationNs"]) for r in sel) / steps / 1e6, sum(int(r["Calls"]) for r in sel) / steps
    print(f"# HBM fractions of the norm families of one step (STUNet-B 128^3 bf16, B=16): algorithmic bytes (tools/norm_fractions.py) / kernel time\n# in-step: `{csv_step}` (weight gradients on the side stream); isolated: `{csv_iso}` (side stream off: every kernel alone on the chip)\n")
    print("| family | launches / step | algorithmic GB / step | isolated ms | isolated TB/s (of 8) | in-step ms | in-step TB/s (of 8) |")
    print("|---|---|---|---|---|---|---|")
    tb = ti = ts = 0.0
    for fam, subs in fam_kernels.items():
        if fam not in acc:
            continue
        gb = acc[fam]["bytes"] / 1e9
        m_iso, n = ms(csv_iso, subs)
        m_step, _ = ms(csv_step, subs)
        tb += gb; ti += m_iso; ts += m_step
        print(f"| `{fam}` | {n:.0f} | {gb:.1f} | {m_iso:.2f} | {gb / m_iso:.2f} ({gb / m_iso / 8:.0%}) | {m_step:.2f} | {gb / m_step:.2f} ({gb / m_step / 8:.0%}) |")
    print(f"| all | | {tb:.1f} | {ti:.2f} | {tb / ti:.2f} ({tb / ti / 8:.0%}) | {ts:.2f} | {tb / ts:.2f} ({tb / ts / 8:.0%}) |")
    print("\nFamily averages over all launches of a step, small tensors included; on the decoder's 4.3 GB tensors every one of these kernels streams at 5.0-5.5 TB/s alone\n(`tools/norm_bench.py`, `profiles/r04_z_norm_bench_b16.txt`), the ceiling these boxes give a streaming kernel;\nthe in-step figures are the same launches queued behind the persistent weight-gradient grid of the side stream (the isolated times sum to the step: DESIGN.md 4,\n`profiles/r04_experiments.md` 2).")


if __name__ == "__main__":
    if sys.argv[1] == "log":
        log()
    else:
        table(*sys.argv[2:5])
