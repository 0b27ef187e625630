"""profiles/<name>.md: per-kernel HBM fractions of the streaming kernels of the bench step, from a committed rocprofv3
`--kernel-trace --stats` CSV:  algorithmic bytes per launch / average launch duration / 8 TB/s.

    python tools/hbm_fractions.py profiles/r02_c_step_kernel_stats_b16.csv [B=16] > profiles/r02_c_kernel_hbm_fractions.md

Only kernels that are launched on ONE shape per step are listed from the step CSV (the norm kernels run on a dozen shapes per step:
their fractions come from the single-shape tools/stream_bench.py run, profiles/*stream_bench*).  Workload: STUNet-B, 128^3, mask 0.6,
bf16 storage, per-GPU batch B: V = B * 128^3 voxels, 40 % active in the student, C0 = 32 stage-0 / decoder-output channels,
P = 53.05 M live parameters."""
import csv
import sys

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
V = B * 128 ** 3
A = 0.4
P4 = 53.05e6 * 4
rows = list(csv.DictReader(open(path)))
steps = [int(r["Calls"]) for r in rows if "adamw_ema_kernel" in r["Name"]][0]


def find(sub):
    return [r for r in rows if sub in r["Name"]]


# (kernel substring, what, bytes per launch averaged over the launches of a step, how the bytes are counted)
T = [
    ("stem_conv_mfma_kernel", "stem conv k3 Cin=1 (teacher and student, each on its 40 % visible patches)", (V * 4 + V * 32 * 2) * A, "x fp32 in + 32-ch bf16 out over the active 40 %"),
    ("stem_wgrad_mfma_kernel", "stem weight gradients (k3 and the k1 shortcut)", V * A * (32 * 2 + 4), "dy bf16 + x fp32 over the active 40 %"),
    ("proj_fwd_kernel", "decoder 1x1 projection 32 -> 1", V * 32 * 2 + V * 4, "x bf16 in + rec fp32 out"),
    ("proj_bwd_kernel", "its backward", V * 32 * 2 * 2 + V * 4, "x in + dx out (bf16) + drec fp32"),
    ("patch_loss_fwd_kernel", "patchify + per-patch MSE", V * 4 * 2, "inp + rec fp32"),
    ("patch_loss_bwd_kernel", "its backward", V * 4 * 3, "inp + rec in, drec out"),
    ("sumsq_kernel", "gradient norm", P4, "flat fp32 gradient"),
    ("adamw_ema_kernel", "clip + AdamW + EMA", P4 * 9, "p, g, m, v, ema read; p, m, v, ema written"),
    ("mask_sampler_kernel", "hard-mask sampler", 0, "B x 512 floats: latency only"),
]
print(f"# Per-kernel HBM fractions, from `{path}` ({steps} steps, B={B})\n")
print("| kernel | what | launches / step | avg launch | algorithmic bytes / launch | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for sub, what, nbytes, how in T:
    rs = find(sub)
    if not rs:
        continue
    calls = sum(int(r["Calls"]) for r in rs)
    tot = sum(float(r["TotalDurationNs"]) for r in rs)
    avg = tot / calls
    gbs = nbytes / avg if nbytes else 0.0
    print(f"| `{sub}` | {what} ({how}) | {calls / steps:.0f} | {avg / 1e3:.1f} us | {nbytes / 1e6:.0f} MB | {gbs:.0f} | {gbs / 8000:.1%} |" if nbytes else
          f"| `{sub}` | {what} ({how}) | {calls / steps:.0f} | {avg / 1e3:.1f} us | - | - | - |")
tot_all = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"\nAll kernels: {sum(int(r['Calls']) for r in rows) / steps:.0f} launches and {tot_all:.1f} ms of kernel time per step (side-stream kernels overlap the main stream).")
fam = {}
for r in rows:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
    fam[n] = fam.get(n, 0.0) + float(r["TotalDurationNs"]) / steps / 1e6
print("\n| kernel family | ms / step |\n|---|---|")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:16]:
    print(f"| `{k}` | {v:.2f} |")
