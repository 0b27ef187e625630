#!/bin/bash
# Counters of the dominant kernels at the bench batch (conv 64->64 k3 @128^3, B=16, bf16):  bash tools/pmc_k3.sh r04_x
#   pass 1: rocprofv3 --kernel-trace --pmc FETCH_SIZE          pass 2: --pmc WRITE_SIZE        (TCC counters do not share a pass)
#   pass 3: SQ counters (LDS conflicts, matrix-pipe busy, waits) pass 4: --kernel-trace --stats (durations without counters)
# writes gpurun_out/<tag>/pmc_k3.md and pmc_k3.json (the json is what bench.py reads for `roofline.traffic`: copy it to profiles/).
tag=${1:-r04_x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export AM_CB_BATCH=16 AM_CB_STATS=1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/conv_bench.py all 10 > $out/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/conv_bench.py all 10 > $out/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python3 $root/tools/conv_bench.py all 10 > $out/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/conv_bench.py all 10 > $out/trace.log 2>&1
cd $root
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def ctr(d, name, sub):
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None, 0
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == name and sub in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (None, 0)
st = {r["Name"]: r for r in csv.DictReader(open(glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)[0]))}
B, S, C = 16, 128, 64
algo = 2 * B * S ** 3 * C * 2
flop = 2.0 * B * S ** 3 * C * C * 27
lines = ["# Counters of the dominant kernels at the bench batch (conv 64->64 k3 @128^3, B=16, bf16, statistics epilogue ON; rocprofv3 --pmc, separate passes; tools/pmc_k3.sh)",
         "# (FETCH_SIZE / WRITE_SIZE are in KB = 1024 B; FETCH_SIZE counts half of the bytes of wide coalesced reads on gfx950 and is doubled: MI355X_MICROARCH.md \"HBM\")", "",
         "| kernel | avg launch (`--kernel-trace --stats`) | fetched (2 x FETCH_SIZE) | written | traffic / algorithmic | matrix pipe busy | LDS array busy | LDS conflict cycles / LDS cycles | clock |", "|---|---|---|---|---|---|---|---|---|"]
js = {}
for sub in ("conv_k3_kernel", "conv_igemm_kernel", "conv_wgrad_kernel", "wgrad_k3_kernel"):
    names = [k for k in st if sub in k]
    if not names:
        continue
    name = max(names, key=lambda k: float(st[k]["TotalDurationNs"]))
    us = float(st[name]["AverageNs"]) / 1e3
    f, n = ctr("fetch", "FETCH_SIZE", sub); w, _ = ctr("write", "WRITE_SIZE", sub)
    mf, _ = ctr("sq", "SQ_VALU_MFMA_BUSY_CYCLES", sub); bc, _ = ctr("sq", "SQ_LDS_BANK_CONFLICT", sub); la, _ = ctr("sq", "SQ_LDS_IDX_ACTIVE", sub)
    gui, _ = ctr("sq", "GRBM_GUI_ACTIVE", sub)
    sqfs = glob.glob(f"{out}/sq/**/*kernel_trace.csv", recursive=True)
    dur_sq = None
    if sqfs:
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(sqfs[0])) if sub in r["Kernel_Name"]]
        dur_sq = sum(d) / len(d) if d else None
    clk = gui / 8 / dur_sq if (gui and dur_sq) else None          # GHz (cycles per ns), summed over the 8 XCDs
    busy = mf / (1024 * clk * dur_sq) if (mf and clk) else None
    lds = la / (256 * clk * dur_sq) if (la and clk) else None
    fb, wb = (2 * f * 1024 if f else None), (w * 1024 if w else None)
    lines.append(f"| `{name[:60]}` | {us:.0f} us ({flop / us / 1e6:.0f} TFLOP/s, {flop / us / 1e6 / 2500:.3f} of 2.5 PF) | {fb / 1e6:.0f} MB | {wb / 1e6:.0f} MB | **{(fb + wb) / algo:.2f}** ({n} dispatches) | "
                 f"{busy:.2f} | {lds:.2f} | {bc / la:.2f} | {clk:.2f} GHz |" if (fb and wb and busy and lds) else f"| `{name[:60]}` | {us:.0f} us | counters missing |")
    js[sub] = {"kernel": name, "launch_us": us, "tflops": flop / us / 1e6, "fetch_bytes": fb, "write_bytes": wb, "algorithmic_bytes": algo, "batch": B,
               "mfma_busy": busy, "lds_busy": lds, "lds_conflict_frac": (bc / la) if (bc is not None and la) else None, "clock_ghz": clk}
open(f"{out}/pmc_k3.md", "w").write("\n".join(lines) + "\n")
json.dump(js, open(f"{out}/pmc_k3.json", "w"), indent=1)
print("\n".join(lines))
PY
find $out -name "*.db" -delete
