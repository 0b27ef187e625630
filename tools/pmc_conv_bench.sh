#!/bin/bash
# HBM traffic of the two dominant kernels at the bench batch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes, kernel trace only
# beside them):  bash tools/pmc_conv_bench.sh r03_p   ->  gpurun_out/r03_p/pmc_conv_bench.md
tag=${1:-r03_p}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export AM_CB_BATCH=16
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/conv_bench.py all 10 > $out/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/conv_bench.py all 10 > $out/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/conv_bench.py all 10 > $out/trace.log 2>&1
cd $root
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
def mean(d, ctr, sub):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and sub in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
st = {r["Name"]: r for r in csv.DictReader(open(glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)[0]))}
algo = 2 * 16 * 128 ** 3 * 64 * 2
lines = ["# (both counters are in KB = 1024 B; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled: MI355X_MICROARCH.md \"HBM\")", "# HBM traffic of the two dominant kernels at the bench batch (conv 64->64 k3 @128^3, B=16, bf16; rocprofv3 --pmc, separate passes; tools/pmc_conv_bench.sh)", "",
         "| kernel | avg launch (`--kernel-trace --stats`) | FETCH_SIZE raw (KB) | fetched, corrected (x2) | WRITE_SIZE (KB) | algorithmic bytes | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
for sub in ("conv_igemm_kernel", "conv_wgrad_kernel"):
    f, n = mean("fetch", "FETCH_SIZE", sub); w, _ = mean("write", "WRITE_SIZE", sub)
    name = [k for k in st if sub in k][0]
    us = float(st[name]["AverageNs"]) / 1e3
    lines.append(f"| `{sub}` | {us:.0f} us ({2.0 * 16 * 128 ** 3 * 64 * 64 * 27 / us / 1e6:.0f} TFLOP/s) | {f:.0f} | {2 * f * 1024 / 1e6:.0f} MB | {w:.0f} = {w * 1024 / 1e6:.0f} MB | {algo / 1e6:.0f} MB | **{(2 * f + w) * 1024 / algo:.2f}** ({n} dispatches) |")
open(f"{out}/pmc_conv_bench.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $out -name "*.db" -delete
