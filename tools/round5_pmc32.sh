#!/bin/bash
# counters of wgrad_k3_kernel<2> (64 -> 32 @128^3, B = 16) + a 200-step soak at batch 16
tag=${1:-r05_pmc32}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python3 $root/tools/wgk3_probe.py 64 32 128 16 6 > $out/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/wgk3_probe.py 64 32 128 16 6 > $out/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/wgk3_probe.py 64 32 128 16 6 > $out/write.log 2>&1
cd $root
python3 - <<PY
import csv, glob
def agg(d, names):
    f = glob.glob(f"$out/{d}/*/*_counter_collection.csv")[0]
    acc = {}; n = 0
    for r in csv.DictReader(open(f)):
        if "wgrad_k3_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc
sq = agg("sq", None); fe = agg("fetch", None); wr = agg("write", None)
nd = 10
print("counters summed over the dispatches:", {k: f"{v:.4g}" for k, v in sq.items()})
if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "SQ_BUSY_CYCLES" in sq:
    print("matrix pipe busy / SQ busy:", sq["SQ_VALU_MFMA_BUSY_CYCLES"] / sq["SQ_BUSY_CYCLES"] if sq["SQ_BUSY_CYCLES"] else None)
print("fetched KB (x2 per the gfx950 note) per dispatch:", 2 * fe.get("FETCH_SIZE", 0) / nd, " written KB per dispatch:", wr.get("WRITE_SIZE", 0) / nd)
PY
timeout 600 python3 tools/soak.py 200 16 2>&1 | grep -v amdgpu > $out/soak_b16.txt; tail -4 $out/soak_b16.txt
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
