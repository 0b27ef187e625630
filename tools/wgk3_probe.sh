#!/bin/bash
# timing ablations + SQ counters of conv_wgk3's kernel on the 64->64 @128^3 B=16 launch:  bash tools/wgk3_probe.sh r05_d
tag=${1:-r05_d}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1; tail -1 $out/build_ablate.txt
L=build_ab/libanatomask_hip_ablate.so
for d in ${WGK3_DBGS:-0 1 2 4 6 7}; do AM_WGK3_DBG=$d timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 2>&1 | grep -v amdgpu.ids; done > $out/wgk3_ablate.txt
for s8 in ${WGK3_S8S:-}; do AM_WGK3_S8=$s8 timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/s8=$s8 /"; done >> $out/wgk3_ablate.txt
AM_WG_NOK3=1 timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 2>&1 | grep -v amdgpu.ids >> $out/wgk3_ablate.txt
cat $out/wgk3_ablate.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python3 $root/tools/wgk3_probe.py 64 64 128 16 6 > $out/sq.log 2>&1
cd $root
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
fs = glob.glob(f"{out}/sq/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "wgrad_k3" in r["Kernel_Name"]]
names = sorted({r["Counter_Name"] for r in rows})
avg = {n: sum(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == n) / max(1, sum(1 for r in rows if r["Counter_Name"] == n)) for n in names}
tr = glob.glob(f"{out}/sq/**/*kernel_trace.csv", recursive=True)
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if "wgrad_k3" in r["Kernel_Name"]]
dur = sum(d) / len(d)
clk = avg["GRBM_GUI_ACTIVE"] / 8 / dur
print({k: round(v) for k, v in avg.items()})
print(f"launch {dur / 1e3:.0f} us, clock {clk:.2f} GHz, matrix pipe busy {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk * dur):.2f}, LDS array busy {avg['SQ_LDS_IDX_ACTIVE'] / (256 * clk * dur):.2f}, "
      f"LDS conflict cycles / LDS cycles {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']:.2f}")
PY
find $out -name "*.db" -delete; find $out -name "*_kernel_trace.csv" -size +2M -delete
