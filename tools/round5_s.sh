#!/bin/bash
tag=${1:-r05_s}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 600 python3 tools/conv_census.py 16 > $out/conv_census.txt 2>&1; grep -v amdgpu $out/conv_census.txt | head -60
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_iso -- python3 $root/tools/step_run.py 16 10 0 > $out/step_iso.log 2>&1
cd $root
python3 - $out <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+"/step_iso/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
calls_adam=[int(r["Calls"]) for r in rows if "adamw_ema" in r["Name"]][0]
print("steps", calls_adam, "sum per step ms", tot/calls_adam/1e6)
for r in rows[:32]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"])[:80]
    print(f"{float(r['TotalDurationNs'])/calls_adam/1e6:7.2f} ms/step  {int(r['Calls'])/calls_adam:5.1f} calls  {n}")
PY
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
