#!/bin/bash
tag=${1:-r05_gap}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for B in 4 16; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace_b$B -- python3 $root/tools/step_run.py $B 14 1 > $out/step_b$B.log 2>&1
grep ms/step $out/step_b$B.log
python3 $root/tools/gap_analysis.py $(ls $out/trace_b$B/*/*_kernel_trace.csv | head -1) 10
done
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*.db" -delete
