#!/bin/bash
tag=${1:-r05_x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "stem" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_iso -- python3 $root/tools/step_run.py 16 20 0 > $out/step_iso.log 2>&1
cd $root
grep "ms/step" $out/step_iso.log
grep -h "stem\|conv_k3_kernel<4, false, true" $out/step_iso/*/*_kernel_stats.csv | cut -c1-60,150-260
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
