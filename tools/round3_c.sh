#!/bin/bash
tag=${1:-r03_c}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1800 python3 -m pytest tests/test_configs_gpu.py tests/test_ops_gpu.py -m gpu -q -s -p no:cacheprovider -k "config or head" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
timeout 600 python3 tools/small_steps.py > $out/small_steps.txt 2>&1; cat $out/small_steps.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_b4 -- python3 $root/tools/step_run.py 4 20 1 > $out/step_b4.log 2>&1
cd $root; cat $out/step_b4.log | grep ms/step
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
