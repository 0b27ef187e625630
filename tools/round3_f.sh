#!/bin/bash
tag=${1:-r03_f}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_configs_gpu.py tests/test_ops_gpu.py -m gpu -q -s -p no:cacheprovider -k "config or batchnorm" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
for b in 16 24 32; do timeout 300 python3 tools/step_run.py $b 10 1 2>&1 | grep ms/step; done | tee $out/batch_sweep.txt
