#!/bin/bash
tag=${1:-r05_r}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1200 python3 -m pytest tests/test_ops_gpu.py tests/test_blocks_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider > $out/pytest_ops.txt 2>&1; echo "pytest rc $?" >> $out/pytest_ops.txt
tail -6 $out/pytest_ops.txt
timeout 400 python3 tools/k3_stress.py > $out/k3_stress.txt 2>&1; tail -7 $out/k3_stress.txt
AM_CB_BATCH=16 timeout 200 python3 tools/convt_bench.py 2>&1 | grep -v amdgpu
timeout 300 python3 tools/step_run.py 16 10 1 2>&1 | grep ms/step
timeout 300 python3 tools/step_run.py 16 10 1 2>&1 | grep ms/step
