#!/bin/bash
tag=${1:-r05_p}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_blocks_gpu.py -m gpu -q -p no:cacheprovider -k "transpose or convT or conv_k3 or k3 or blocks" > $out/pytest_k3.txt 2>&1; echo "pytest rc $?" >> $out/pytest_k3.txt
tail -6 $out/pytest_k3.txt; for i in 1 2 3; do timeout 300 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "transpose_persistent" 2>&1 | tail -1; done
timeout 300 python3 tools/k3_stress.py > $out/k3_stress.txt 2>&1; tail -3 $out/k3_stress.txt
AM_CB_BATCH=16 AM_CB_STATS=1 timeout 200 python3 tools/conv_bench.py fwd 20 2>&1 | grep TFLOP
timeout 200 python3 tools/conv_shapes_bench.py 16 2>&1 | grep -v amdgpu
AM_CB_BATCH=16 timeout 200 python3 tools/convt_bench.py 2>&1 | grep ConvT
timeout 300 python3 tools/step_run.py 16 10 1 2>&1 | grep ms/step
