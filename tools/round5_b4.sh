#!/bin/bash
tag=${1:-r05_b4}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
timeout 300 python3 tools/conv_census.py 4 2>&1 | grep -v amdgpu > $out/census_b4.txt; grep "k1s2\|step\|total" $out/census_b4.txt
timeout 300 python3 tools/conv_census.py 16 2>&1 | grep -v amdgpu > $out/census_b16.txt; grep "k1s2\|step\|total" $out/census_b16.txt
timeout 300 python3 tools/step_run.py 4 30 1 2>&1 | grep ms/step
timeout 300 python3 tools/step_run.py 16 20 1 2>&1 | grep ms/step
