#!/bin/bash
tag=${1:-r03_d}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1800 python3 -m pytest tests/test_configs_gpu.py tests/test_ops_gpu.py -m gpu -q -s -p no:cacheprovider -k "config or norm or batchnorm or densify" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
