#!/bin/bash
tag=${1:-r05_t}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
for i in 1 2 3 4; do timeout 600 python3 -m pytest tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider -s -k "config2_stunet_b_128_bf16_step" 2>&1 | grep -E "passed|failed|worst:|ideal bf16|AssertionError: \(" | cut -c1-400; done
