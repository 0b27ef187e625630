#!/bin/bash
tag=${1:-r05_e}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "wgrad_k3_dense" > $out/pytest_wgrad.txt 2>&1; echo "pytest rc $?" >> $out/pytest_wgrad.txt
tail -5 $out/pytest_wgrad.txt
bash tools/wgk3_probe.sh $tag
