#!/bin/bash
tag=${1:-r05_h96}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv_k3" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for v in 0 1 0 1; do
echo "== AM_K3_NO32=$v"
AM_K3_NO32=$v timeout 600 python3 tools/with_lib.py $L bench.py --size H --patch 192 --batch 2 --recompute --steps 5 --warmup 2 --no-h2d 2>/dev/null | head -c 200; echo
done
