#!/bin/bash
tag=${1:-r05_b5}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad or transpose or large_model" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
for tc in 1 0; do
echo "== AM_WG_TILECAP=$tc"
AM_WG_TILECAP=$tc timeout 300 python3 tools/with_lib.py $L tools/conv_census.py 4 2>&1 | grep "wgrad convT\|total"
AM_WG_TILECAP=$tc timeout 300 python3 tools/with_lib.py $L tools/conv_census.py 16 2>&1 | grep "wgrad convT\|total"
AM_WG_TILECAP=$tc timeout 600 python3 tools/with_lib.py $L bench.py --size L --patch 160 --mask-ratio 0.7 --batch 4 --steps 8 --warmup 3 --no-h2d 2>/dev/null | head -c 200; echo
AM_WG_TILECAP=$tc timeout 600 python3 tools/with_lib.py $L bench.py --size H --patch 192 --batch 2 --recompute --steps 5 --warmup 2 --no-h2d 2>/dev/null | head -c 200; echo
done
