#!/bin/bash
tag=${1:-r05_l}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv_k3 or k3" > $out/pytest_k3.txt 2>&1; echo "pytest rc $?" >> $out/pytest_k3.txt
tail -5 $out/pytest_k3.txt
timeout 600 python3 -m pytest tests/test_blocks_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider -x -k "blocks or config2_stunet_b_128_bf16" > $out/pytest_blocks.txt 2>&1; echo "pytest rc $?" >> $out/pytest_blocks.txt
tail -4 $out/pytest_blocks.txt
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
timeout 400 python3 tools/with_lib.py $L tools/step_ab.py AM_K3_NO32=0,1 16 > $out/ab_k3_cout32.txt 2>&1; cat $out/ab_k3_cout32.txt
