#!/bin/bash
tag=${1:-r05_k}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
timeout 500 python3 tools/with_lib.py $L tools/step_ab.py AM_WGK3_S8=21,42,84,10 16 > $out/ab_s8.txt 2>&1; cat $out/ab_s8.txt
AM_WG_NOK3=1 timeout 300 python3 tools/with_lib.py $L tools/step_run.py 16 8 1 2>&1 | grep -v amdgpu
