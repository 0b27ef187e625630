#!/bin/bash
tag=${1:-r05_o}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
for d in 0 1 2 8 9 10 11 3; do AM_K3_DBG=$d timeout 120 python3 tools/with_lib.py $L tools/ct_probe.py 2>&1 | grep ConvT; done > $out/ct_ablate.txt
for d in 0 1 8; do AM_K3_DBG=$d timeout 120 python3 tools/with_lib.py $L tools/ct_probe.py 128 32 16 2>&1 | grep ConvT; done >> $out/ct_ablate.txt
cat $out/ct_ablate.txt
