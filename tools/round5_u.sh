#!/bin/bash
tag=${1:-r05_u}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
run() { t=$1; shift; env "$@" timeout 600 python3 tools/with_lib.py $L tests/probe_bf16_order_noise.py $t 2>&1 | grep "^\[" | cut -c1-700; }
run new AM_NONE=1
run new2 AM_NONE=1
run oldct AM_CV_NOK3T=1
run old32 AM_K3_NO32=1
run oldwg AM_WG_NOK3=1
run oldall AM_CV_NOK3T=1 AM_K3_NO32=1 AM_WG_NOK3=1
run nok3 AM_CV_NOK3=1 AM_CV_NOK3T=1 AM_WG_NOK3=1
python3 tests/probe_bf16_order_noise.py cmp new new2 oldct old32 oldwg oldall nok3
