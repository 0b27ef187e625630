#!/bin/bash
# 32-wide cy tiles of conv_wgk3: op tests, same-box A/B of the launch (tools build), ablations, step A/B
tag=${1:-r05_y2}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
for rep in 1 2; do
for d in 0 1 2 4 6 16; do AM_WGK3_DBG=$d timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 64 32 128 16 2>&1 | grep "wgrad k3"; done
AM_WG_NOK3=1 timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 64 32 128 16 2>&1 | grep "wgrad k3"
done
timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 192 96 96 2 2>&1 | grep "wgrad k3"
AM_WG_NOK3=1 timeout 120 python3 tools/with_lib.py $L tools/wgk3_probe.py 192 96 96 2 2>&1 | grep "wgrad k3"
timeout 400 python3 tools/with_lib.py $L tools/step_ab.py AM_WG_NOK3=0,1 16 2>&1 | grep -v amdgpu
