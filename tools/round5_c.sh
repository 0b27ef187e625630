#!/bin/bash
# round 5: conv_wgk3 (8-wave LDS-DMA weight gradient) -- op tests, same-box A/B of the launch (tools build: AM_WG_NOK3=1 = conv_wgrad.hip), step A/B
tag=${1:-r05_c}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "wgrad" > $out/pytest_wgrad.txt 2>&1; echo "pytest rc $?" >> $out/pytest_wgrad.txt
tail -12 $out/pytest_wgrad.txt
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1; tail -1 $out/build_ablate.txt
for rep in 1 2; do
  echo "== conv_wgk3 (rep $rep)"; timeout 300 python3 tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/wgrad_shapes_bench.py 16 2>&1 | grep -v amdgpu.ids
  echo "== conv_wgrad (rep $rep)"; AM_WG_NOK3=1 timeout 300 python3 tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/wgrad_shapes_bench.py 16 2>&1 | grep -v amdgpu.ids
done > $out/ab_wgk3_shapes_b16.txt 2>&1
cat $out/ab_wgk3_shapes_b16.txt
timeout 400 python3 tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/step_ab.py AM_WG_NOK3=0,1 16 > $out/ab_wgk3_step_b16.txt 2>&1; cat $out/ab_wgk3_step_b16.txt
