#!/bin/bash
# in-step behaviour of conv_wgk3 vs conv_wgrad: side stream on / off, plain timing (tools build)
tag=${1:-r05_j}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
for rep in 1 2; do for nok in 0 1; do for side in 1 0; do
  AM_WG_NOK3=$nok timeout 200 python3 tools/with_lib.py $L tools/step_run.py 16 8 $side 2>&1 | grep -v amdgpu.ids | sed "s/^/nok3=$nok /"
done; done; done > $out/step_matrix.txt
cat $out/step_matrix.txt
cd /tmp && export TMPDIR=/tmp
for nok in 0 1; do
  AM_WG_NOK3=$nok timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_nok$nok -- python3 $root/tools/with_lib.py $root/$L $root/tools/step_run.py 16 6 1 > $out/tr_nok$nok.log 2>&1
  f=$(ls $out/tr_nok$nok/*/*kernel_stats.csv | head -1)
  echo "--- nok3=$nok"; head -12 $f | cut -c1-150
done
find $out -name "*_kernel_trace.csv" -size +2M -delete; find $out -name "*.db" -delete
