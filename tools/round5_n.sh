#!/bin/bash
# conv_k3's transposed instantiation: op tests, launch A/B (tools build, AM_CV_NOK3T=1 = conv_igemm), step A/B
tag=${1:-r05_n}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "transpose or convT or conv_k3" > $out/pytest_ct.txt 2>&1; echo "pytest rc $?" >> $out/pytest_ct.txt
tail -12 $out/pytest_ct.txt
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
for rep in 1 2; do for no in 0 1; do echo "== AM_CV_NOK3T=$no (rep $rep)"; AM_CB_BATCH=16 AM_CV_NOK3T=$no timeout 300 python3 tools/with_lib.py $L tools/convt_bench.py 2>&1 | grep ConvT; done; done > $out/ab_convt_b16.txt 2>&1
cat $out/ab_convt_b16.txt
timeout 400 python3 tools/with_lib.py $L tools/step_ab.py AM_CV_NOK3T=0,1 16 > $out/ab_convt_step_b16.txt 2>&1; cat $out/ab_convt_step_b16.txt
