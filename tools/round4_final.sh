#!/bin/bash
# everything profiles/r04_* is made of, ONE gpurun call (same box): bash tools/round4_final.sh r04_z
tag=${1:-r04_z}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 1200 $out/bench.json
timeout 900 python3 bench.py --size L --patch 160 --mask-ratio 0.7 --batch 4 --steps 8 --warmup 3 --no-h2d > $out/bench_stunet_L_160_m07_b4.json 2> $out/bench_L.err; tail -c 700 $out/bench_stunet_L_160_m07_b4.json
timeout 900 python3 bench.py --size H --patch 192 --batch 2 --recompute --steps 5 --warmup 2 --no-h2d > $out/bench_stunet_H_192_recompute_b2.json 2> $out/bench_H.err; tail -c 700 $out/bench_stunet_H_192_recompute_b2.json
timeout 600 python3 tools/conv_census.py 16 > $out/conv_census.txt 2>&1; tail -1 $out/conv_census.txt
AM_CENSUS_SIZE=L AM_CENSUS_PATCH=160 AM_CENSUS_MASK=0.7 timeout 600 python3 tools/conv_census.py 4 > $out/conv_census_stunet_L_160_m07_b4.txt 2>&1; tail -1 $out/conv_census_stunet_L_160_m07_b4.txt
AM_CENSUS_SIZE=H AM_CENSUS_PATCH=192 AM_CENSUS_RECOMPUTE=1 timeout 600 python3 tools/conv_census.py 2 > $out/conv_census_stunet_H_192_recompute_b2.txt 2>&1; tail -1 $out/conv_census_stunet_H_192_recompute_b2.txt
TARGETS=512,1024,2048,4096 timeout 300 python3 tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/norm_bench.py 16 > $out/norm_bench_b16.txt 2>&1
timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1
timeout 600 python3 tools/batch_cliff.py 16 24 32 > $out/batch_cliff.txt 2>&1
timeout 300 python3 tools/conv_shapes_bench.py 16 > $out/conv_shapes_b16.txt 2>&1
timeout 300 python3 tools/with_lib.py build_ab/libanatomask_hip_ablate.so tools/k3_ablate.py 16 > $out/k3_ablate_b16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python3 $root/tools/step_run.py 16 20 1 > $out/step.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_iso -- python3 $root/tools/step_run.py 16 20 0 > $out/step_iso.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f32s -- python3 $root/tools/step_run_f32.py 1 6 > $out/f32s.log 2>&1
AM_CB_BATCH=16 AM_CB_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv_b16 -- python3 $root/tools/conv_bench.py all 20 > $out/conv_b16.log 2>&1
cd $root
bash tools/pmc_k3.sh $tag > $out/pmc_k3.log 2>&1
cat $out/step.log $out/step_iso.log $out/f32s.log | grep ms/step; cat $out/conv_b16.log | grep TFLOP
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
