#!/bin/bash
# everything profiles/r05_* is made of, ONE gpurun call (same box): bash tools/round5_final.sh r05_z
tag=${1:-r05_final}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
# ---- counters first: bench.py reads the two json files from profiles/ (a missing file is an error)
bash tools/pmc_k3.sh $tag > $out/pmc_k3.log 2>&1; cp $out/pmc_k3.json profiles/r05_pmc_k3.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch -- python3 $root/tools/encoder_profile.py 16 > $out/enc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/enc_write -- python3 $root/tools/encoder_profile.py 16 > $out/enc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_trace -- python3 $root/tools/encoder_profile.py 16 > $out/enc_trace.log 2>&1
cd $root
python3 tools/enc_traffic.py $out/enc_fetch $out/enc_write 16 B 128 0.6 $out/encoder_fwd_traffic.json > $out/encoder_fwd_traffic.md 2>&1; tail -3 $out/encoder_fwd_traffic.md
cp $out/encoder_fwd_traffic.json profiles/r05_encoder_fwd_traffic.json
# ---- tests, smoke, bench
timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
timeout 900 python3 tools/k3_stress.py 24 2>&1 | grep -v amdgpu.ids > $out/k3_stress.txt; tail -1 $out/k3_stress.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; head -c 400 $out/bench.json; echo
timeout 900 python3 bench.py --size L --patch 160 --mask-ratio 0.7 --batch 4 --steps 8 --warmup 3 --no-h2d > $out/bench_stunet_L_160_m07_b4.json 2> $out/bench_L.err; head -c 300 $out/bench_stunet_L_160_m07_b4.json; echo
timeout 900 python3 bench.py --size H --patch 192 --batch 2 --recompute --steps 5 --warmup 2 --no-h2d > $out/bench_stunet_H_192_recompute_b2.json 2> $out/bench_H.err; head -c 300 $out/bench_stunet_H_192_recompute_b2.json; echo
timeout 600 python3 tools/conv_census.py 16 > $out/conv_census.txt 2>&1; tail -1 $out/conv_census.txt
timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1
timeout 300 python3 tools/conv_shapes_bench.py 16 > $out/conv_shapes_b16.txt 2>&1
timeout 300 python3 tools/wgrad_shapes_bench.py 16 > $out/wgrad_shapes_b16.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python3 $root/tools/step_run.py 16 20 1 > $out/step.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_iso -- python3 $root/tools/step_run.py 16 20 0 > $out/step_iso.log 2>&1
AM_CB_BATCH=16 AM_CB_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv_b16 -- python3 $root/tools/conv_bench.py all 20 > $out/conv_b16.log 2>&1
cd $root
cat $out/step.log $out/step_iso.log | grep ms/step; cat $out/conv_b16.log | grep TFLOP
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
