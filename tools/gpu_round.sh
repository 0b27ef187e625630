#!/bin/bash
# ONE parametrised GPU-box script (replaces the per-experiment round*_X.sh files):  bash tools/gpu_round.sh TAG STAGE [STAGE ...]
# Every stage writes under gpurun_out/TAG/.  Stages:
#   tests [PYTEST_ARGS via AM_PYTEST]   pytest -m gpu (default: whole suite)      smoke      __graft_entry__.smoke()
#   bench        default bench line                  benchLH    STUNet-L 160^3 / STUNet-H 192^3 lines
#   census       tools/conv_census.py 16             shapes     conv_shapes_bench + wgrad_shapes_bench
#   phases       tools/phase_times.py 16             step       rocprofv3 kernel stats of the step (side stream on / off)
#   convb        rocprofv3 kernel stats of tools/conv_bench.py (the dominant launch)
#   enc          encoder forward: kernel stats + FETCH_SIZE / WRITE_SIZE passes        pmc    tools/pmc_k3.sh (dominant kernel counters)
#   norm         tools/norm_bench.py 16              stress     tools/k3_stress.py        cmd    runs "$AM_CMD" (a one-off probe)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
prof() { (cd /tmp && export TMPDIR=/tmp && timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d "$@"); }
for stage in "$@"; do
  case $stage in
    tests) timeout ${AM_PYTEST_TIMEOUT:-3000} python3 -m pytest ${AM_PYTEST:-tests} -m gpu -q -p no:cacheprovider -x ${AM_PYTEST_FLAGS:--s} > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
           grep -v amdgpu.ids $out/pytest.txt | grep -n "passed\|failed\|FAILED\|Error\|rc " | tail -8 ;;
    smoke) timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt ;;
    bench) timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; head -c 400 $out/bench.json; echo ;;
    benchq) timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-h2d > $out/bench_quick.json 2> $out/bench_quick.err; head -c 300 $out/bench_quick.json; echo ;;
    benchLH) timeout 900 python3 bench.py --size L --patch 160 --mask-ratio 0.7 --batch 4 --steps 8 --warmup 3 --no-h2d > $out/bench_stunet_L_160_m07_b4.json 2> $out/bench_L.err; head -c 300 $out/bench_stunet_L_160_m07_b4.json; echo
             timeout 900 python3 bench.py --size H --patch 192 --batch 2 --recompute --steps 5 --warmup 2 --no-h2d > $out/bench_stunet_H_192_recompute_b2.json 2> $out/bench_H.err; head -c 300 $out/bench_stunet_H_192_recompute_b2.json; echo ;;
    census) timeout 600 python3 tools/conv_census.py 16 > $out/conv_census.txt 2>&1; tail -1 $out/conv_census.txt ;;
    shapes) timeout 300 python3 tools/conv_shapes_bench.py 16 > $out/conv_shapes_b16.txt 2>&1; timeout 300 python3 tools/wgrad_shapes_bench.py 16 > $out/wgrad_shapes_b16.txt 2>&1 ;;
    phases) timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1; cat $out/phase_times_b16.txt | grep -v amdgpu.ids | tail -4 ;;
    step) prof $out/step -- python3 $root/tools/step_run.py 16 20 1 > $out/step.log 2>&1; prof $out/step_iso -- python3 $root/tools/step_run.py 16 20 0 > $out/step_iso.log 2>&1
          cat $out/step.log $out/step_iso.log | grep ms/step ;;
    convb) AM_CB_BATCH=16 AM_CB_STATS=1 prof $out/conv_b16 -- python3 $root/tools/conv_bench.py all 20 > $out/conv_b16.log 2>&1; grep TFLOP $out/conv_b16.log ;;
    enc) (cd /tmp && export TMPDIR=/tmp
          timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch -- python3 $root/tools/encoder_profile.py 16 > $out/enc_fetch.log 2>&1
          timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/enc_write -- python3 $root/tools/encoder_profile.py 16 > $out/enc_write.log 2>&1)
         prof $out/enc_trace -- python3 $root/tools/encoder_profile.py 16 > $out/enc_trace.log 2>&1
         python3 tools/enc_traffic.py $out/enc_fetch $out/enc_write 16 B 128 0.6 $out/encoder_fwd_traffic.json > $out/encoder_fwd_traffic.md 2>&1; tail -3 $out/encoder_fwd_traffic.md ;;
    pmc) bash tools/pmc_k3.sh $tag > $out/pmc_k3.log 2>&1; tail -3 $out/pmc_k3.log ;;
    norm) timeout 300 python3 tools/norm_bench.py 16 > $out/norm_bench_b16.txt 2>&1; tail -5 $out/norm_bench_b16.txt ;;
    stress) timeout 900 python3 tools/k3_stress.py 24 2>&1 | grep -v amdgpu.ids > $out/k3_stress.txt; tail -1 $out/k3_stress.txt ;;
    cmd) bash -c "$AM_CMD" > $out/cmd.txt 2>&1; tail -${AM_CMD_TAIL:-30} $out/cmd.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
