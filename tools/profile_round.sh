#!/bin/bash
# tools/profile_round.sh TAG  -- everything the round's profiles/ are made of, in ONE gpurun call (same box for all of it):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r02_b'     -> gpurun_out/TAG/*  (copy what is judged into profiles/)
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python3 tools/stream_bench.py > $out/stream_bench.txt 2>&1
python3 tools/layer_bench.py > $out/layer_bench.txt 2>&1
python3 tools/conv_census.py 16 > $out/conv_census.txt 2>&1
python3 tools/small_steps.py > $out/small_steps.txt 2>&1
AM_CB_BATCH=16 python3 tools/conv_bench.py all 20 > $out/conv_bench_b16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python3 $root/bench.py --steps 20 --warmup 5 > $out/bench_under_rocprof.json 2> $out/step.err
AM_CB_BATCH=16 rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv_b16 -- python3 $root/tools/conv_bench.py all 20 > $out/conv_b16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv_b2 -- python3 $root/tools/conv_bench.py all 10 > $out/conv_b2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/tools/conv_bench.py all 3 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/tools/conv_bench.py all 3 > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stream -- python3 $root/tools/stream_bench.py > $out/stream.log 2>&1
cd $root
# keep only the small summaries (kernel_stats + counter collection), drop the per-dispatch traces
find $out -name "*_kernel_trace.csv" -size +4M -delete
find $out -name "*.db" -delete
ls -R $out | head -60
