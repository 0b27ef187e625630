#!/bin/bash
# first GPU pass of round 3: the whole GPU suite, the bench line, the phase table, isolated per-kernel times (side stream off)
tag=${1:-r03_a}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -5 $out/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1; cat $out/phase_times_b16.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_noside -- python3 $root/tools/step_run.py 16 10 0 > $out/step_noside.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_side -- python3 $root/tools/step_run.py 16 10 1 > $out/step_side.log 2>&1
cd $root
cat $out/step_noside.log $out/step_side.log | grep ms/step
find $out -name "*_kernel_trace.csv" -size +4M -delete
find $out -name "*.db" -delete
