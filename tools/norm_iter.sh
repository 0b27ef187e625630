#!/bin/bash
# quick loop for the norm kernels: the norm / e2e parity tests, the isolated step under rocprofv3 (kernel stats), the bench line
tag=${1:-r04_n}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1200 python3 -m pytest tests/test_ops_gpu.py tests/test_blocks_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -p no:cacheprovider -k "norm or block or reference or determin or recompute" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_iso -- python3 $root/tools/step_run.py 16 20 0 > $out/step_iso.log 2>&1
cd $root
grep ms/step $out/step_iso.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-h2d > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
