#!/bin/bash
tag=${1:-r05_w}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_blocks_gpu.py tests/test_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "stem or config or recipe or block or step or fixture or train" > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; head -c 300 $out/bench.json; echo
timeout 300 python3 tools/phase_times.py 16 2>&1 | grep -v amdgpu
