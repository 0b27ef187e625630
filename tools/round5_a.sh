#!/bin/bash
# round 5, first GPU call: the new tests first, then the whole GPU suite, smoke, bench, phase times (one box)
tag=${1:-r05_a}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -x -k "nonfinite or single_rank_nccl or sample_index_beyond or recipe_112 or f32_split or two_ranks" > $out/pytest_new.txt 2>&1; echo "pytest rc $?" >> $out/pytest_new.txt
tail -15 $out/pytest_new.txt
timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED\|rc " $out/pytest.txt | tail -12
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1; cat $out/phase_times_b16.txt
