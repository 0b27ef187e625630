#!/bin/bash
# the GPU suite + smoke + race screen + the default bench line, nothing else:  bash tools/round5_tests_only.sh TAG
tag=${1:-r05_final2}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
timeout 900 python3 tools/k3_stress.py 24 2>&1 | grep -v amdgpu.ids > $out/k3_stress.txt; tail -1 $out/k3_stress.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; head -c 330 $out/bench.json; echo
