#!/bin/bash
tag=${1:-r05_b6}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 -m anatomask_amd.build --ablate > $out/build_ablate.txt 2>&1
L=build_ab/libanatomask_hip_ablate.so
timeout 900 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad or large_model" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
for tc in 1 0; do
echo "== AM_WG_FILL=$tc"
AM_WG_FILL=$tc timeout 300 python3 tools/with_lib.py $L tools/conv_census.py 4 2>&1 | grep -v amdgpu > $out/census_b4_$tc.txt; grep "total" $out/census_b4_$tc.txt
AM_WG_FILL=$tc timeout 300 python3 tools/with_lib.py $L tools/conv_census.py 16 2>&1 | grep "total"
AM_WG_FILL=$tc timeout 300 python3 tools/with_lib.py $L tools/step_run.py 4 30 1 2>&1 | grep ms/step
AM_WG_FILL=$tc timeout 300 python3 tools/with_lib.py $L tools/step_run.py 4 30 1 2>&1 | grep ms/step
done
python3 - <<'PY'
import re,sys
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$",l)
        if m: d[m.group(1).strip()]=float(m.group(3))
    return d
a=load(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r05_b6/census_b4_1.txt"); b=load("gpurun_out/r05_b6/census_b4_0.txt")
for k in a:
    if k in b and abs(a[k]-b[k])>0.015: print(f"{k:45s} fill {a[k]:.3f}  old {b[k]:.3f}")
PY
