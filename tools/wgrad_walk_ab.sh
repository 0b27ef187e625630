#!/bin/bash
# tools/wgrad_walk_ab.sh -- brick-walk order of conv_wgrad_kernel: time and L2->fabric fetch bytes, walk 0 (w fastest, contiguous runs)
# against walk 1 (d fastest, columns interleaved over the slots of an XCD), tools build (env switches), one box.
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/walk
mkdir -p $out
cd $root
alt=$root/build_ab/libanatomask_hip_ablate.so    # bound explicitly through tools/with_lib.py (the product loader reads no environment)
for b in 2 16; do
  for w in 0 1; do for pl in 0 1; do
    echo "== B=$b walk=$w plane=$pl"; AM_CB_BATCH=$b AM_WG_WALK=$w AM_WG_PLANE=$pl python3 tools/with_lib.py $alt tools/conv_bench.py wgrad 20
  done; done
  for sg in 8 16 32 128; do echo "== B=$b walk=1 plane=1 seg=$sg"; AM_CB_BATCH=$b AM_WG_SEG=$sg python3 tools/with_lib.py $alt tools/conv_bench.py wgrad 20; done
done 2>&1 | grep -v amdgpu.ids | tee $out/times.txt
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do for pl in 0 1; do
  AM_WG_WALK=$w AM_WG_PLANE=$pl rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_w${w}p$pl -- python3 $root/tools/with_lib.py $alt $root/tools/conv_bench.py wgrad 3 > $out/fetch_w${w}p$pl.log 2>&1
done; done
AM_WG_SEG=128 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_w1p1s128 -- python3 $root/tools/with_lib.py $alt $root/tools/conv_bench.py wgrad 3 > $out/fetch_w1p1s128.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
for d in ("fetch_w0p0", "fetch_w0p1", "fetch_w1p0", "fetch_w1p1", "fetch_w1p1s128"):
    for f in glob.glob(f"gpurun_out/walk/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "wgrad_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "n=%d mean FETCH_SIZE %.0f KB -> x2 = %.1f MB" % (len(v), sum(v) / len(v), 2 * sum(v) / len(v) / 1e3))
PY
find $out -name "*.db" -delete; find $out -name "*_kernel_trace.csv" -size +1M -delete
