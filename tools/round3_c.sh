#!/bin/bash
# PMC traffic of the student sparse-encoder forward (FETCH_SIZE / WRITE_SIZE in separate passes, kernel trace only beside them) for
# STUNet-B 128^3 (headline) and STUNet-L 160^3 mask 0.7 (BASELINE configs[3]: "rocprof GB/s report"):  bash tools/round3_c.sh r03_v
tag=${1:-r03_v}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "B 128 0.6 16" "L 160 0.7 4"; do
  set -- $cfg
  export AM_ENC_SIZE=$1 AM_ENC_PATCH=$2 AM_ENC_MASK=$3
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch_$1 -- python3 $root/tools/encoder_profile.py $4 > $out/enc_fetch_$1.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/enc_write_$1 -- python3 $root/tools/encoder_profile.py $4 > $out/enc_write_$1.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_trace_$1 -- python3 $root/tools/encoder_profile.py $4 > $out/enc_trace_$1.log 2>&1
  grep "encoder forward" $out/enc_trace_$1.log
  python3 $root/tools/enc_traffic.py $out/enc_fetch_$1 $out/enc_write_$1 $4 $1 $2 $3 > $out/enc_traffic_$1.md 2>&1; tail -3 $out/enc_traffic_$1.md
done
find $out -name "*.db" -delete; find $out -name "*_kernel_trace.csv" -size +4M -delete
