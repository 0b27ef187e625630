#!/bin/bash
# GPU pass b of round 3: GPU suite, fused-head A/B, PMC traffic of the encoder forward
tag=${1:-r03_b}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1800 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
grep -n "passed\|failed\|FAILED" $out/pytest.txt | tail -8
timeout 400 python3 tools/step_ab.py engine.FUSED_HEAD=1,0 16 > $out/ab_fused_head.txt 2>&1; tail -3 $out/ab_fused_head.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch -- python3 $root/tools/encoder_profile.py 16 > $out/enc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/enc_write -- python3 $root/tools/encoder_profile.py 16 > $out/enc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_trace -- python3 $root/tools/encoder_profile.py 16 > $out/enc_trace.log 2>&1
cd $root
cat $out/enc_trace.log | tail -2
find $out -name "*.db" -delete
ls -la $out/enc_fetch/*/ | head
