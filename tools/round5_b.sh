#!/bin/bash
# round 5: the eval-mode tail as one stencil -- op test, the tests that see the teacher pass, same-process A/B on the step
tag=${1:-r05_b}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -x -k "head_stencil or config2 or plain_spark or validation or trainer_n_steps or recipe_112 or full_size or nonfinite_step_is" > $out/pytest_new.txt 2>&1; echo "pytest rc $?" >> $out/pytest_new.txt
tail -8 $out/pytest_new.txt
timeout 300 python3 tools/step_ab.py engine.HEAD_STENCIL=1,0 16 > $out/ab_head_stencil_b16.txt 2>&1; cat $out/ab_head_stencil_b16.txt
timeout 300 python3 tools/phase_times.py 16 > $out/phase_times_b16.txt 2>&1; cat $out/phase_times_b16.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python3 $root/tools/step_run.py 16 8 1 > $out/step.log 2>&1
grep -h "head_stencil\|head_fold" $out/step/*/*kernel_stats.csv | cut -c1-200
find $out -name "*_kernel_trace.csv" -size +4M -delete; find $out -name "*.db" -delete
