#!/bin/bash
tag=${1:-r05_m}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -x -k "prenorm or config2_stunet_b_128_bf16 or standalone_encoder or sparse_encoder_forward or trainer_n_steps_bf16 or reference_style or forward_matches" > $out/pytest_new.txt 2>&1; echo "pytest rc $?" >> $out/pytest_new.txt
tail -5 $out/pytest_new.txt
timeout 300 python3 tools/step_ab.py engine.FUSED_PRENORM=1,0 16 > $out/ab_prenorm_b16.txt 2>&1; cat $out/ab_prenorm_b16.txt
timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-h2d > $out/bench_quick.json 2> $out/bench_quick.err; python3 - $out/bench_quick.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(r["value"], r["ms_per_step"], r.get("encoder_fwd_hbm"), r["roofline"]["frac"], r["roofline"]["launch_ms"])
PY
