#!/bin/bash
# tools/collect_round.sh TAG [ROUND]  -- copy what tools/gpu_round.sh left under gpurun_out/TAG into profiles/TAG_* (the judged copies);
# ROUND (default r06) names the two counter files bench.py reads: profiles/ROUND_pmc_k3.json, profiles/ROUND_encoder_fwd_traffic.json
tag=${1:-r06_final}; rnd=${2:-r06}
src=gpurun_out/$tag
cd "$(dirname "$0")/.."
cpif() { [ -f "$1" ] && cp "$1" "$2"; }
cpif $src/bench.json profiles/${tag}_bench.json
cpif $src/bench_stunet_L_160_m07_b4.json profiles/${tag}_bench_stunet_L_160_m07_b4.json
cpif $src/bench_stunet_H_192_recompute_b2.json profiles/${tag}_bench_stunet_H_192_recompute_b2.json
for f in conv_census phase_times_b16 conv_shapes_b16 wgrad_shapes_b16 k3_stress norm_bench_b16; do [ -f $src/$f.txt ] && grep -v amdgpu.ids $src/$f.txt > profiles/${tag}_$f.txt; done
[ -f $src/pytest.txt ] && { grep -v amdgpu.ids $src/pytest.txt | tail -12; echo "--- smoke"; tail -2 $src/smoke.txt 2>/dev/null; } > profiles/${tag}_gpu_tests.txt
first() { ls $1 2>/dev/null | head -1; }
for pair in step:step_kernel_stats_b16 step_iso:step_kernel_stats_b16_isolated conv_b16:conv_bench_b16_kernel_stats enc_trace:encoder_fwd_kernel_stats_b16; do
  d=${pair%%:*}; n=${pair##*:}; f=$(first "$src/$d/*/*_kernel_stats.csv"); [ -n "$f" ] && cp $f profiles/${tag}_$n.csv
done
[ -f $src/conv_b16.log ] && grep -v amdgpu.ids $src/conv_b16.log > profiles/${tag}_conv_bench_b16.txt
cpif $src/pmc_k3.md profiles/${tag}_pmc_k3.md; cpif $src/pmc_k3.json profiles/${rnd}_pmc_k3.json
cpif $src/encoder_fwd_traffic.md profiles/${tag}_encoder_fwd_traffic.md; cpif $src/encoder_fwd_traffic.json profiles/${rnd}_encoder_fwd_traffic.json
grep -h "ms/step" $src/step.log $src/step_iso.log 2>/dev/null
ls profiles/${tag}_* profiles/${rnd}_*.json 2>/dev/null
