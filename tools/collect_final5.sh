#!/bin/bash
# tools/collect_final5.sh TAG  -- copy what tools/round5_final.sh left under gpurun_out/TAG into profiles/TAG_* (the judged copies)
tag=${1:-r05_final}
src=gpurun_out/$tag
cd "$(dirname "$0")/.."
cp $src/bench.json profiles/${tag}_bench.json
cp $src/bench_stunet_L_160_m07_b4.json profiles/${tag}_bench_stunet_L_160_m07_b4.json
cp $src/bench_stunet_H_192_recompute_b2.json profiles/${tag}_bench_stunet_H_192_recompute_b2.json
for f in conv_census phase_times_b16 conv_shapes_b16 wgrad_shapes_b16 k3_stress; do [ -f $src/$f.txt ] && grep -v amdgpu.ids $src/$f.txt > profiles/${tag}_$f.txt; done
{ grep -v amdgpu.ids $src/pytest.txt | tail -12; echo "--- smoke"; tail -2 $src/smoke.txt; } > profiles/${tag}_gpu_tests.txt
cp $(ls $src/step/*/*_kernel_stats.csv | head -1) profiles/${tag}_step_kernel_stats_b16.csv
cp $(ls $src/step_iso/*/*_kernel_stats.csv | head -1) profiles/${tag}_step_kernel_stats_b16_isolated.csv
cp $(ls $src/conv_b16/*/*_kernel_stats.csv | head -1) profiles/${tag}_conv_bench_b16_kernel_stats.csv
cp $(ls $src/enc_trace/*/*_kernel_stats.csv | head -1) profiles/${tag}_encoder_fwd_kernel_stats_b16.csv
grep -v amdgpu.ids $src/conv_b16.log > profiles/${tag}_conv_bench_b16.txt
cp $src/pmc_k3.md profiles/${tag}_pmc_k3.md; cp $src/pmc_k3.json profiles/r05_pmc_k3.json
cp $src/encoder_fwd_traffic.md profiles/${tag}_encoder_fwd_traffic.md; cp $src/encoder_fwd_traffic.json profiles/r05_encoder_fwd_traffic.json
grep "ms/step" $src/step.log $src/step_iso.log
ls -la profiles/${tag}_* profiles/r05_*.json
