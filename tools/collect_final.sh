#!/bin/bash
# tools/collect_final.sh TAG  -- copy what tools/round3_final.sh left under gpurun_out/TAG into profiles/TAG_* (the judged copies)
tag=${1:-r03_z}
src=gpurun_out/$tag
cd "$(dirname "$0")/.."
cp $src/bench.json profiles/${tag}_bench.json
cp $src/bench_stunet_L_160_m07_b4.json profiles/${tag}_bench_stunet_L_160_m07_b4.json
cp $src/bench_stunet_H_192_recompute_b2.json profiles/${tag}_bench_stunet_H_192_recompute_b2.json
grep -v amdgpu.ids $src/conv_census.txt > profiles/${tag}_conv_census.txt
grep -v amdgpu.ids $src/conv_census_stunet_L_160_m07_b4.txt > profiles/${tag}_conv_census_stunet_L_160_m07_b4.txt
grep -v amdgpu.ids $src/phase_times_b16.txt > profiles/${tag}_phase_times_b16.txt
{ grep -v amdgpu.ids $src/pytest.txt | tail -12; echo "--- smoke"; tail -2 $src/smoke.txt; } > profiles/${tag}_gpu_tests.txt
cp $(ls $src/step/*/*_kernel_stats.csv | head -1) profiles/${tag}_step_kernel_stats_b16.csv
cp $(ls $src/conv_b16/*/*_kernel_stats.csv | head -1) profiles/${tag}_conv_bench_b16_kernel_stats.csv
grep -v amdgpu.ids $src/conv_b16.log > profiles/${tag}_conv_bench_b16.txt
grep "ms/step" $src/step.log
ls -la profiles/${tag}_*
