#!/bin/bash
# sample sclk / power while the conv micro-benchmark runs
(for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/clock_probe.txt &
P=$!
AM_ABLATE=0,0,0,0,0,0 python tools/conv_ablate.py 2>&1 | grep "64->64 @128" | cut -c1-120
wait $P
cat gpurun_out/clock_probe.txt
