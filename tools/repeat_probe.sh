#!/bin/bash
# run the bench several times back to back, sampling clocks/power/temperatures while it runs
for r in 1 2 3 4; do
  (for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|junction|memory" | sed 's/=*//g' | tr '\n' ' '; echo; done) > gpurun_out/probe_$r.txt &
  P=$!
  python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
  wait $P
  grep -o "sclk[^)]*)\|mclk[^)]*)\|Power (W): [0-9.]*\|junction) (C): [0-9.]*\|memory) (C): [0-9.]*" gpurun_out/probe_$r.txt | tr '\n' ' ' | cut -c1-700; echo
done
