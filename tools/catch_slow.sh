#!/bin/bash
# repeat the bench until a run is slow, then look at the machine (is it the GPU, the memory, the host?)
for r in 1 2 3 4 5 6 7 8; do
  v=$(python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "run $r: $v vol/s"
  if python -c "import sys; sys.exit(0 if float('$v') < 70 else 1)"; then
    echo "--- slow state: diagnostics"
    python tools/conv_bench.py all 10 2>&1 | grep TFLOP
    python tools/stream_bench.py 2>&1 | grep -E "copy|norm_apply \("
    rocm-smi --showmemuse --showpower --showclocks 2>/dev/null | grep -E "VRAM|Power|sclk|mclk"
    rocm-smi --showpids 2>/dev/null | tail -8
    python - <<'PY'
import time, torch
t0=time.time(); x=torch.empty(1<<20); 
for _ in range(200): x.add_(1)
print("host loop s", time.time()-t0)
PY
    uptime
    break
  fi
done
