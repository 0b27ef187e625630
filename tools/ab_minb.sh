# same-box A/B of the weight gradient's minimum bricks per slot (tools build, AM_WG_MINB): step time at batch 4 / 16, STUNet-L / H
cd ${GRAFT_REPO_ROOT:-$PWD}
L=build_ab/libanatomask_hip_ablate.so
for rep in 1 2; do for m in 16 32 64 128; do echo -n "MINB=$m B=4:  "; AM_WG_MINB=$m python tools/with_lib.py $L tools/step_run.py 4 30 1 2>&1 | grep ms/step; done; done
for m in 0 32 64 128; do echo -n "MINB=$m L: "; AM_WG_MINB=$m python tools/with_lib.py $L bench.py --size L --patch 160 --mask-ratio 0.7 --batch 4 --steps 6 --warmup 2 --no-h2d --no-roofline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(b['value'], b['ms_per_step'])"; done
for m in 0 32 64; do echo -n "MINB=$m H: "; AM_WG_MINB=$m python tools/with_lib.py $L bench.py --size H --patch 192 --batch 2 --recompute --steps 4 --warmup 2 --no-h2d --no-roofline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(b['value'], b['ms_per_step'])"; done
