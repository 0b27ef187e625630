# same-box A/B of the weight gradient's staging variants (tools build): register-staged (product) / LDS-DMA 1x8x16 bricks / LDS-DMA 1x4x16 bricks
cd ${GRAFT_REPO_ROOT:-$PWD}
L=build_ab/libanatomask_hip_ablate.so
for v in "AM_WG_DMA=0" "AM_WG_DMA=1" "AM_WG_DMA_BH4=1"; do echo "$v"; env $v AM_CB_BATCH=16 python tools/with_lib.py $L tools/conv_bench.py all 20 2>&1 | grep -i wgrad; done
for rep in 1 2; do for v in "AM_WG_DMA=0" "AM_WG_DMA=1" "AM_WG_DMA_BH4=1"; do echo -n "$v "; env $v python tools/with_lib.py $L tools/step_run.py 16 20 1 2>&1 | grep ms/step; done; done
