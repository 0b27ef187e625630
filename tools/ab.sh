#!/bin/bash
# tools/ab.sh "<python args>" lib1 lib2 ...  -- run the same tool with each build_ab/<lib>.so, interleaved twice (same GPU box)
cmd=$1; shift
for rep in 1 2; do for l in "$@"; do echo "== $l (rep $rep)"; timeout 600 python tools/with_lib.py $PWD/build_ab/$l.so $cmd 2>&1 | grep -v amdgpu.ids; done; done
