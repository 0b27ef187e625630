#!/bin/bash
# tools/ab.sh "<python args>" lib1 lib2 ...  -- run the same tool with each build_ab/<lib>.so, interleaved twice (same GPU box)
cmd=$1; shift
for rep in 1 2; do for l in "$@"; do echo "== $l (rep $rep)"; AM_HIP_LIB=$PWD/build_ab/$l.so timeout 600 python $cmd 2>&1 | grep -v amdgpu.ids; done; done
