#!/bin/bash
tag=${1:-r03_g}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
AM_ABLATE="0,4096,8192,16384,32768" timeout 600 python3 tools/conv_ablate.py > $out/stagger.txt 2>&1; cat $out/stagger.txt
