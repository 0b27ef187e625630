#!/bin/bash
# tools/ab_variant.sh NAME "<extra hipcc flags>"  -- on the GPU box: build build_ab/base.so (the tree as it is) and build_ab/NAME.so (the tree + flags),
# check NAME against F.conv3d (tools/k3_check.py quick) and time both, interleaved: the dominant launch (with / without statistics), the decoder shapes, the step
name=$1; flags=$2
bash tools/mkvariant.sh base > /dev/null 2>&1; bash tools/mkvariant.sh $name "" "$flags" > /dev/null 2>&1; ls build_ab/*.so
python tools/with_lib.py build_ab/$name.so tools/k3_check.py quick 2>&1 | grep -v amdgpu.ids | tail -4
export AM_CB_BATCH=16
AM_CB_STATS=1 bash tools/ab.sh "tools/conv_bench.py fwd 20" base $name
AM_CB_STATS=0 bash tools/ab.sh "tools/conv_bench.py fwd 20" base $name
bash tools/ab.sh "tools/conv_shapes_bench.py 16" base $name
bash tools/ab.sh "tools/step_run.py 16 12 1" base $name
