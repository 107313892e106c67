#!/bin/bash
# Round-5 session Z: in-kernel timeline of conv2ds_kernel after the r14o epilogue fix (tools/probe_conv2ds.py) on the three stage-1 layers of the
# 54.9 M ERes2NetV2 (80 x 298 maps, B = 16) and the kernels' durations next to it (tools/bench_conv2d.py)
TAG=${1:-r14z}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
MV_BENCH_SHAPES="s1 conv1,s1 3x3,s1 conv3,s2 3x3" timeout 300 python tools/probe_conv2ds.py run 16 > $OUT/conv2ds_inkernel_timeline.log 2>&1; grep -v "^$\|amdgpu" $OUT/conv2ds_inkernel_timeline.log | head -70 | cut -c1-170
MV_BENCH_SHAPES="s1 conv1,s1 3x3,s1 conv3,s2 3x3,s2 conv3,s3 3x3" timeout 300 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d.log 2>&1; grep -v "^$\|amdgpu" $OUT/bench_conv2d.log | tail -12 | cut -c1-200
