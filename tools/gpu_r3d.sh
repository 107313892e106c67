#!/bin/bash
# ERes2NetV2 54.9 M (config 5 model) at 64 x 3 s: bench line + rocprofv3 kernel stats per conv2d instantiation.  usage: bash tools/gpu_r3d.sh <tag>
TAG=${1:-r07d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1
grep "^{" $OUT/rocprof.log | cut -c1-700
head -16 $OUT/prof/bench_kernel_stats.csv | cut -c1-200
