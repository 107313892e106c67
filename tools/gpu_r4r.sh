#!/bin/bash
# Round-4 session R: split conv2d v3 -- parity, per-layer timings with hints, timeline of three layers
TAG=${1:-r12r}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or eres2net" > $OUT/pytest_conv2ds.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_conv2ds.log
MV_BENCH_SWEEP=1 timeout 600 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d_b16.log 2>&1; echo "bench rc=$?"; cat $OUT/bench_conv2d_b16.log | cut -c1-700
MV_BENCH_SHAPES="s1 conv1,s1 3x3,s4 conv1" timeout 200 python tools/probe_conv2ds.py run 16 > $OUT/timeline.log 2>&1; grep -v "stage [2-9][0-9]\|occupancy" $OUT/timeline.log | head -60 | cut -c1-160
