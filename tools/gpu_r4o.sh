#!/bin/bash
# Round-4 session O: split-fp16 conv2d, persistent ring form -- parity cases, per-layer timings with the launch-shape hints, ERes2Net goldens
TAG=${1:-r12o}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or test_conv2d or tstp or eres2net" > $OUT/pytest_conv2ds.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_conv2ds.log
MV_BENCH_SWEEP=1 timeout 900 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d_b16.log 2>&1; echo "bench rc=$?"; cat $OUT/bench_conv2d_b16.log | cut -c1-900
