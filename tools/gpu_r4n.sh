#!/bin/bash
# Round-4 session N: first GPU contact of the split-fp16 conv2d form -- MFMA subnormal probe, parity cases, per-layer fp32 vs split timings
TAG=${1:-r12n}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o /tmp/mfma_denorm_probe > /dev/null 2>&1 && timeout 60 /tmp/mfma_denorm_probe > $OUT/mfma_denorm_probe.log 2>&1
cat $OUT/mfma_denorm_probe.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or test_conv2d or tstp" > $OUT/pytest_conv2ds.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_conv2ds.log
MV_BENCH_SWEEP=1 timeout 900 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d_b16.log 2>&1; echo "bench rc=$?"; cat $OUT/bench_conv2d_b16.log | cut -c1-600
