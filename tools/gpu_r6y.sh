#!/bin/bash
# Round-6 session Y (experiment, nothing shipped depends on it): what the platform's performance level does to the ring GEMM's clock.  rocm-smi --setperflevel high /
# back to auto around the hot ring launch (MFA shape) with its in-kernel clock.  Read-only if the container may not set it.
TAG=${1:-r15y}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
probe() {
MV_BENCH_CLOCK=1 MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$1', d['shape'], d['us'], d['TFLOPs'], d.get('clock_ghz'))" | tee -a $OUT/perflevel.log
}
rocm-smi --showperflevel 2>&1 | grep -i "level" | tee -a $OUT/perflevel.log
probe auto
timeout 60 rocm-smi --setperflevel high 2>&1 | tail -3 | tee -a $OUT/perflevel.log
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "level\|sclk" | tee -a $OUT/perflevel.log
probe high
probe high
timeout 60 rocm-smi --setperflevel auto 2>&1 | tail -2 | tee -a $OUT/perflevel.log
probe auto_again
