#!/bin/bash
# Round-6 session AE: the quarter-tile kernel of the ring GEMM's tail: 4 waves of 64 x 64 (product) / 8 waves of 64 co x 32 rows (tail_8w) / 8 waves of
# 32 co x 64 rows (tail_8wb) / no split (tail_off); per-launch HIP events of the library (ring kernel and quarter kernel apart), alternating in one call
TAG=${1:-r15ae}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for lib in tail_off product tail_8w tail_8wb; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_PROF=1 MV_BENCH_T=300 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="mfa 3072,mfa 1536" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'], 'ring', d.get('ring_us'), 'quarters', d.get('other_launches'), d.get('other_us'))" | tee -a $OUT/bench_conv_ab.log
  done
done
