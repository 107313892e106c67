#!/bin/bash
# Round-6 session D: TIMING PROBES of the ring GEMM (wrong results on purpose, never shipped): what would a form with fewer LDS fragment reads per MFMA
# (one wave per SIMD on 128 x 128 wave tiles: 32 reads per 128 MFMAs instead of 48) or fewer transfers gain?  product / fewreads (K half 1 reuses K half 0's
# weight fragments: 16 instead of 24 reads per stage and wave) / nobf1 (20 of 24) / now (no weight transfers: 4 instead of 8 per wave and stage), with the
# in-kernel clock of each (MvConv1dDesc.clock_probe).
TAG=${1:-r15d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
  for lib in product ring_probe_noepi ring_probe_nostore; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_CLOCK=1 MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'], d.get('clock_ghz'))" | tee -a $OUT/ring_probes.log
  done
done
