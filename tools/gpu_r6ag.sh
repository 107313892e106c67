#!/bin/bash
# Round-6 session AG: how fast would 64 x 64 sub-tiles run the ring GEMM's tail?  probe build: the 48 last-round tiles of the MFA layer as 768 sixteenths on the
# 64 x 64 kernel (three rounds of the chip) beside the product's 192 quarters and the unsplit launch; per-launch events
TAG=${1:-r15ag}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for lib in tail_off product tail_s16; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_PROF=1 MV_BENCH_T=300 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="mfa 3072,mfa 1536" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'], 'ring', d.get('ring_us'), 'sub-tile launches', d.get('other_launches'), d.get('other_us'))" | tee -a $OUT/bench_conv_ab.log
  done
done
