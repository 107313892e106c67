#!/bin/bash
# Round-6 session AK: fewer resident workgroups than CUs for the K = 1024 layers?  1192 tiles on 256 workgroups are 4.66 rounds (5 tile times, 168 tiles in the
# last round); 240 workgroups walk 4.97 rounds -- the same five tile times with 16 CUs idle throughout
TAG=${1:-r15ak}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for blocks in 0 248 240; do
    MV_BENCH_BLOCKS=$blocks MV_BENCH_T=298 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('blocks', $blocks, $rep, d['shape'], d['us'], d['TFLOPs'])" | tee -a $OUT/bench_conv_blocks.log
  done
done
