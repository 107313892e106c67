#!/bin/bash
# Round-4 session J: which conv tile for which batch size (predict_batch-sized batches): 64 / 128 / 160 / 256 forced, B = 8 .. 128
TAG=${1:-r12j}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
for B in 8 16 32 48 64 96 128; do
  MV_BENCH_B=$B MV_BENCH_TILES=64,128,160,256 MV_BENCH_SHAPES="c2c 1024,mfa 3072,asp 3072" timeout 300 python tools/bench_conv.py 2>&1 | grep "^{" | tee -a $OUT/conv_tiles_by_batch.log
done
