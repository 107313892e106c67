#!/bin/bash
# Round-5 session E: GPU suite on the tree without the fp32 conv2d form (58 exports) + bench.py under a launcher with one rank (RCCL new_group);
# the fp32 yard-stick's own check; Fbank warm vs cold memory hierarchy; the tile-quantisation staircase of the K = 1024 ring-GEMM layer
TAG=${1:-r14e}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --durations=8 --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log | cut -c1-200
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python tools/yardstick/check_conv2d.py > $OUT/yardstick_check_conv2d.log 2>&1; echo "yardstick rc=$?"; tail -1 $OUT/yardstick_check_conv2d.log
MV_BENCH_SHAPES="s2 3x3,s3 conv1" timeout 300 python tools/bench_conv2d.py 16 2>/dev/null | grep "^{" | cut -c1-300 > $OUT/bench_conv2d_yardstick_vs_split.log; cat $OUT/bench_conv2d_yardstick_vs_split.log
for f in 0 1; do timeout 200 python tools/bench_fbank_cold.py $f 2>/dev/null | grep "^{" >> $OUT/fbank_warm_vs_cold.log; done; cat $OUT/fbank_warm_vs_cold.log
# K = 1024 dense layer at 772 .. 1284 tiles of 256 x 256 on 256 CUs: 3.02 / 3.5 / 3.98 / 4.33 / 4.66 (the headline) / 4.98 / 5.02 rounds
for b in 165 192 219 238 256 274 275; do
  MV_BENCH_B=$b MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024" timeout 120 python tools/bench_conv.py 2>/dev/null | grep "^{" >> $OUT/ring_gemm_k1024_tile_staircase.log
done
cat $OUT/ring_gemm_k1024_tile_staircase.log
