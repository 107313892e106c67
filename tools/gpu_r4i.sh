#!/bin/bash
# Round-4 session I: kernel breakdown of predict_batch-sized batches (B = 32 / 64)
TAG=${1:-r12i}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for B in 32 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b$B -o b -- python $REPO/tools/bench_latency.py ecapa1024 $B 30 > $OUT/b$B.log 2>&1
  grep "GPU time" $OUT/b$B.log
  find $OUT/prof_b$B -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "head -16 {} | cut -c1-140"
done
