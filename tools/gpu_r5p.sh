#!/bin/bash
# Round-5 session P: rocprofv3 kernel stats of ERes2NetV2 (54.9 M) 64 x 3 s and of CAM++ 256 x 3 s after the r14k-o changes
TAG=${1:-r14p}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for m in "eres2netv2_w96s4 64 3" "campp 256 10"; do
  set -- $m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1 -o bench -- python $REPO/bench.py --model $1 --batch $2 --steps $3 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_$1.log 2>&1
  f=$(find $OUT/prof_$1 -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_$1.csv; rm -rf $OUT/prof_$1
  grep "^{" $OUT/rocprof_$1.log | cut -c1-300
  head -14 $OUT/rocprofv3_kernel_stats_$1.csv | cut -c1-170
done
