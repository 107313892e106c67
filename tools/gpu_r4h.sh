#!/bin/bash
# Round-4 session H: all GPU tests, small-batch GPU time + batch-1 kernel breakdown after the small-batch forms (64 x 64 conv tiles, stand-alone ASP
# statistics, 5-tile Res2Net chain chunks, pipelined time_stats / Fbank finish loads)
TAG=${1:-r12h}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED|^E  " $OUT/pytest_gpu.log | tail -12
for m in ecapa1024 campp; do for B in 1 2 8 32; do timeout 300 python tools/bench_latency.py $m $B 50 2>&1 | grep "GPU time" | tee -a $OUT/latency.log; done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -o b1 -- python $REPO/tools/bench_latency.py ecapa1024 1 50 > $OUT/b1.log 2>&1
find $OUT/prof_b1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "head -22 {} | cut -c1-150"
