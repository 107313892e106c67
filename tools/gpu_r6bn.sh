#!/bin/bash
# Round-6 session BN: the tree as it is left: the driver's commands (full GPU suite, smoke, python bench.py), then rocprofv3 --kernel-trace --stats of the bench
# command (Ecapa headline and the CAM++ leg)
TAG=${1:-r15bn}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
bash $REPO/tools/gpu_r6g.sh $TAG
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_ecapa1024.csv; rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_campp_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_campp.csv; rm -rf $OUT/prof
head -14 $OUT/kernel_stats_ecapa1024.csv | cut -c1-140
