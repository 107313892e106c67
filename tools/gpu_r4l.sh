#!/bin/bash
# Round-4 session L: rocprofv3 kernel stats of the headline command + PMC passes (FETCH / WRITE / TCC / SQ) on the final code of the round
TAG=${1:-r12l}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
head -20 $OUT/prof/bench_kernel_stats.csv | cut -c1-160
grep "^{" $OUT/rocprof.log | cut -c1-300
bash $REPO/tools/gpu_pmc.sh $TAG/pmc
