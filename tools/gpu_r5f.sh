#!/bin/bash
# Round-5 final session: GPU suite, smoke, the driver's bench command, kernel stats of the headline and CAM++ on the final tree
TAG=${1:-r14f}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench.log | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ecapa -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_ecapa.log 2>&1
for f in $(find $OUT/prof_ecapa -name "*kernel_stats*.csv"); do head -6 $f | cut -c1-150; done
