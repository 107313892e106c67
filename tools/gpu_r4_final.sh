#!/bin/bash
# Round-4 final session: all GPU tests, smoke, the default bench (headline + other configurations + CPU baseline), kernel stats of the headline and of the 54.9 M ERes2NetV2
TAG=${1:-r12final}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench.log | tail -1 | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_w96s4 -o bench -- python $REPO/bench.py --model eres2netv2_w96s4 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/rocprof_w96s4.log 2>&1
for f in $(find $OUT/prof_w96s4 -name "*kernel_stats*.csv"); do head -9 $f | cut -c1-150; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ecapa -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_ecapa.log 2>&1
for f in $(find $OUT/prof_ecapa -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-150; done
