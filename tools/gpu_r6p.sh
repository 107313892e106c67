#!/bin/bash
# Round-6 session P: element-wise passes with one 16-byte group per thread (no capped grid-stride walk): product against the previous pool.hip, headline and CAM++
# (bn_relu_rows) alternating; full GPU suite first
TAG=${1:-r15p}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
  for lib in product pool_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    for m in ecapa1024 campp; do
      timeout 300 python tools/bench_with_lib.py $P --model $m --no-cpu-baseline --no-other-configs --no-box 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, '$m', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $OUT/elementwise_grid_ab.log
    done
  done
done
