#!/bin/bash
# Round-6 session K: asp_pool_ring_kernel ring depth 4 (product) / 6 (rounds 2-5) / 3, alternating; then the ASP / Ecapa GPU tests on the product
TAG=${1:-r15k}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3 4; do
  for lib in product asp_r6 asp_r3; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 120 python tools/bench_asp.py 2>/dev/null | grep "^{" | grep nomax | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', $rep, d['us'], d['x_GBps'])" | tee -a $OUT/bench_asp_pool_ab.log
  done
done
unset MV_PROBE_LIB
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "asp or ecapa or bits or tdnn or predictor" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log | cut -c1-200
