#!/bin/bash
# Round-4 session E: all GPU tests (stand-alone ASP input statistics for small batches, batch-size invariance of embeddings), small-batch GPU time
TAG=${1:-r12e}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED|^E  " $OUT/pytest_gpu.log | tail -12
for m in ecapa1024 campp; do for B in 1 8 32; do timeout 300 python tools/bench_latency.py $m $B 50 2>&1 | grep "GPU time" | tee -a $OUT/latency.log; done; done
timeout 600 python tools/stress_determinism.py 30 2>&1 | grep "^{" | tee $OUT/stress_determinism.log
