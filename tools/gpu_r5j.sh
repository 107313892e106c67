#!/bin/bash
# Round-5 session J: the split-K linears chosen by (K, O) only -- full GPU suite (bit-identity over batch sizes), batch-1 latency and the headline,
# product against -DMV_LINEAR_NO_SPLITK alternating in one call
TAG=${1:-r14j}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
cat > /tmp/lat.py <<PY
import sys, json, ctypes
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
r=bench.latency_batch1('ecapa1024', torch.device('cuda',0))
print(json.dumps(dict(lib=lib.split('/')[-1], eager_p50=r['eager_p50'], gpu_us=r['gpu_us_back_to_back'], graph_p50=r['hipgraph_p50'])))
PY
for rep in 1 2; do
  for lib in product $REPO/tools/probe/liblinear_nosplitk.so; do
    timeout 300 python /tmp/lat.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/latency_batch1_splitk_ab.log
  done
done
for rep in 1 2 3; do
  for lib in product nosplitk; do
    if [ $lib = product ]; then P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so; else P=$REPO/tools/probe/liblinear_nosplitk.so; fi
    timeout 300 python tools/bench_with_lib.py $P --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'])" | tee -a $OUT/headline_splitk_ab.log
  done
done
