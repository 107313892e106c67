#!/bin/bash
# Round-5 session V: res2_chain_kernel (direct form) with the epilogue's bias / scale / shift requested one K stage ahead, against the previous kernel
# (libres2_base): the chain alone (tools/bench_res2.py, B = 256), one utterance (bench.latency_batch1) and the headline, alternating; res2 / golden tests first
TAG=${1:-r14v}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 300 python -m pytest tests -q -m gpu --timeout 300 -k "res2 or golden or batch_size" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log | cut -c1-200
for rep in 1 2; do
  for lib in product res2_base; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/libres2_base.so; fi
    timeout 200 python tools/bench_res2.py 2>/dev/null | grep "res2 chain" | sed "s/^/$lib /" | tee -a $OUT/bench_res2_ab.log
  done
done
unset MV_PROBE_LIB
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
  for lib in product $REPO/tools/probe/libres2_base.so; do
    timeout 300 python /tmp/lat.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/latency_batch1_res2_ab.log
  done
done
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
  for lib in product res2_base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libres2_base.so; fi
    timeout 300 python tools/bench_with_lib.py $P --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'])" | tee -a $OUT/headline_res2_ab.log
  done
done
