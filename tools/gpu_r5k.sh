#!/bin/bash
# Round-5 session K: cam_dense_block_kernel's tail and layer entry -- the product (branch-free k = 3 phase, parameter loads as GLOBAL loads, BN1 tables
# stored in front of the k = 3 phase, context column sums from the h epilogue's registers, counted wait at the layer entry) against the kernel of the
# previous commit (camblock.hip@HEAD~: libcb_base) and against single switches turned off / on, CAM++ 256 x 3 s alternating in one call; CAM++ GPU tests first
TAG=${1:-r14k}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests -q -m gpu --timeout 400 -k "campp or long_and_short or batch_size" > $OUT/pytest_campp.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_campp.log | cut -c1-200
for rep in 1 2 3; do
  for lib in product base no_lazy lds_sums early_w; do
    case $lib in
      product) P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so;;
      base) P=$REPO/tools/probe/libcb_base.so;;
      no_lazy) P=$REPO/tools/probe/libcb_no_lazy.so;;
      lds_sums) P=$REPO/tools/probe/libcb_lds_sums.so;;
      early_w) P=$REPO/tools/probe/libcb_early_w.so;;
    esac
    timeout 300 python tools/bench_with_lib.py $P --model campp --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/campp_tail_ab.log
  done
done
cat > /tmp/lat.py <<PY
import sys, json, ctypes
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
r=bench.latency_batch1('campp', torch.device('cuda',0))
print(json.dumps(dict(lib=lib.split('/')[-1], eager_p50=r['eager_p50'], gpu_us=r['gpu_us_back_to_back'], graph_p50=r['hipgraph_p50'])))
PY
for rep in 1 2; do
  for lib in product $REPO/tools/probe/libcb_base.so; do
    timeout 300 python /tmp/lat.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/latency_batch1_campp_ab.log
  done
done
