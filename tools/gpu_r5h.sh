#!/bin/bash
# Round-5 session H: the CAM++ exact-head leg with the peak word's atomic guarded by a relaxed load (product) against every wave's atomic going out
# (tools/probe/libpeak_unguarded.so = the same sources with -DMV_S16_UNGUARDED_PEAK), alternating in one call
TAG=${1:-r14h}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
cat > /tmp/exact_leg.py <<PY
import sys, json, ctypes
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
for head in ('f32', None):
    r=bench.short_run('campp', torch.device('cuda',0), 256, 10, 3, 8, head=head, repeats=3)
    print(json.dumps(dict(lib=lib.split('/')[-1], head=head or 'auto', value=r['value'], ms=r['ms_per_step'], repeats=r.get('repeats'), fcm_head=r.get('fcm_head'))))
PY
for rep in 1 2; do
  for lib in product $REPO/tools/probe/libpeak_unguarded.so; do
    timeout 300 python /tmp/exact_leg.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/exact_head_peak_atomic_ab.log | cut -c1-330
  done
done
