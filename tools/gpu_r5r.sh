#!/bin/bash
# Round-5 session R: the bucketed ERes2NetV2 leg (BASELINE config 5: 64 utterances of 1-10 s, 8 buckets), product against the conv2ds kernel of before
# r14o (libc2ds_base), alternating in one call
TAG=${1:-r14r}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
cat > /tmp/bucket_leg.py <<PY
import sys, json, ctypes
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
r=bench.bucketed_run('eres2netv2_w96s4', torch.device('cuda',0), 64, 3)
print(json.dumps(dict(lib=lib.split('/')[-1], value=r['value'], parity=r.get('parity', {}).get('max_one_minus_cos'))))
PY
for rep in 1 2 3; do
  for lib in product $REPO/tools/probe/libc2ds_base.so; do
    timeout 300 python /tmp/bucket_leg.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/config5_bucketed_conv2ds_epilogue_ab.log
  done
done
