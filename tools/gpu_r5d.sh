#!/bin/bash
# Round-5 session D: what the NaN-propagating S16 clamps cost the ERes2Net family (r14c: config 5 517 utt/s against 692 in r13b) -- product
# (NaN test, peak tracking as a template arm) against the same source with the round-4 clamp (-DMV_S16_PLAIN_CLAMP), alternating in one call:
# per-layer timings and the config-5 model line
TAG=${1:-r14d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
  for lib in product plainclamp; do
    if [ $lib = product ]; then P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so; else P=$REPO/tools/probe/libconv2ds_plainclamp.so; fi
    MV_PROBE_LIB=$P timeout 300 python tools/bench_conv2d.py 16 2>/dev/null | grep "^{" > $OUT/layers_${lib}_$rep.log
    timeout 300 python tools/bench_with_lib.py $P --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | cut -c1-400 > $OUT/model_${lib}_$rep.log
    echo "$lib $rep: $(python -c "import json,sys; d=json.loads(open('$OUT/model_${lib}_$rep.log').read()[:400].split(', \"higher')[0]+'}'); print(d['value'], d['ms_per_step'])")"
  done
done
python - <<PY
import json,glob
for rep in (1,2):
    rows={}
    for lib in ('product','plainclamp'):
        for l in open('$OUT/layers_%s_%d.log'%(lib,rep)):
            d=json.loads(l); rows.setdefault(d['layer'],{})[lib]=d.get('split_us')
    print('rep',rep,' | '.join(f"{k}: {v.get('product')} / {v.get('plainclamp')}" for k,v in rows.items()))
PY
timeout 300 python bench.py --model eres2netv2_w96s4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_config5_default_path.log 2>&1
timeout 600 python - <<PY > $OUT/config5_bucketed.log 2>&1
import sys, json
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch, bench
print(json.dumps(bench.bucketed_run('eres2netv2_w96s4', torch.device('cuda',0), 64, 2))[:600])
PY
tail -2 $OUT/config5_bucketed.log | cut -c1-600
