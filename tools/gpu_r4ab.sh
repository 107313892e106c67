#!/bin/bash
# Round-4 session AB: tile height by cost model -- 3x3 layers at B = 16 and B = 4 (defaults only), the bucketed config-5 leg
TAG=${1:-r12ab}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for b in 16 4; do
MV_BENCH_SHAPES="3x3" timeout 300 python tools/bench_conv2d.py $b > $OUT/bench_conv2d_b$b.log 2>&1; python - <<PY
import json
for l in open('$OUT/bench_conv2d_b$b.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('B=$b', d['layer'], '| f32', d['f32_us'], '| split', d.get('split_us'), '| TFx3', d.get('split_mfma_tflops_x3'))
PY
done
timeout 300 python -c "
import json, sys, torch
sys.argv = ['bench.py']
import bench
print(json.dumps(bench.bucketed_run('eres2netv2_w96s4', torch.device('cuda:0'), 64, 2)))
" > $OUT/bench_config5_bucketed.log 2>&1; echo "config5 rc=$?"; grep "^{" $OUT/bench_config5_bucketed.log | cut -c1-420
timeout 300 python bench.py --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_w96s4_b64.log 2>&1; grep "^{" $OUT/bench_w96s4_b64.log | cut -c1-250
