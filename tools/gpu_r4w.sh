#!/bin/bash
# Round-4 session W: pixel-split consumer variants -- parity, per-layer timings (incl. the m32 shapes), m32 model lines, timeline of two layers
TAG=${1:-r12w}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or eres2net" > $OUT/pytest_conv2ds.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_conv2ds.log
MV_BENCH_SWEEP=1 timeout 600 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d_b16.log 2>&1; echo "bench rc=$?"; python - <<PY
import json
for l in open('$OUT/bench_conv2d_b16.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['layer'], '| f32', d['f32_us'], '| split', d.get('split_us'), '| GB/s', d.get('split_GBps'), '| TFx3', d.get('split_mfma_tflops_x3'), '|', {k.replace('_us',''): v for k, v in d.items() if k.startswith('nbw')})
PY
for m in eres2netv2 eres2net; do
  timeout 300 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_$m.log 2>&1; echo "$m rc=$?"; grep "^{" $OUT/bench_$m.log | cut -c1-330
done
timeout 300 python bench.py --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_w96s4_b64.log 2>&1; echo "w96s4 rc=$?"; grep "^{" $OUT/bench_w96s4_b64.log | cut -c1-330
MV_BENCH_SHAPES="s4 conv1" timeout 200 python tools/probe_conv2ds.py run 16 > $OUT/timeline.log 2>&1; grep -v "stage [2-9][0-9]\|occupancy\|workgroups" $OUT/timeline.log | head -50 | cut -c1-150
