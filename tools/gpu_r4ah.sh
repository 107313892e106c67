#!/bin/bash
# Round-4 session AH: parity + model lines after the last default-shape changes
TAG=${1:-r12ah}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or eres2net or campp or hipgraph or batch_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for m in eres2netv2 eres2net; do timeout 300 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_$m.log 2>&1; grep "^{" $OUT/bench_$m.log | cut -c1-200; done
timeout 300 python bench.py --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_w96s4_b64.log 2>&1; grep "^{" $OUT/bench_w96s4_b64.log | cut -c1-200
timeout 300 python -c "
import json, sys, torch
sys.argv = ['bench.py']
import bench
print(json.dumps(bench.bucketed_run('eres2netv2_w96s4', torch.device('cuda:0'), 64, 2)))
" > $OUT/bench_config5_bucketed.log 2>&1; grep "^{" $OUT/bench_config5_bucketed.log | cut -c1-330
