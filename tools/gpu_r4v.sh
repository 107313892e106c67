#!/bin/bash
# Round-4 session V: the ERes2Net family end to end on the split conv path (m32 at bs 256, the 54.9 M model at 64 x 3 s, the bucketed config-5 leg), kernel stats
TAG=${1:-r12v}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for m in eres2netv2 eres2net; do
  timeout 300 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_$m.log 2>&1; echo "$m rc=$?"; grep "^{" $OUT/bench_$m.log | cut -c1-900
done
timeout 300 python bench.py --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_w96s4_b64.log 2>&1; echo "w96s4 rc=$?"; grep "^{" $OUT/bench_w96s4_b64.log | cut -c1-900
timeout 300 python -c "
import json, sys, torch
sys.argv = ['bench.py']
import bench
print(json.dumps(bench.bucketed_run('eres2netv2_w96s4', torch.device('cuda:0'), 64, 2)))
" > $OUT/bench_config5_bucketed.log 2>&1; echo "config5 rc=$?"; grep "^{" $OUT/bench_config5_bucketed.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --model eres2netv2_w96s4 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/rocprof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -10 $f | cut -c1-170; done
