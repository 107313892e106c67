#!/bin/bash
# Round-6 session R: does the headline depend on how long the chip has been under load?  (the ring probe of the box block: 1515-1550 us at 1.50-1.57 GHz behind two
# warm-up launches, 1217 us at 1.80-1.82 GHz behind twenty-four)  bench.py --steps 20 with --warmup 5 / 30 / 100, and --steps 100 / 300 --warmup 5
TAG=${1:-r15t}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
  for cfg in "20 5" "20 30" "20 100" "100 5" "300 5" "20 5"; do
    set -- $cfg
    timeout 600 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-other-configs --no-box 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('steps', $1, 'warmup', $2, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_fbank']['avg_launch_us'])" | tee -a $OUT/warmup_sweep.log
  done
done
