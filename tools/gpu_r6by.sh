#!/bin/bash
# Round-6 session BY (third session): bench.py with the box yard-stick directly in front of the warm-up steps (the synthetic batch drawn before it) against the previous order,
# with the DRIVER's arguments (--steps 20 --warmup 5), ABBA x 3 in one call.  bench_prev_order.py = `git show HEAD~:bench.py`, a temporary file of this session.
TAG=${1:-r15by}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
run() { timeout 600 python $1 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$2', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], 'fbank_us', d['roofline_fbank']['avg_launch_us'], 'ring clock', d['box'].get('ring_k3072_clock_ghz'))"; }
for i in 1 2 3; do
  run bench_prev_order.py prev; run bench.py new; run bench.py new; run bench_prev_order.py prev
done | tee $OUT/box_probe_order_abba.log
