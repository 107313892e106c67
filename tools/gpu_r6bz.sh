#!/bin/bash
# Round-6 session BZ (third session): what do the library's per-launch HIP events (mv_profile_enable on every 4th timed step: the source of roofline.achieved) cost the headline?
# bench.py as shipped (every 4th step) against every 10th step and against no profiled step at all, the driver's arguments, A B C C B A x 2 in one call.
# bench_prof10.py / bench_prof0.py = bench.py with that one expression replaced by sed, temporary files of this session.
TAG=${1:-r15bz}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
run() { timeout 600 python $1 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$2', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline'].get('launches'))"; }
for i in 1 2; do
  run bench.py every4th; run bench_prof10.py every10th; run bench_prof0.py never; run bench_prof0.py never; run bench_prof10.py every10th; run bench.py every4th
done | tee $OUT/profile_event_cost_abccba.log
