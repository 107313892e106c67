#!/bin/bash
# Round-6 session AZ: rocprofv3 --kernel-trace --stats of the CAM++ leg (BASELINE config 3) on the tree as it is left
TAG=${1:-r15az}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_campp_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_campp.csv; rm -rf $OUT/prof
head -12 $OUT/kernel_stats_campp.csv | cut -c1-150
grep "^{" $OUT/bench_campp_under_rocprof.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), d.get('parity'))"
