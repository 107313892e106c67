#!/bin/bash
# Round-6 session CB (fourth session): rocprofv3 --kernel-trace --stats of the bench command on the tree as it is committed last (Ecapa headline and the CAM++ leg)
TAG=${1:-r15cb}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_ecapa1024.csv; rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_campp_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_campp.csv; rm -rf $OUT/prof
head -8 $OUT/kernel_stats_ecapa1024.csv | cut -c1-160
grep '^{' $OUT/bench_under_rocprof.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('live', d['value'], r['frac'], r['avg_launch_us'], r['launches'])"
