#!/bin/bash
# Round-6 session E: rocprofv3 kernel stats of the headline step on the current tree (Ecapa-1024) and of CAM++
TAG=${1:-r15e}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for m in ecapa1024 campp; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o k -- python $REPO/bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_$m.log 2>&1; echo "$m rc=$?"
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_$m.csv
  rm -rf $OUT/prof_$m
done
cd $REPO
python - <<PY
import csv
for m in ('ecapa1024', 'campp'):
    rows = list(csv.DictReader(open('$OUT/kernel_stats_%s.csv' % m)))
    print(m)
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:26]:
        print('  %-90s %5s %9.1f %5.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
