#!/bin/bash
# Round-6 session BD: the Res2Net chain's arms IN SITU (the micro-benchmark and the headline disagreed in r15bb / r15bc): per-dispatch durations of every kernel of
# the headline step under rocprofv3 --kernel-trace, one bench.py run per arm, arms alternating.  base = res2.hip@HEAD, U = unconditional fragment requests + six
# written-out stages, UB = U + epilogue parameters requested in front of the last stage's last ten MFMAs, AB = UB + two-pass epilogue
TAG=${1:-r15bd}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for lib in base U UB AB base U UB AB; do
    rm -rf $OUT/prof
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/tools/bench_with_lib.py $REPO/tools/probe/libres2_$lib.so --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_$lib.log 2>&1
    f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
    python - <<PY | tee -a $OUT/in_situ.log
import csv, statistics, json
rows = [r for r in csv.DictReader(open('$f'))]
by = {}
for r in rows:
    n = r['Kernel_Name']
    by.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
line = [l for l in open('$OUT/bench_$lib.log') if l.startswith('{')]
v = json.loads(line[0])['value'] if line else None
r2 = [v_ for n, v_ in by.items() if 'res2_chain' in n][0][-90:]
ring = sorted([v_ for n, v_ in by.items() if 'ring_persistent' in n][0][-210:])
print('%-5s headline %s  res2 chain in situ median %.1f us (min %.1f, n %d)  ring median %.1f  se_gate %.1f' % ('$lib', v, statistics.median(r2), min(r2), len(r2), statistics.median(ring), statistics.median([v_ for n, v_ in by.items() if 'se_gate' in n][0][-90:])))
PY
done
rm -rf $OUT/prof
