#!/bin/bash
# Round-6 session BG: Res2Net chain epilogue with a per-wave fast path for interior tiles (F) against the tree before it (UB2 = res2.hip@HEAD); micro-benchmark,
# then in situ (per-dispatch medians under rocprofv3 --kernel-trace)
TAG=${1:-r15bg}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "res2 or ecapa or bit or batch" 2>&1 | tail -2 | tee $OUT/pytest_subset_tail.log
for rep in 1 2 3; do
for lib in UB2 F; do
    export MV_PROBE_LIB=$REPO/tools/probe/libres2_$lib.so
    echo "== $lib" | tee -a $OUT/res2_micro.log
    timeout 300 python tools/bench_res2.py 2>&1 | grep "res2 chain" | tee -a $OUT/res2_micro.log
done
done
unset MV_PROBE_LIB
cd /tmp && export TMPDIR=/tmp
for lib in UB2 F F UB2 UB2 F; do
    rm -rf $OUT/prof
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/tools/bench_with_lib.py $REPO/tools/probe/libres2_$lib.so --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_$lib.log 2>&1
    f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
    python - <<PY | tee -a $OUT/in_situ.log
import csv, statistics, json
rows = [r for r in csv.DictReader(open('$f'))]
by = {}
for r in rows:
    by.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
line = [l for l in open('$OUT/bench_$lib.log') if l.startswith('{')]
v = json.loads(line[0])['value'] if line else None
r2 = [v_ for n, v_ in by.items() if 'res2_chain' in n][0][-90:]
ring = [v_ for n, v_ in by.items() if 'ring_persistent' in n][0][-210:]
print('%-5s headline %s  res2 chain in situ median %.1f us (min %.1f, n %d)  ring median %.1f' % ('$lib', v, statistics.median(r2), min(r2), len(r2), statistics.median(ring)))
PY
done
rm -rf $OUT/prof
