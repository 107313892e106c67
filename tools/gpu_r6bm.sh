#!/bin/bash
# Round-6 session BM: time_stats_kernel LEAN form for the SE squeeze (mean alone, no pre-activation: 3 instead of 7 vector operations per value) against pool.hip@HEAD:
# GPU tests, in situ per-dispatch medians, headline ABBA
TAG=${1:-r15bm}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "stat or ecapa or tdnn or bit or batch or pool" 2>&1 | tail -2 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
cd /tmp && export TMPDIR=/tmp
for lib in pool_prev product product pool_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    rm -rf $OUT/prof
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/tools/bench_with_lib.py $P --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_$lib.log 2>&1
    f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
    python - <<PY | tee -a $OUT/in_situ.log
import csv, statistics, json
rows = [r for r in csv.DictReader(open('$f'))]
by = {}
for r in rows:
    by.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
line = [l for l in open('$OUT/bench_$lib.log') if l.startswith('{')]
v = json.loads(line[0])['value'] if line else None
def med(key, n):
    x = [v_ for k, v_ in by.items() if key in k]
    return statistics.median(x[0][-n:]) if x else float('nan')
print('%-10s headline %s  time_stats in situ median %.1f us  se_gate %.1f  ring %.1f' % ('$lib', v, med('time_stats', 90), med('se_gate', 90), med('ring_persistent', 210)))
PY
done
rm -rf $OUT/prof
cd $REPO
for lib in pool_prev product product pool_prev pool_prev product product pool_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
