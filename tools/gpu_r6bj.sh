#!/bin/bash
# Round-6 session BJ: the ASP hidden conv's two forms IN SITU (per-dispatch medians of the headline step under rocprofv3 --kernel-trace): conv_prev = conv1d.hip@HEAD
# (statistics rows stored from the reduction), product = through the LDS buffer
TAG=${1:-r15bj}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
cd /tmp && export TMPDIR=/tmp
for lib in conv_prev product product conv_prev conv_prev product; do
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
print('%-9s headline %s  asp hidden conv in situ median %.1f us  asp pool %.1f  block0 %.1f  ring %.1f  res2 %.1f' % ('$lib', v, med('conv1d_glds_kernel<2, 2, 4, 5, true', 30), med('asp_pool_ring', 30), med('glds_persistent', 30), med('ring_persistent', 210), med('res2_chain', 90)))
PY
done
rm -rf $OUT/prof
