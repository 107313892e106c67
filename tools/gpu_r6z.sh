#!/bin/bash
# Round-6 session Z: per-dispatch durations of the ring GEMM inside the headline step (rocprofv3 --kernel-trace): the six K = 1024 layers and the MFA layer in situ
TAG=${1:-r15zz}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench.log 2>&1; echo "rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - <<PY | tee $OUT/ring_in_situ.log
import csv, statistics
rows = [r for r in csv.DictReader(open('$f'))]
ring = [(int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'conv1d_ring_persistent' in r['Kernel_Name']]
ring.sort()
durs = [d for _, d in ring][-7 * 30:]           # the last 30 steps
mfa = [d for d in durs if d > 600]
k1 = [d for d in durs if d <= 600]
print('in situ, last 30 steps: MFA (K = 3072) median %.1f us (min %.1f max %.1f, n %d); K = 1024 layers median %.1f us (min %.1f max %.1f, n %d)' % (statistics.median(mfa), min(mfa), max(mfa), len(mfa), statistics.median(k1), min(k1), max(k1), len(k1)))
by = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-48:]
    by.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print('%-50s n %5d median %8.1f' % (n, len(v), statistics.median(v)))
PY
rm -rf $OUT/prof
