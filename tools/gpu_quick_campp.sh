#!/bin/bash
# CAM++ quick session: FCM layer parity tests + CAM++ model parity, A/B of the FCM kernels, kernel stats.  usage: bash tools/gpu_quick_campp.sh <tag>
TAG=${1:-r04a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fcm or campp or native_library" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for impl in band row band row; do MV_FCM_IMPL=$impl timeout 300 python bench.py --model campp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$impl', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/ab.log; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
head -8 $OUT/prof/bench_kernel_stats.csv | cut -c1-160
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof/bench_kernel_trace.csv")))
fc=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])-int(r['Start_Timestamp']),r['Kernel_Name'][:60]) for r in rows if 'fcm' in r['Kernel_Name']]
fc.sort()
for s,d,n in fc[-10:]: print(n, d/1000)
PY
