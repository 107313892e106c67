#!/bin/bash
# Round-3 session a: fused BasicResBlock kernel -- parity tests, CAM++ A/B (fused vs two launches, tile widths), kernel stats.
# usage (repo root on the GPU box): bash tools/gpu_r3a.sh <tag>
TAG=${1:-r06a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fcm or campp or native_library" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
line() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], j['parity'])"; }
for rep in 1 2; do
  MV_FCM_FUSED=0 timeout 300 python bench.py --model campp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | line unfused | tee -a $OUT/ab.log
  for nt in 5 4 3; do MV_FCM_BLOCK_NT=$nt timeout 300 python bench.py --model campp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | line fused_nt$nt | tee -a $OUT/ab.log; done
done
cd /tmp && export TMPDIR=/tmp
for nt in 5 3; do
MV_FCM_BLOCK_NT=$nt timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_nt$nt -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof_nt$nt.log 2>&1
head -12 $OUT/prof_nt$nt/bench_kernel_stats.csv | cut -c1-200
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof_nt$nt/bench_kernel_trace.csv")))
fc=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])-int(r['Start_Timestamp']),r['Kernel_Name'][:70]) for r in rows if 'fcm' in r['Kernel_Name']]
fc.sort()
for s,d,n in fc[-6:]: print(n, d/1000)
PY
done
