#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprof kernel stats.  Everything lands in gpurun_out/.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
cd $(dirname $0)/..
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt 2>&1
nproc >> $OUT/rocminfo.txt
echo "== smoke" ; timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
tail -3 $OUT/bench.log
echo "== rocprof"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -30 $f; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/prof_campp -o bench -- python $REPO/bench.py --model campp --steps 3 --warmup 2 --no-cpu-baseline > $OUT/rocprof_campp.log 2>&1
cd $REPO
echo "== bench campp"; timeout 600 python bench.py --model campp --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_campp.log 2>&1; tail -1 $OUT/bench_campp.log
echo "== bench ecapa512"; timeout 600 python bench.py --model ecapa512 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_ecapa512.log 2>&1; tail -1 $OUT/bench_ecapa512.log
