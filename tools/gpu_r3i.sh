#!/bin/bash
# round 3, call i: CAM++ transit layers -- BatchNorm + ReLU written out once + the conv on the direct path (MV_CAMPP_TRANSIT=pre) against the
# transform-on-load register path (load); parity tests, end-to-end A/B alternating in one call, kernel stats
TAG=${1:-r10d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "campp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in load pre default load pre default; do
  if [ $v = default ]; then unset MV_CAMPP_TRANSIT; else export MV_CAMPP_TRANSIT=$v; fi
  timeout 300 python bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('MV_CAMPP_TRANSIT=$v', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/ab.log
done
unset MV_CAMPP_TRANSIT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
head -14 $OUT/prof/bench_kernel_stats.csv | cut -c1-200
