#!/bin/bash
# round 3, call h: Res2Net chain drip form (y_j leaves through the x_{j+1} region during the next step's K loop) -- parity tests, micro A/B
# against the previous kernel (tools/probe/libres2_base.so = res2.hip@HEAD) and the same source with MV_RES2_DRIP=0, end to end
TAG=${1:-r10b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "res2 or ecapa" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2; do
  for lib in tools/probe/libres2_base.so tools/probe/libres2_nodrip.so ""; do
    echo "== lib=${lib:-product}" | tee -a $OUT/res2_micro.log
    MV_PROBE_LIB=$lib timeout 200 python tools/bench_res2.py 2>&1 | grep "res2 chain" | tee -a $OUT/res2_micro.log
  done
done
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('product', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/e2e.log; done
