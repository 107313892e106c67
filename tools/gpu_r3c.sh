#!/bin/bash
# quick session: fcm block micro-benchmark + parity tests + CAM++ bench line.  usage: bash tools/gpu_r3c.sh <tag>
TAG=${1:-r06e}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fcm_block or campp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do timeout 300 python tools/bench_fcm.py 2>&1 | grep "fcm block" | tee -a $OUT/fcm.log; done
for rep in 1 2; do timeout 300 python bench.py --model campp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('campp', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/campp.log; done
