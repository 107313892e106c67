#!/bin/bash
# ASP pooling quick session: parity tests, ring vs register form A/B, headline bench A/B.  usage: bash tools/gpu_quick_asp.sh <tag>
TAG=${1:-r04b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "asp or ecapa or native_library" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for impl in ring regs ring regs; do MV_ASP_IMPL=$impl timeout 300 python tools/bench_asp.py 2>&1 | grep nomax | sed "s/^/$impl /" | tee -a $OUT/asp_ab.log; done
for impl in ring regs; do MV_ASP_IMPL=$impl timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$impl', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/ab.log; done
