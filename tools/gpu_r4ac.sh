#!/bin/bash
# Round-4 session AC: straight-line MFMA groups for full tiles (one block per wave) against the committed kernel, in one call; then the parity tests
TAG=${1:-r12ac}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export MV_PROBE_LIB=$REPO/tools/probe/libconv2ds_base.so; else unset MV_PROBE_LIB; fi
  MV_BENCH_SHAPES="s1 3x3,s2 3x3,m32" timeout 300 python tools/bench_conv2d.py 16 > $OUT/bench_${lib}_$rep.log 2>&1
  python - <<PY
import json
print('$lib $rep', ' | '.join('%s %.1f' % (json.loads(l)['layer'][:16], json.loads(l)['split_us']) for l in open('$OUT/bench_${lib}_$rep.log') if l.startswith('{')))
PY
done; done
unset MV_PROBE_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or eres2net or campp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for m in eres2netv2; do timeout 300 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_$m.log 2>&1; grep "^{" $OUT/bench_$m.log | cut -c1-200; done
