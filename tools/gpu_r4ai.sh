#!/bin/bash
# Round-4 session AI (last of the round): the tree after the emulator sweep's fixes -- all GPU tests, smoke, conv2ds layers against the library of
# the previous session (tools/probe/libmvector_hip_r12.so, A/B in one call), the default bench, kernel stats of the headline
TAG=${1:-r13b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 90 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for arm in r12 new; do
  if [ $arm = r12 ]; then export MV_PROBE_LIB=$REPO/tools/probe/libmvector_hip_r12.so; else unset MV_PROBE_LIB; fi
  timeout 90 python tools/bench_conv2d.py 16 > $OUT/bench_conv2d_$arm.log 2>&1; echo "conv2d $arm rc=$?"
done
unset MV_PROBE_LIB
python - <<PY
import json
rd = lambda f: {d['layer']: d for d in (json.loads(l) for l in open(f) if l.startswith('{'))}
a, b = rd('$OUT/bench_conv2d_r12.log'), rd('$OUT/bench_conv2d_new.log')
for k in a:
    if k in b: print(f"{k:28s} split r12 {a[k].get('split_us')} us  new {b[k].get('split_us')} us  ratio {b[k].get('split_us', 0) / max(a[k].get('split_us', 1), 1e-9):.3f}")
PY
timeout 300 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench.log | tail -1 | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ecapa -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_ecapa.log 2>&1
for f in $(find $OUT/prof_ecapa -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-150; done
