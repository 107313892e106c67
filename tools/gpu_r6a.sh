#!/bin/bash
# Round-6 session A: (1) the tree after the hygiene / ABI-5 batch: smoke, full GPU suite; (2) ring GEMM with the last MFMA group carried across the stage
# barrier: base (24a38ec) / v1 (carry) / product (carry + K-half-1 activation fragments one step early) -- the layers alone and the headline, alternating
# in one call; (3) the driver's bench command with the new `box` block.
TAG=${1:-r15a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-200
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
for rep in 1 2; do
  for lib in base v1 product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/libring_$lib.so; fi
    MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072,c2c 512,mfa 1536" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'])" | tee -a $OUT/bench_conv_ab.log
  done
done
unset MV_PROBE_LIB
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d = json.loads([l for l in open('$OUT/bench.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'conv frac', d['roofline']['frac'], 'fbank', d['roofline_fbank']['avg_launch_us'], d['roofline_fbank']['frac'])
print('box', d.get('box'))
oc = d.get('other_configs', {})
for k, v in oc.items():
    print(k, v.get('value'), v.get('parity'), v.get('error'))
print('lat', {k: (v.get('eager_p50'), v.get('gpu_us_back_to_back')) for k, v in d.get('latency_batch1', {}).items()})
PY
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2; do
  for lib in base v1 product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libring_$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/headline_ab.log
  done
done
