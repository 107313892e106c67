#!/bin/bash
# Round-6 session AH: dense tile ids (no holes in the persistent walk) + the ring GEMM's tail as 64 x 64 sixteenths / 128 x 128 quarters (the product) against
# tail_off (dense ids, no split), tail_q_only (dense ids, quarters only) and tail_q (the previous commit: ids with holes, 48 tiles as 192 quarters):
# full GPU suite, the MFA layers alone with per-launch events, then the headline alternating
TAG=${1:-r15ah}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
for rep in 1 2 3; do
  for lib in tail_off tail_q tail_q_only product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_PROF=1 MV_BENCH_T=300 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="mfa 3072,mfa 1536,c2c 1024" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'], 'ring', d.get('ring_us'), 'sub-tile launches', d.get('other_launches'), d.get('other_us'))" | tee -a $OUT/bench_conv_ab.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3 4; do
  for lib in tail_off tail_q product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
rc = d.get('roofline_conv1d_class', {})
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'class', rc.get('frac'), rc.get('launches'), d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
