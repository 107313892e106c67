#!/bin/bash
# Round-6 session AD: the ring GEMM's tail as 128 x 128 quarter tiles (conv1d_launch): new tests, then the MFA layer alone and the headline, alternating in one
# call: tail_off (the split disabled) / tail_ns3 (three-stage ring in the quarter kernel) / product (four stages)
TAG=${1:-r15ad}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "ring_tail or conv1d or bits_do_not_depend or profile_classes or golden" > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_subset.log | cut -c1-300
for rep in 1 2 3; do
  for lib in tail_off tail_ns3 product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_T=300 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="mfa 3072,c2c 1024,mfa 1536" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'])" | tee -a $OUT/bench_conv_ab.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
  for lib in tail_off product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
rc = d.get('roofline_conv1d_class', {})
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'class', rc.get('frac'), rc.get('launches'), d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
