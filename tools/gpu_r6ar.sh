#!/bin/bash
# Round-6 session AR: the ring GEMM's epilogue with the fp16 saturation left to MODE.FP16_OVFL (256 v_med3 gone: 2.5 instead of 3.5 vector issue slots per value):
# conv / model tests of the GPU suite, the layers alone and the headline, alternating with conv_base (the previous conv1d.hip)
TAG=${1:-r15ar}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "conv1d or golden or bits or identical or ecapa or ring or saturates" > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_subset.log | cut -c1-300
for rep in 1 2 3; do
  for lib in conv_base product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_T=298 MV_BENCH_WARM=30 MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'])" | tee -a $OUT/bench_conv_ab.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3 4; do
  for lib in conv_base product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
