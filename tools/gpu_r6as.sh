#!/bin/bash
# Round-6 session AS: timing probe (wrong results on purpose): the ring GEMM's epilogue as ONE v_med3 per value -- what folding the BatchNorm scale into the
# weights and the shift into the bias would leave (y = relu(z) s + t = med3(z s + t, t, +-inf)) -- against the product's v_max + v_pk_fma; headline alternating
TAG=${1:-r15as}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3 4; do
  for lib in product ring_fold_probe; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
