#!/bin/bash
# Round-6 session AU: the ring epilogue's MODE.FP16_OVFL saturation once more, in ABBA order (r15at showed the box warming up over the first runs of a session, which
# favours the arm that runs second in every pair): conv_base = conv1d.hip before the change, everything else equal
TAG=${1:-r15au}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
timeout 300 python tools/bench_with_lib.py $P0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1   # (one untimed run first)
for lib in conv_base product product conv_base conv_base product product conv_base conv_base product product conv_base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
