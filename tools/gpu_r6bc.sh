#!/bin/bash
# Round-6 session BC: Res2Net chain, three steps on top of res2.hip@HEAD (base): U = weight-fragment requests unconditional + six written-out stages (r15bb),
# A = U + the epilogue in two passes (all y stores first, then the LDS part), AB = A + bias / scale / shift requested in front of the last stage's last ten MFMAs
TAG=${1:-r15bc}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "res2 or ecapa or bit or batch" 2>&1 | tail -5 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
for lib in base U A AB; do
    export MV_PROBE_LIB=$REPO/tools/probe/libres2_$lib.so
    echo "== $lib" | tee -a $OUT/res2_micro.log
    timeout 300 python tools/bench_res2.py 2>&1 | grep "res2 chain" | tee -a $OUT/res2_micro.log
done
done
unset MV_PROBE_LIB
timeout 300 python tools/bench_with_lib.py $P0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1   # (one untimed run first)
for lib in base AB AB base base AB AB base; do
    P=$REPO/tools/probe/libres2_$lib.so
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
