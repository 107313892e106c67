#!/bin/bash
# Round-6 session BA: Res2Net chain with every second workgroup of an XCD started late (the late K stages sit behind the
# chip-wide y-store burst of the previous epilogue; all 256 workgroups run in lockstep, so the bursts of a whole XCD arrive at its L2 at once).
# Arms: product, odd workgroups ((blockIdx >> 3) & 1) late by ~1 / 2 / 4 / 6 us, four phases of ~2 us.  Micro-benchmark alternating twice, then headline A/B of the best.
TAG=${1:-r15ba}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2; do
for lib in product late1 late2 late4 late6 q4x15; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/libres2_$lib.so; fi
    echo "== $lib" | tee -a $OUT/res2_stagger_micro.log
    timeout 300 python tools/bench_res2.py 2>&1 | grep "res2 chain" | tee -a $OUT/res2_stagger_micro.log
done
done
unset MV_PROBE_LIB
timeout 300 python tools/bench_with_lib.py $P0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1   # (one untimed run first)
for lib in product late2 late4 late4 late2 product product late2 late4; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libres2_$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
done
