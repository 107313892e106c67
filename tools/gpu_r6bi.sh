#!/bin/bash
# Round-6 session BI: ASP hidden conv, the fused input statistics' partial rows through an LDS buffer (one 16-byte store per thread every four stages, in front of a
# stage's transfers) instead of four stores per wave in the middle of every stage; conv_prev = conv1d.hip@HEAD
TAG=${1:-r15bi}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "stat or asp or ecapa or conv1d" 2>&1 | tail -2 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
for lib in conv_prev product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    echo "== $lib" | tee -a $OUT/asp_hidden.log
    timeout 300 python tools/bench_asp_hidden.py 2>&1 | grep "^{" | grep '"y": "f16"' | tee -a $OUT/asp_hidden.log
done
done
unset MV_PROBE_LIB
timeout 300 python tools/bench_with_lib.py $P0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1   # (one untimed run first)
for lib in conv_prev product product conv_prev conv_prev product product conv_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
