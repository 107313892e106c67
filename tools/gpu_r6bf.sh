#!/bin/bash
# Round-6 session BF: ring GEMM epilogue with the row check per wave (straight-line stores for waves whose 64 time steps are all valid) against conv1d.hip@HEAD
TAG=${1:-r15bf}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "conv1d or ecapa or ring or tail" 2>&1 | tail -3 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
for lib in base product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/libconv_$lib.so; fi
    echo "== $lib" | tee -a $OUT/conv_micro.log
    MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | tail -4 | tee -a $OUT/conv_micro.log
done
done
unset MV_PROBE_LIB
timeout 300 python tools/bench_with_lib.py $P0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1   # (one untimed run first)
for lib in base product product base base product product base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libconv_$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
