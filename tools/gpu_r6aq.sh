#!/bin/bash
# Round-6 session AQ: the Res2Net chain's fp16 saturation through MODE.FP16_OVFL (one s_setreg per wave instead of a v_med3 per value and a packed min / max per
# pair: 480 instructions gone from the four instantiations): Res2Net / model tests of the GPU suite, the chain alone and the headline, alternating with res2_base
TAG=${1:-r15aq}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "res2 or golden or bits or identical or ecapa" > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_subset.log | cut -c1-300
for rep in 1 2 3; do
  for lib in res2_base product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_res2.py 2>/dev/null | grep "res2 chain" | sed "s/^/$lib $rep /" | tee -a $OUT/bench_res2_ab.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3 4; do
  for lib in res2_base product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
