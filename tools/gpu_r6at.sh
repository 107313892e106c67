#!/bin/bash
# Round-6 session AT: BatchNorm folded into the operands of EcapaTdnn's dense 1x1 TDNN blocks (tdnn1 / tdnn2 / MFA: W' = s W, bias' = s b + t) and the conv
# epilogue as one median per value (MvConv1dDesc.fold_floor / fold_limit): full GPU suite, then the headline alternating with fold_base (the previous commit's
# whole library, built in a worktree)
TAG=${1:-r15at}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-300
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3 4; do
  for lib in fold_base product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'], d['box'].get('ring_k3072_us'))" | tee -a $OUT/headline_ab.log
  done
done
for m in ecapa512; do
  for lib in fold_base product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs --model $m 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', '$m', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/headline_ab.log
  done
done
