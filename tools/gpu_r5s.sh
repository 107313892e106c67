#!/bin/bash
# Round-5 session S: the fuzzer's generators against the product library on the device for the families whose kernels changed since r14c (whole models incl.
# CAM++, conv2ds, linear, fcm_block), then the utterance-length sweep of CAM++ (tools/bench_long.py)
TAG=${1:-r14s}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for fam in "model 40" "conv2ds 150" "linear 150" "fcm_block 40"; do
  set -- $fam
  timeout 240 python tools/emu_fuzz.py $1 $2 --device gpu --jobs 1 --seed 5 > $OUT/fuzz_gpu_$1.log 2>&1; echo "fuzz $1 rc=$?"; tail -2 $OUT/fuzz_gpu_$1.log | cut -c1-200
done
timeout 300 python tools/bench_long.py campp > $OUT/bench_long_campp.log 2>&1; grep -v "^$\|INFO\|amdgpu" $OUT/bench_long_campp.log | tail -8
