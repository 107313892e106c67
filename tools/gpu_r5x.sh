#!/bin/bash
# Round-5 session X: every family of the fuzzer against the product library of the final tree on the device (seed 9); second call: the two conv1d cases the
# first call failed (in_affine with fp32 output: the test's reference evaluated x * s + t with two roundings, the kernel's fma with one -- an fp16
# rounding boundary crossed for one element in ~30 000) replayed, then conv1d again with two seeds
TAG=${1:-r14x}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
if [ "$2" = "replay" ]; then
  timeout 200 python tools/emu_fuzz.py conv1d --device gpu --replay "dict(B=130, T=129, cin=64, cout=256, k=1, dil=1, seed=778, y_f32=True, in_affine=True, pre_act=0)" 2>&1 | tail -1 | tee -a $OUT/fuzz_gpu_conv1d_replay.log
  timeout 200 python tools/emu_fuzz.py conv1d --device gpu --replay "dict(B=130, T=298, cin=64, cout=128, k=1, dil=1, seed=875, y_f32=True, in_affine=True, pre_act=0, row_bias=True, post_act=2)" 2>&1 | tail -1 | tee -a $OUT/fuzz_gpu_conv1d_replay.log
  for seed in 9 10 11; do
    timeout 600 python tools/emu_fuzz.py conv1d 300 --device gpu --jobs 1 --seed $seed 2>&1 | grep "RESULT\|FAIL\|conv1d " | tail -4 | tee -a $OUT/fuzz_gpu_conv1d_replay.log
  done
  exit 0
fi
MV_FUZZ_STREAM=0 timeout 1200 python tools/emu_fuzz.py all 150 --device gpu --jobs 1 --seed 9 > $OUT/fuzz_gpu_all.log 2>&1; echo "fuzz rc=$?"; grep "^RESULT\|FAIL" $OUT/fuzz_gpu_all.log | cut -c1-200
