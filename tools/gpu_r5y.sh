#!/bin/bash
# Round-5 session Y: every family of the fuzzer on the device once more, 300 cases each, another seed (after r14x's fix of the conv1d reference)
TAG=${1:-r14y}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
MV_FUZZ_STREAM=0 timeout 1500 python tools/emu_fuzz.py all 300 --device gpu --jobs 1 --seed 13 > $OUT/fuzz_gpu_all_300.log 2>&1; echo "fuzz rc=$?"; grep "^RESULT\|FAIL\|^plain" $OUT/fuzz_gpu_all_300.log | cut -c1-220
