#!/bin/bash
# Round-6 session BW (third session): the remaining families at 1200 cases each on a fourth seed (conv2ds, fcm_block, fcm_conv, model, linear, window, small), fbank / melspec once more
TAG=${1:-r15bw}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 3000 python tools/emu_fuzz.py --device gpu --seed 4060 --jobs 4 conv2ds,fcm_block,fcm_conv,model,linear,window,small,fbank,melspec 1200 > $OUT/fuzz_rest_1200.log 2>&1; echo "fuzz rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_rest_1200.log | cut -c1-400
