#!/bin/bash
# Round-4 session AJ: all GPU tests on the final tree (new golden campp_c64, layer checks with zeroed operand padding) + the geometry fuzzer of
# tools/emu_fuzz.py against the product library on the device (random launch geometries incl. full-size maps / long utterances / chip-filling batches)
TAG=${1:-r13c}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 150 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
run() { MV_FUZZ_STREAM=1 timeout 100 python tools/emu_fuzz.py $1 $2 --device gpu --jobs $3 --seed 21 > $OUT/fuzz_$1.log 2>&1 & }
run conv2ds 160 4; run conv1d 160 4; run conv2d 60 2; run res2 40 2; run asp_pool 60 2; run time_stats 60 1; run linear 60 1; run fbank 40 2; run melspec 60 2; run fcm_block 40 2
wait
cat $OUT/fuzz_*.log | grep -E " ok, |FAIL|CRASH" | cut -c1-400
