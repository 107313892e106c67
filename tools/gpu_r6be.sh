#!/bin/bash
# Round-6 session BE: the tree after the Res2Net chain's straight-line fragment requests: device fuzz of the res2 and model families, then the driver's commands
# (full GPU suite, smoke, python bench.py)
TAG=${1:-r15be}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 1200 python tools/emu_fuzz.py --device gpu --seed 731 --jobs 4 res2 400 > $OUT/fuzz_res2.log 2>&1; echo "fuzz res2 rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_res2.log | cut -c1-260
timeout 1200 python tools/emu_fuzz.py --device gpu --seed 731 --jobs 4 model 120 > $OUT/fuzz_model.log 2>&1; echo "fuzz model rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_model.log | cut -c1-260
bash tools/gpu_r6g.sh $TAG
