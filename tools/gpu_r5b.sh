#!/bin/bash
# Round-5 session B: the GPU suite again after r14a (torch CPU threads capped in conftest, Fbank bar as a distribution, big batches checked on a
# row subset), with per-test durations; the S16 peak / NaN layer tests and the campp_hot golden; the Fbank device fuzz under the new bar
TAG=${1:-r14b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 1200 python -m pytest tests -q -m gpu --durations=40 --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -60 $OUT/pytest_gpu.log | cut -c1-220
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
MV_FUZZ_STREAM=1 timeout 400 python tools/emu_fuzz.py fbank 200 --device gpu --jobs 4 --seed 3 > $OUT/fuzz_gpu_fbank.log 2>&1; echo "fuzz fbank rc=$?"; grep "RESULT\|FAIL" $OUT/fuzz_gpu_fbank.log | head -20
