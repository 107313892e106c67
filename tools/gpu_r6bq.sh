#!/bin/bash
# Round-6 session BQ (third session): device fuzz of every kernel family on the tree as it is left (the second session changed the ring GEMM's epilogue, the Res2Net chain's
# requests, time_stats and the ASP hidden conv's statistics rows), then the driver's commands (full GPU suite, smoke, python bench.py)
TAG=${1:-r15bq}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 1500 python tools/emu_fuzz.py --device gpu --seed 1030 --jobs 4 all 250 > $OUT/fuzz_all.log 2>&1; echo "fuzz rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_all.log | cut -c1-260
bash tools/gpu_r6g.sh $TAG
