#!/bin/bash
# Round-6 session BT (third session): a second, larger device fuzz on the tree as it is left (another seed, 600 cases per family; conv1d 1200: its generator has the
# multi-round persistent walks and sub-tile tails, and the second session changed the ring GEMM's store path), with the size-aware fcm_c1 bar of r15br
TAG=${1:-r15bt}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 2400 python tools/emu_fuzz.py --device gpu --seed 2040 --jobs 4 all 600 > $OUT/fuzz_all_600.log 2>&1; echo "fuzz all rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_all_600.log | cut -c1-300
timeout 1200 python tools/emu_fuzz.py --device gpu --seed 2041 --jobs 4 conv1d,res2,asp_pool,time_stats 1200 > $OUT/fuzz_touched_1200.log 2>&1; echo "fuzz touched rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_touched_1200.log | cut -c1-300
