#!/bin/bash
# Round-6 session AB: device fuzz of every kernel family on the tree as it is left (seeded random launch geometries against the layer checks' references)
TAG=${1:-r15ab}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 2400 python tools/emu_fuzz.py --device gpu --seed 626 --jobs 4 all 250 > $OUT/fuzz_all.log 2>&1; echo "rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_all.log | cut -c1-260
