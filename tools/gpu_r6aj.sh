#!/bin/bash
# Round-6 session AJ: device fuzz of the conv1d family after the tile-id change (persist_blocks_hint and >= 8 K stages now in the generator: multi-round walks, sub-tile tails)
TAG=${1:-r15aj}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 2400 python tools/emu_fuzz.py --device gpu --seed 727 --jobs 4 conv1d 900 > $OUT/fuzz_conv1d.log 2>&1; echo "rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_conv1d.log | cut -c1-300 | tail -20
