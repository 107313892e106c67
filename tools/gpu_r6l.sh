#!/bin/bash
# Round-6 session L: device fuzz of the kernel families this round touched (conv1d: the carried MFMA group of the ring kernel; asp_pool: four-tile rings) and of the
# whole models, seeded random launch geometries against the fp32 / fp64 evaluations of the layer tests
TAG=${1:-r15l}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for fam in conv1d asp_pool model fbank; do
  timeout 900 python tools/emu_fuzz.py --device gpu --seed 615 --jobs 4 $fam 300 > $OUT/fuzz_$fam.log 2>&1; echo "$fam rc=$?"; tail -3 $OUT/fuzz_$fam.log | cut -c1-300
done
