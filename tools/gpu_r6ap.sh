#!/bin/bash
# Round-6 session AP: does gfx950 honour MODE.FP16_OVFL?  (tools/fp16_ovfl_probe.hip)
TAG=${1:-r15ap}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/fp16_ovfl_probe.hip -o /tmp/fp16_ovfl_probe > /dev/null 2>&1 && timeout 60 /tmp/fp16_ovfl_probe > $OUT/fp16_ovfl_probe.log 2>&1
cat $OUT/fp16_ovfl_probe.log
