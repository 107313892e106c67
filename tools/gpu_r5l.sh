#!/bin/bash
# Round-5 session L: in-kernel timeline of cam_dense_block_kernel after the r14k changes (tools/probe_camblock.py: s_memtime of one wave at the phase boundaries;
# waves 0, 5 and 7 of workgroup 100)
TAG=${1:-r14l}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for t in 0 320 448; do
  MV_PROBE_TID=$t timeout 300 python tools/probe_camblock.py run > $OUT/cam_dense_block_inkernel_timeline_t$t.log 2>&1; grep -A30 "^cam_dense" $OUT/cam_dense_block_inkernel_timeline_t$t.log | grep "mean\|events"
done
