#!/bin/bash
# Round-6 session BR (third session): (1) diagnosis of the one device-fuzz failure of r15bq (fcm_c1 at B = 64, F = 80, T = 998: 4.09e-3 against a 4e-3 bar -- wrong value
# or the metric's tail over 81.7 M outputs?), (2) PMC passes + rocprofv3 --kernel-trace --stats of the bench command on the tree as it is left (tools/gpu_r6i.sh)
TAG=${1:-r15br}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 1200 python tools/diag_fcm_c1_tail.py 526 0 1 2 3 4 5 6 7 8 9 10 > $OUT/diag_fcm_c1_tail.log 2>&1; echo "diag rc=$?"; cat $OUT/diag_fcm_c1_tail.log | cut -c1-260
bash tools/gpu_r6i.sh $TAG
