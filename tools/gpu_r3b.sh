#!/bin/bash
# Round-3 session b: fused FCM block micro-benchmark, product vs timing probes (tools/probe_fcm.py).  usage: bash tools/gpu_r3b.sh <tag>
TAG=${1:-r06d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for rep in 1; do
timeout 300 python tools/bench_fcm.py 2>&1 | grep "fcm block" | tee -a $OUT/fcm_probes.log
for v in prodonly consonly nobarrier nomfma nolds nostore noload; do
  MV_PROBE_LIB=$REPO/tools/probe/libfcm_$v.so timeout 300 python tools/bench_fcm.py 2>&1 | grep "fcm block" | tee -a $OUT/fcm_probes.log
done
done
