#!/bin/bash
# Round-6 session J: asp_pool_ring_kernel: ring depth 6 (product) / 4 (48 KiB: three workgroups per CU) / 5 / 8 (96 KiB: one), and the whole tiles' zero start as the
# MFMA's inline-constant C operand (16 fewer vector moves per step and wave); the kernel alone, alternating
TAG=${1:-r15j}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for lib in product asp_r4 asp_r5 asp_r8 asp_zc; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 120 python tools/bench_asp.py 2>/dev/null | grep "^{" | grep nomax | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', $rep, d['us'], d['x_GBps'])" | tee -a $OUT/bench_asp_pool_ab.log
  done
done
