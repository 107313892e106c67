#!/bin/bash
# Round-6 session BS (third session): the race detector on the tree as it is left (tools/stress_determinism.py: the same batch 30 times with another batch in between, and the
# two-stream arm -- this round rewrote counted waits in the ring GEMM, the Res2Net chain, the ASP kernels and time_stats; last run on the device in round 4), then the
# utterance-length sweep of README / DESIGN section 6 re-measured on this tree (tools/bench_long.py; last taken in round 3 + 5)
TAG=${1:-r15bs}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
MV_STRESS_MODELS=ecapa1024,ecapa512,campp,campp_f32,ecapa512_mel,eres2netv2,eres2net timeout 1500 python tools/stress_determinism.py 30 > $OUT/stress_determinism.log 2> $OUT/stress_determinism.err; echo "stress rc=$?"; grep -v INFO $OUT/stress_determinism.log | cut -c1-240
timeout 900 python tools/bench_long.py campp ecapa1024 > $OUT/bench_long.log 2> $OUT/bench_long.err; echo "bench_long rc=$?"; grep -v INFO $OUT/bench_long.log | cut -c1-200
