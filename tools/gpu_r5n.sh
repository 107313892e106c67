#!/bin/bash
# Round-5 session N: fcm_block_kernel's C1 prologue (all feature loads of the window and of the prologue rows requested before the first is used) against
# the previous kernel (libfcm_base), CAM++ 256 x 3 s alternating in one call; the full GPU suite first; rocprofv3 kernel stats of the CAM++ leg last
TAG=${1:-r14n}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
for rep in 1 2 3; do
  for lib in product fcm_base; do
    case $lib in
      product) P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so;;
      fcm_base) P=$REPO/tools/probe/libfcm_base.so;;
    esac
    timeout 300 python tools/bench_with_lib.py $P --model campp --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/campp_fcm_prologue_ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_campp -o campp -- python $REPO/bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_campp_rocprof.log 2>&1
find $OUT/prof_campp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprofv3_kernel_stats_campp.csv
rm -rf $OUT/prof_campp
head -12 $OUT/rocprofv3_kernel_stats_campp.csv | cut -c1-150
