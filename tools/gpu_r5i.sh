#!/bin/bash
# Round-5 session I: long-K linears as K slices (linear_f32_splitk_kernel + slice-ordered finish) against the direct kernel: layer tests, the two
# layers alone, and the headline alternating product / -DMV_LINEAR_NO_SPLITK in one call
TAG=${1:-r14i}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "linear or native_model or bits_do_not or eres2net_matches or full_batch or repeated" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log
timeout 200 python tools/bench_linear.py 2>/dev/null | grep "^linear" | tee $OUT/bench_linear_direct_vs_splitk.log
for rep in 1 2 3; do
  for lib in product nosplitk; do
    if [ $lib = product ]; then P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so; else P=$REPO/tools/probe/liblinear_nosplitk.so; fi
    timeout 300 python tools/bench_with_lib.py $P --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms'], d['parity'])" | tee -a $OUT/headline_splitk_ab.log
  done
done
