#!/bin/bash
# Round-5 session M: cam_dense_block_kernel, second pass over the tail -- the next layer's context parameters requested behind the k = 3 phase and left in
# flight across the layer entry, the k = 3 phase's fragment reads in six pipelined groups, row sums as explicit v_add_f32_dpp chains, one clamp for
# ReLU + saturation -- against the r14k kernel (libcb_r14k) and against -DMV_CB_LATE_PARAMS=0; CAM++ tests first, the in-kernel timeline last
TAG=${1:-r14m}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests -q -m gpu --timeout 400 -k "campp or long_and_short or batch_size" > $OUT/pytest_campp.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_campp.log | cut -c1-200
for rep in 1 2 3; do
  for lib in product r14k late0; do
    case $lib in
      product) P=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so;;
      r14k) P=$REPO/tools/probe/libcb_r14k.so;;
      late0) P=$REPO/tools/probe/libcb_late0.so;;
    esac
    timeout 300 python tools/bench_with_lib.py $P --model campp --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/campp_tail_ab.log
  done
done
for t in 0 448; do
  MV_PROBE_TID=$t timeout 300 python tools/probe_camblock.py run > $OUT/cam_dense_block_inkernel_timeline_t$t.log 2>&1; grep -A30 "^cam_dense" $OUT/cam_dense_block_inkernel_timeline_t$t.log | grep "mean\|events"
done
