#!/bin/bash
# Round-5 session ZD: conv2ds epilogue -- residual-only layers (every conv3) in batches of two segments with their own per-unit code (product) against the
# r14zb kernel (libc2ds_r14zb): the layers alone, ERes2NetV2 54.9 M 64 x 3 s, the m32 model 256 x 3 s, alternating; conv2ds / ERes2Net GPU tests first
TAG=${1:-r14zd}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 300 python -m pytest tests -q -m gpu --timeout 300 -k "eres2 or conv2ds or hot_head or stress" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log | cut -c1-200
for lib in product r14zb; do
  if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/libc2ds_r14zb.so; fi
  MV_BENCH_SHAPES="s1 conv1,s1 3x3,s1 conv3,s2 conv1,s2 3x3,s2 conv3,s3 conv1,s3 3x3" timeout 300 python tools/bench_conv2d.py 16 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['layer'], d['split_us'], d.get('split_GBps'))" | tee -a $OUT/bench_conv2d_ab.log
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2; do
  for lib in product r14zb; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libc2ds_r14zb.so; fi
    for cfg in "eres2netv2_w96s4 64 3" "eres2netv2 256 10"; do
      set -- $cfg
      timeout 300 python tools/bench_with_lib.py $P --model $1 --batch $2 --steps $3 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, '$1', d['value'], d['ms_per_step'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/eres2net_ab.log
    done
  done
done
