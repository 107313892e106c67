#!/bin/bash
# Round-5 session O: two more finds of the ISA audit.  (1) conv2ds_kernel's epilogue operands (residual / AFF / second-output loads under `if (has1)`, read
# under another `if (has1)`): each load of a batch sat behind its own s_waitcnt vmcnt(0) -- six round trips per batch; used on every path now.
# (2) linear_f32_splitk_kernel: the per-lane choice between 16-byte and guarded loads ran both forms behind vmcnt(0) waits -- two copies of the loop now.
# Product against libc2ds_base / liblin_base, alternating in one call: ERes2NetV2 (54.9 M) 64 x 3 s and ERes2NetV2-m32 256 x 3 s; the linears alone,
# one utterance and the headline.  The ERes2Net / linear GPU tests first.
TAG=${1:-r14o}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests -q -m gpu --timeout 400 -k "eres2 or conv2ds or linear or batch_size or golden" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log | cut -c1-200
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2; do
  for lib in product c2ds_base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libc2ds_base.so; fi
    for cfg in "eres2netv2_w96s4 64 3" "eres2netv2 256 10"; do
      set -- $cfg
      timeout 300 python tools/bench_with_lib.py $P --model $1 --batch $2 --steps $3 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, '$1', d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/eres2net_conv2ds_epilogue_ab.log
    done
  done
done
for lib in product lin_base product lin_base; do
  if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/liblin_base.so; fi
  echo "== $lib" >> $OUT/bench_linear_ab.log
  timeout 200 python tools/bench_linear.py 2>/dev/null | grep -v "^$" >> $OUT/bench_linear_ab.log
done
unset MV_PROBE_LIB
tail -14 $OUT/bench_linear_ab.log | cut -c1-200
cat > /tmp/lat.py <<PY
import sys, json, ctypes
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
r=bench.latency_batch1('ecapa1024', torch.device('cuda',0))
print(json.dumps(dict(lib=lib.split('/')[-1], eager_p50=r['eager_p50'], gpu_us=r['gpu_us_back_to_back'], graph_p50=r['hipgraph_p50'])))
PY
for rep in 1 2; do
  for lib in product $REPO/tools/probe/liblin_base.so; do
    timeout 300 python /tmp/lat.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/latency_batch1_linear_ab.log
  done
done
for rep in 1 2 3; do
  for lib in product lin_base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/liblin_base.so; fi
    timeout 300 python tools/bench_with_lib.py $P --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'])" | tee -a $OUT/headline_linear_ab.log
  done
done
