#!/bin/bash
# Round-5 session T: conv1d_glds_persistent_kernel<false, 0> (EcapaTdnn block 0: k = 5 on 80 mel bins) without its pointer table in scratch memory --
# the kernel's average under rocprofv3, product against libconv1d_base, and the headline alternating
TAG=${1:-r14t}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 300 python -m pytest tests -q -m gpu --timeout 300 -k "conv1d or golden or batch_size" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log | cut -c1-200
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
  for lib in product conv1d_base; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libconv1d_base.so; fi
    timeout 300 python tools/bench_with_lib.py $P --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['stage_ms']['backbone'])" | tee -a $OUT/headline_conv1d_block0_ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in product conv1d_base; do
  if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/libconv1d_base.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$lib -o bench -- python $REPO/tools/bench_with_lib.py $P --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_$lib.log 2>&1
  f=$(find $OUT/prof_$lib -name "*kernel_stats.csv" | head -1); grep "glds_persistent" "$f" | cut -c1-160 | sed "s/^/$lib /" | tee -a $OUT/block0_kernel_avg.log; rm -rf $OUT/prof_$lib
done
