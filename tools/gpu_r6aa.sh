#!/bin/bash
# Round-6 session AA: timing probes of the Res2Net chain (wrong results on purpose): without the y stores / without the whole epilogue
TAG=${1:-r15aa}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
  for lib in product res2_probe_nostore res2_probe_noepi; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 200 python tools/bench_res2.py 2>/dev/null | grep "^res2" | sed "s/^/$lib $rep /" | tee -a $OUT/res2_probes.log
  done
done
