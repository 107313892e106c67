#!/bin/bash
# Round-6 session BH: what the ASP hidden conv's fused input statistics cost -- timing probes (wrong statistics): the partial rows never stored / never reduced
TAG=${1:-r15bh}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
for lib in product hid_nostore hid_noreduce; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    echo "== $lib" | tee -a $OUT/asp_hidden_probes.log
    timeout 300 python tools/bench_asp_hidden.py 2>&1 | grep "^{" | grep '"y": "f16"' | tee -a $OUT/asp_hidden_probes.log
done
done
