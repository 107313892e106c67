#!/bin/bash
# Round-6 session N: knob sweeps.  (1) ring GEMM: where a stage issues its eight LDS-DMA transfers (product: two per step over steps 0-3; dma8: one per step over all eight;
# dma2: four at once in steps 0 and 1; dmalate: steps 2-5).  (2) se_gate_residual_kernel: two elements per thread in flight, grids of 2048 / 8192 workgroups (product 4096):
# per-kernel averages of rocprofv3 --stats under the headline bench
TAG=${1:-r15n}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for lib in; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_CLOCK=1 MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'], d.get('clock_ghz'))" | tee -a $OUT/ring_dma_placement.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for lib in product gate_g8k gate_g16384 gate_g65536; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$lib -o k -- python $REPO/tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs --no-box > $OUT/b.log 2>&1
    f=$(find $OUT/prof_$lib -name "*kernel_stats.csv" | head -1)
    python - <<PY | tee -a $OUT/se_gate_variants.log
import csv, json
rows = {r['Name']: r for r in csv.DictReader(open('$f'))}
g = [r for n, r in rows.items() if 'se_gate_residual' in n][0]
d = json.loads([l for l in open('$OUT/b.log') if l.startswith('{')][-1])
print('$lib', $rep, 'se_gate avg us', round(float(g['AverageNs']) / 1e3, 2), 'calls', g['Calls'], 'headline', d['value'])
PY
    rm -rf $OUT/prof_$lib
  done
done
rm -f $OUT/b.log
