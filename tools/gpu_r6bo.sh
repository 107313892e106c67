#!/bin/bash
# Round-6 session BO: linear_f32_kernel with cascading trips (16 / 8 / 4 / 2 / 1 blocks per load round trip: K = 1024 at four waves is one trip instead of four)
# against linear.hip@HEAD: GPU tests, bit-identity of the two libraries on the bench batch, in situ per-dispatch medians (Ecapa headline and CAM++), headline ABBA
TAG=${1:-r15bo}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "linear or cosine or ecapa or campp or tdnn or bit or batch or gallery or predictor" 2>&1 | tail -2 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
cd /tmp && export TMPDIR=/tmp
for spec in linear_prev:ecapa1024 product:ecapa1024 linear_prev:campp product:campp product:ecapa1024 linear_prev:ecapa1024; do
    lib=${spec%%:*}; model=${spec##*:}
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    rm -rf $OUT/prof
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/tools/bench_with_lib.py $P --model $model --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_${lib}_$model.log 2>&1
    f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
    python - <<PY | tee -a $OUT/in_situ.log
import csv, statistics, json
rows = [r for r in csv.DictReader(open('$f'))]
by = {}
for r in rows:
    by.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
line = [l for l in open('$OUT/bench_${lib}_$model.log') if l.startswith('{')]
d = json.loads(line[0]) if line else {}
x = [v_ for k, v_ in by.items() if 'linear_f32_kernel' in k]
lin = x[0][-240:] if x else [float('nan')]
print('%-12s %-10s value %s  parity %s  linear_f32_kernel in situ: median %.2f us, sum per step %.1f us (n %d)' % ('$lib', '$model', d.get('value'), d.get('parity', {}).get('max_one_minus_cos'), statistics.median(lin), sum(lin) / 40 if len(lin) >= 240 else float('nan'), len(lin)))
PY
done
rm -rf $OUT/prof
cd $REPO
for lib in linear_prev product product linear_prev linear_prev product product linear_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
for lib in linear_prev product product linear_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --model campp --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib campp', d['value'], d['ms_per_step'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/campp_ab.log
done
