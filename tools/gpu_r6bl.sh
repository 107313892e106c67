#!/bin/bash
# Round-6 session BL: (1) the double-buffer GEMM kernel had SPILLED behind the epilogue change of r15bf (block 0 of the backbone 88-92 -> 101-104 us): the per-wave
# store path is the ring kernel's only now -- block 0 and the headline, conv_spill (conv1d.hip@HEAD) against the tree; (2) conv2ds epilogue with the per-wave all-ok
# path (plain and residual-only layers, <= 2 blocks per wave) against conv2ds.hip@HEAD on the ERes2Net family
TAG=${1:-r15bl}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "conv1d or ecapa or ring or tail or conv2ds or eres2net" 2>&1 | tail -2 | tee $OUT/pytest_subset_tail.log
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for lib in conv2ds_prev product product conv2ds_prev; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    for m in eres2netv2_w96s4:64 eres2netv2:256 eres2net:256; do
        timeout 600 python tools/bench_with_lib.py $P --model ${m%%:*} --batch ${m##*:} --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-box 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', '$m', d['value'], d['ms_per_step'], d.get('parity', {}).get('max_one_minus_cos'))" | tee -a $OUT/eres2net_ab.log
    done
done
cd /tmp && export TMPDIR=/tmp
for lib in conv_spill product product conv_spill; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    rm -rf $OUT/prof
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python $REPO/tools/bench_with_lib.py $P --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_$lib.log 2>&1
    f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
    python - <<PY | tee -a $OUT/in_situ.log
import csv, statistics, json
rows = [r for r in csv.DictReader(open('$f'))]
by = {}
for r in rows:
    by.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
line = [l for l in open('$OUT/bench_$lib.log') if l.startswith('{')]
v = json.loads(line[0])['value'] if line else None
def med(key, n):
    x = [v_ for k, v_ in by.items() if key in k]
    return statistics.median(x[0][-n:]) if x else float('nan')
print('%-10s headline %s  block0 (double-buffer GEMM) in situ median %.1f us  ring %.1f  asp hidden %.1f' % ('$lib', v, med('glds_persistent', 30), med('ring_persistent', 210), med('conv1d_glds_kernel<2, 2, 4, 5, true', 30)))
PY
done
rm -rf $OUT/prof
cd $REPO
for lib in conv_spill product product conv_spill conv_spill product product conv_spill; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], 'ring', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_abba.log
done
