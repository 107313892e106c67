#!/bin/bash
# Round-4 session B: all GPU tests after the knob retirement + the CAM++ head probes, finer Fbank phase probes, batch-1 kernel breakdown
# (rocprofv3 kernel stats of B = 1 forwards), the default bench line.
TAG=${1:-r12b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED|fbank 256|campp_mid|campp_stress|campp:|campp_short:" $OUT/pytest_gpu.log | tail -20
echo "== fbank probes"
for v in base noloop notrans nopost nowin notw nolog nopower base; do
  MV_PROBE_LIB=tools/probe/libfbankp_$v.so timeout 300 python tools/bench_fbank.py 2>&1 | grep "^{" | cut -c70-200 | tee -a $OUT/fbank_probes.log
done
echo "== bench"; timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
grep "^{" $OUT/bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], 'fbank', j['roofline_fbank']['frac'], j['roofline_fbank']['avg_launch_us'], 'backbone', j.get('roofline_backbone'))
print('parity', j.get('parity'))
for k, v in j.get('other_configs', {}).items(): print(k, {a: v.get(a) for a in ('value', 'ms_per_step', 'ms_per_pass', 'parity', 'fcm_head', 'frontend_us', 'error')})
print('latency', j.get('latency_batch1'))
print('two_streams', j.get('two_streams'))
print('cpu', j.get('cpu_baseline'))
"
echo "== batch-1 kernel breakdown"
cd /tmp && export TMPDIR=/tmp
for m in ecapa1024 campp; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1_$m -o b1 -- python $REPO/tools/bench_latency.py $m 1 50 > $OUT/b1_$m.log 2>&1
  grep "GPU time" $OUT/b1_$m.log
  find $OUT/prof_b1_$m -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "head -24 {} | cut -c1-150"
done
