#!/bin/bash
# Round-3 full session: all GPU tests, smoke, the default bench line (headline + other configs + batch-1 latency), single-model lines,
# rocprofv3 kernel stats of the headline and of CAM++.  usage: bash tools/gpu_r3_full.sh <tag> [skip-rocprof]
TAG=${1:-r07a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED" $OUT/pytest_gpu.log | tail -8
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== bench"; timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
grep "^{" $OUT/bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], 'fbank', j['roofline_fbank']['frac'], j['roofline_fbank']['avg_launch_us'], 'backbone', j.get('roofline_backbone'))
print('parity', j.get('parity'))
for k, v in j.get('other_configs', {}).items(): print(k, {a: v.get(a) for a in ('value', 'ms_per_step', 'ms_per_pass', 'parity', 'cosine_block', 'roofline', 'frontend_us', 'error', 'algorithmic_gflop_per_utt_conv2d')})
print('latency', j.get('latency_batch1'))
print('h2d', j.get('h2d_inclusive'))
"
for m in campp ecapa512 ecapa512_mel eres2netv2; do timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$m.log 2>&1; grep "^{" $OUT/bench_$m.log | cut -c1-330; done
if [ -z "$2" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
head -22 $OUT/prof/bench_kernel_stats.csv | cut -c1-170
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_campp -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof_campp.log 2>&1
head -14 $OUT/prof_campp/bench_kernel_stats.csv | cut -c1-170
fi
