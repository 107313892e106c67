#!/bin/bash
# Round-6 session G: the driver's commands on the current tree: full GPU suite, smoke, python bench.py
TAG=${1:-r15g}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-200
timeout 1200 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d = json.loads([l for l in open('$OUT/bench.log') if l.startswith('{')][-1])
rc = d.get('roofline_conv1d_class', {})
print('ring', d['roofline']['frac'], d['roofline']['launches'], d['roofline']['avg_launch_us'], d['roofline']['share_of_step'], 'class', rc.get('frac'), rc.get('launches'), rc.get('share_of_step'))
print('headline', d['value'], d['ms_per_step'], 'conv frac', d['roofline']['frac'], 'fbank', d['roofline_fbank']['avg_launch_us'], d['roofline_fbank']['frac'])
print('box', {k: v for k, v in d['box'].items() if k != 'note'})
for k, v in d.get('other_configs', {}).items():
    print(k, v.get('value'), v.get('repeats'), v.get('value_one_stream'), v.get('parity', {}).get('max_one_minus_cos'), v.get('error'))
print('lat', {k: (v.get('eager_p50'), v.get('gpu_us_back_to_back'), v.get('hipgraph_p50')) for k, v in d.get('latency_batch1', {}).items()})
print('two_streams', d.get('two_streams', {}).get('value'), 'cpu', d['cpu_baseline']['value'], d.get('cpu_baseline_all_cores', {}).get('value'), 'parity', d['parity'])
PY
