#!/bin/bash
# Round-4 session D: vector-instruction issue rates (the Fbank floor), all-core CPU baseline through the bench line
TAG=${1:-r12d}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/valu_rate_probe.hip && timeout 300 /tmp/valu_rate 2>&1 | tee $OUT/valu_rate_probe.log
echo "== bench"; timeout 1500 python bench.py --steps 20 --warmup 5 --no-other-configs > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep "^{" $OUT/bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], 'fbank', j['roofline_fbank']['frac'], j['roofline_fbank']['avg_launch_us'])
for k in ('cpu_baseline', 'cpu_baseline_as_shipped', 'cpu_baseline_all_cores'): print(k, j.get(k))
"
