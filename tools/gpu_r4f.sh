#!/bin/bash
# Round-4 session F: all GPU tests (batch-size invariance of embeddings on four backbones), ring GEMM K stage on 32x32x16 MFMAs (timing probe) beside
# the product, default bench line
TAG=${1:-r12f}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED|^E  " $OUT/pytest_gpu.log | tail -12
echo "== ring GEMM: 16x16x32 (product) vs 32x32x16 (timing probe)"
for rep in 1 2; do
  timeout 300 python tools/bench_conv.py 2>&1 | grep "tile\": 256" | grep "c2c 1024\|mfa 3072" | sed 's/^/product  /' | tee -a $OUT/gemm_mfma32_ab.log
  MV_PROBE_LIB=tools/probe/libconv_mfma32.so timeout 300 python tools/bench_conv.py 2>&1 | grep "tile\": 256" | grep "c2c 1024\|mfa 3072" | sed 's/^/mfma32   /' | tee -a $OUT/gemm_mfma32_ab.log
done
echo "== bench"; timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
grep "^{" $OUT/bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], 'fbank', j['roofline_fbank']['frac'], j['roofline_fbank']['avg_launch_us'])
print('parity', j.get('parity'))
for k, v in j.get('other_configs', {}).items(): print(k, {a: v.get(a) for a in ('value', 'ms_per_step', 'ms_per_pass', 'parity', 'fcm_head', 'error')})
print('latency', j.get('latency_batch1'))
print('two_streams', j.get('two_streams'))
"
