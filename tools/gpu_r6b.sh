#!/bin/bash
# Round-6 session B: ring GEMM carry, second form (every stage leaves its last MFMA group to the next one; two stage forms, 231 VGPRs, no spill -- the
# four-form build of r15a spilled 115-152 registers around the K loop and ran 40-70 % slower): base (24a38ec) / c1 (carry) / product (carry + K-half-1
# activation fragments one step early); ASP hidden conv with fused input statistics as a ring: base double buffer (2 workgroups per CU) / ns3 / product (ns4,
# one workgroup per CU); the layers alone and the headline, alternating in one call; full GPU suite first.
TAG=${1:-r15b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
for rep in 1 2; do
  for lib in ring_base ring_c1 product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    MV_BENCH_TILES=256 MV_BENCH_SHAPES="c2c 1024,mfa 3072,c2c 512,mfa 1536" timeout 300 python tools/bench_conv.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d['shape'], d['us'], d['TFLOPs'])" | tee -a $OUT/bench_conv_ab.log
  done
  for lib in ring_base asp_ns2 asp_ns3 product; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_asp_hidden.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', $rep, d)" | tee -a $OUT/bench_asp_hidden_ab.log
  done
done
unset MV_PROBE_LIB
P0=$REPO/voiceprintrecognition-pytorch_amd/mvector/lib/libmvector_hip.so
for rep in 1 2 3; do
  for lib in ring_base ring_c1 asp_ns2 product; do
    if [ $lib = product ]; then P=$P0; else P=$REPO/tools/probe/lib$lib.so; fi
    timeout 300 python tools/bench_with_lib.py $P --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $rep, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['stage_ms'], d.get('parity', {}).get('max_one_minus_cos'), d['box']['mfma_f16_tflops'], d['box']['copy_gbs'])" | tee -a $OUT/headline_ab.log
  done
done
