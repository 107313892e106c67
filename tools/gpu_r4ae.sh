#!/bin/bash
# Round-4 session AE: race detection on the split conv path (same batch N times, another batch in between, two streams) + the parity tests on the final library
TAG=${1:-r12ae}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
MV_STRESS_MODELS=eres2netv2,eres2net,eres2netv2_w96s4,campp_f32 timeout 500 python tools/stress_determinism.py 30 > $OUT/stress_determinism.log 2>&1; echo "stress rc=$?"; grep "^{" $OUT/stress_determinism.log | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv2ds or eres2net or campp or hipgraph or batch_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
