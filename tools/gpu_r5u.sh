#!/bin/bash
# Round-5 session U: the 64 x 64 conv kernel of small batches as a ring of four K stages (three in flight, counted waits) against the double-buffer form
# (libconv1d_ns2 = conv1d.hip of the previous commit): one utterance through EcapaTdnn-1024 and CAM++ (bench.latency_batch1), batches of 2 / 8 / 32,
# and the GPU suite (bit-identity over batch sizes) first
TAG=${1:-r14u}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -x -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
cat > /tmp/lat.py <<PY
import sys, json, ctypes, time
sys.path[:0]=['$REPO','$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
lib=sys.argv[1]
if lib!='product':
    _hip._lib=_hip.bind(ctypes.CDLL(lib))
import bench
dev=torch.device('cuda',0)
for name in ('ecapa1024','campp'):
    r=bench.latency_batch1(name, dev)
    print(json.dumps(dict(lib=lib.split('/')[-1], model=name, eager_p50=r['eager_p50'], gpu_us=r['gpu_us_back_to_back'], graph_p50=r['hipgraph_p50'])))
featurizer, model, _ = bench.build('ecapa1024', dev)
for B in (2, 8, 32):
    g = torch.Generator().manual_seed(1)
    wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        for _ in range(5): model(featurizer(wav))
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): model(featurizer(wav))
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    print(json.dumps(dict(lib=lib.split('/')[-1], model='ecapa1024', B=B, us=round(dt*1e6,1))))
PY
for rep in 1 2; do
  for lib in product $REPO/tools/probe/libconv1d_ns2.so; do
    timeout 300 python /tmp/lat.py $lib 2>/dev/null | grep "^{" | tee -a $OUT/latency_small_batches_ring_ab.log
  done
done
