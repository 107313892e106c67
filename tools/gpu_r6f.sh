#!/bin/bash
# Round-6 session F: length buckets on two HIP streams (one native handle, a workspace per stream): GPU tests of the bucketed / ERes2Net paths, then
# BASELINE config 5's leg (64 utterances of 1-10 s, 8 buckets, ERes2NetV2 54.9 M) with 1 / 2 / 3 / 4 streams
TAG=${1:-r15f}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "bucket or eres2 or bits or predictor or embed_stream" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log | cut -c1-300
timeout 900 python - <<'PY' 2>&1 | grep -v Warning | tee $OUT/config5_streams.log
import sys, time, json, torch
sys.path[:0] = ['.', 'voiceprintrecognition-pytorch_amd']
import bench
from mvector import parallel
dev = torch.device('cuda:0')
r = bench.bucketed_run('eres2netv2_w96s4', dev, 64, 2)
print(json.dumps({k: r[k] for k in ('value', 'ms_per_pass', 'value_one_stream', 'ms_per_pass_one_stream', 'identical_to_one_stream', 'parity')}))
featurizer, model, _ = bench.build('eres2netv2_w96s4', dev)
g = torch.Generator().manual_seed(4321)
lens = torch.randint(16000, 160001, (64,), generator=g).tolist()
waves = [(0.1 * torch.randn(n, generator=g)).clamp(-1, 1).to(dev) for n in lens]
for rep in range(2):
    for ns in (1, 2, 3, 4):
        parallel.embed_bucketed(featurizer, model, waves, max_buckets=8, device=dev, streams=ns)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            parallel.embed_bucketed(featurizer, model, waves, max_buckets=8, device=dev, streams=ns)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(json.dumps(dict(streams=ns, rep=rep, ms_per_pass=round(dt * 1e3, 1), utt_per_s=round(64 / dt, 1))), flush=True)
PY
