"""Fbank kernel duration with a cold memory hierarchy: every launch is preceded by a 1 GiB fill that evicts L2 and the
Infinity Cache (what the kernel sees inside the real step, after the backbone has streamed gigabytes); HIP events bracket
the launch only.  python tools/bench_fbank_cold.py [flush=1]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
flush = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fb = _hip.Fbank(dict(sample_frequency=16000, num_mel_bins=80))
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([256, 48000], generator=g)).clamp(-1, 1).cuda()
junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for _ in range(3):
    fb(wav)
ts = []
for _ in range(20):
    if flush:
        junk.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fb(wav)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(json.dumps(dict(info=fb.info(), impl=os.environ.get('MV_FBANK_IMPL', 'tile'), flush=flush, median_us=round(ts[len(ts) // 2], 1), min_us=round(ts[0], 1), max_us=round(ts[-1], 1))))
