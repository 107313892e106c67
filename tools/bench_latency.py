"""One 3 s utterance at a time (the reference's predict() shape, predict.py:214-229): N back-to-back forwards of front-end + backbone at B = 1
(or a small batch), for `rocprofv3 --kernel-trace --stats` to say which kernels the batch-1 latency is made of.
usage: python tools/bench_latency.py [model=ecapa1024] [B=1] [n=50]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'ecapa1024'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device('cuda', 0)
featurizer, model, _ = bench.build(name, dev)
g = torch.Generator().manual_seed(99)
wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
with torch.no_grad():
    for _ in range(5):
        model(featurizer(wav))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model(featurizer(wav))
    e1.record()
    torch.cuda.synchronize()
print(f'{name} B={B}: {e0.elapsed_time(e1) / n * 1e3:.1f} us of GPU time per forward (back to back, {n} forwards)')
