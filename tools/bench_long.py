"""Throughput in audio-seconds per second as a function of the utterance length (the fused per-utterance kernels have length limits: CAM++ dense
layers T/2 <= 160 frames = 3.2 s, Res2Net chain T <= 320 frames; beyond them the models run their multi-launch forms).
usage: python tools/bench_long.py [model ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench

dev = torch.device('cuda', 0)
for name in (sys.argv[1:] or ['campp', 'ecapa1024']):
    featurizer, model, _ = bench.build(name, dev)
    for secs, B in ((3.0, 256), (3.2, 240), (3.3, 232), (6.0, 128), (10.0, 76)):
        g = torch.Generator().manual_seed(1)
        wav = (0.1 * torch.randn([B, int(secs * 16000)], generator=g)).clamp(-1, 1).to(dev)
        with torch.no_grad():
            for _ in range(3):
                model(featurizer(wav))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                model(featurizer(wav))
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'{name}: {B} x {secs:.1f} s  {dt * 1e3:.2f} ms per batch  {B / dt:.0f} utt/s  {B * secs / dt / 1e3:.1f} k audio-seconds/s', flush=True)
