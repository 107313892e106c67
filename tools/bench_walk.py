"""A/B of the walk-direction knobs of the EcapaTdnn forward (MV_WALK bit mask, MV_ASP_CHUNKS) in ONE process: the knobs are read per
forward, so every variant runs on the same box, the same model handle and the same waveforms; variants alternate over `rounds` passes.
usage: python tools/bench_walk.py [steps] [rounds]      prints ms per step (front-end + backbone + cosine as in bench.py's step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda', 0)
featurizer, model, _ = bench.build('ecapa1024', dev)
g0 = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([256, bench.SAMPLES], generator=g0)).clamp(-1, 1).to(dev)

# (MV_WALK, MV_ASP_CHUNKS)
VARIANTS = [(0, 1), (1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1), (64, 1), (128, 1), (5, 1), (13, 1), (1 | 4 | 64, 1), (1 | 4 | 32, 1),
            (64 | 128, 1), (0, 2), (64, 2), (128, 2), (1 | 4 | 64, 2), (0, 4)]


def run(walk, chunks, n):
    os.environ['MV_WALK'] = str(walk)
    os.environ['MV_ASP_CHUNKS'] = str(chunks)
    with torch.no_grad():
        for _ in range(3):
            emb = model(featurizer(wav))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            emb = model(featurizer(wav))
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, emb


ref = None
res = {v: [] for v in VARIANTS}
for r in range(rounds):
    for v in VARIANTS:
        ms, emb = run(v[0], v[1], steps)
        if ref is None:
            ref = emb.clone()
        assert torch.equal(emb, ref), f'variant {v} changes the embeddings'
        res[v].append(ms)
base = sorted(res[(0, 1)])[len(res[(0, 1)]) // 2]
for v in VARIANTS:
    xs = sorted(res[v])
    med = xs[len(xs) // 2]
    print(f'MV_WALK={v[0]:3d} MV_ASP_CHUNKS={v[1]}  median {med:.4f} ms  ({(med / base - 1) * 100:+.2f} %)  all {" ".join(f"{x:.4f}" for x in res[v])}')
