"""Latency of one small-batch forward, eager launches vs hipGraph replay (torch.cuda.CUDAGraph around the native forward).
usage: python tools/bench_graph.py [model] [batch]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'campp'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device('cuda', 0)
featurizer, model, _ = bench.build(name, dev)
g0 = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g0)).clamp(-1, 1).to(dev)
with torch.no_grad():
    def fwd(w):
        return model(featurizer(w))
    for _ in range(5):
        ref = fwd(wav)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        fwd(wav)
    torch.cuda.synchronize()
    eager_us = (time.perf_counter() - t0) / n * 1e6
    static_in = wav.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = fwd(static_in)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph_us = (time.perf_counter() - t0) / n * 1e6
    print(json.dumps({'model': name, 'batch': B, 'eager_us_per_forward': round(eager_us, 1), 'graph_us_per_forward': round(graph_us, 1),
                      'identical': bool(torch.equal(out, ref))}))
