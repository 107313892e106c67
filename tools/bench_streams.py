"""Experiment: the 256-utterance batch as N micro-batches on N HIP streams (one native model handle + workspace per stream), so that
the tail of one stream's launches (partial last round of GEMM tiles, one-workgroup-per-utterance kernels, small FCs) is filled by the
other stream's work.  usage: python tools/bench_streams.py [model] [steps]"""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'ecapa1024'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
featurizer, model, _ = bench.build(name, dev)
B = 256
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
with torch.no_grad():
    ref = model(featurizer(wav))
    torch.cuda.synchronize()
    for n in (1, 2, 4, 2, 1):
        streams = [torch.cuda.Stream() for _ in range(n)]
        models = [copy.deepcopy(model) for _ in range(n)]
        feats = [copy.deepcopy(featurizer) for _ in range(n)]
        chunks = list(wav.chunk(n))
        outs = [None] * n

        def step():
            for k in range(n):
                with torch.cuda.stream(streams[k]):
                    outs[k] = models[k](feats[k](chunks[k]))

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        emb = torch.cat(outs)
        err = (1 - torch.nn.functional.cosine_similarity(emb.double(), ref.double(), dim=1)).max().item()
        print(json.dumps({'model': name, 'streams': n, 'ms_per_step': round(dt / steps * 1e3, 4), 'utt_per_s': round(B * steps / dt, 1),
                          'max_one_minus_cos_vs_single_stream': err}))
        del models, feats
