"""Race detector for the hand-synchronised kernels (counted s_waitcnt rings, barrier-free K loops, cross-wave LDS hand-overs): the same
batch through the same model N times must give bit-identical embeddings every time -- there are no atomics on the path, so any difference
between two runs is a synchronisation bug.  usage: python tools/stress_determinism.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda', 0)
for name in ('ecapa1024', 'ecapa512', 'campp', 'ecapa512_mel', 'eres2netv2'):
    featurizer, model, _ = bench.build(name, dev)
    B = 256 if not name.startswith('eres') else 64
    g = torch.Generator().manual_seed(99)
    wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
    # a second, different batch interleaved with the first: stale data of the other batch would show up as a mismatch
    wav2 = wav.flip(0).contiguous() * 0.5
    bad = 0
    with torch.no_grad():
        f0 = featurizer(wav).clone()
        e0 = model(f0).clone()
        for i in range(steps):
            model(featurizer(wav2))
            f = featurizer(wav)
            e = model(f)
            if not (torch.equal(f, f0) and torch.equal(e, e0)):
                bad += 1
    torch.cuda.synchronize()
    print(json.dumps({'model': name, 'steps': steps, 'batch': B, 'mismatching_steps': bad}), flush=True)
    del model, featurizer
    torch.cuda.empty_cache()
