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
# MV_STRESS_MODELS=a,b,... selects the models; "campp_f32" = CAM++ pinned onto its exact (split-operand) head
names = os.environ.get('MV_STRESS_MODELS', 'ecapa1024,ecapa512,campp,ecapa512_mel,eres2netv2').split(',')
for name in names:
    featurizer, model, _ = bench.build('campp' if name == 'campp_f32' else name, dev)
    if name == 'campp_f32':
        model.head_precision = 'f32'
    B = 256 if not name.startswith('eres') else (16 if 'w96' in name else 64)
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
    # two-stream arm: ONE featurizer handle driven from two streams at once (the handles own no mutable state, include/mvector_hip.h), each
    # stream with its own model handle + workspace (a Python Model keeps one workspace); sub-chip batches, i.e. the several-workgroups form of
    # the front-end with its per-call scratch.  Every result must carry the bits of the serial run above.
    import copy
    n1 = min(96, B // 2)
    halves = [wav[:n1].contiguous(), wav[n1:min(224, B)].contiguous()]
    models = [model, copy.deepcopy(model)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad2 = 0
    with torch.no_grad():
        refs = [(featurizer(h).clone(), models[k](featurizer(h)).clone()) for k, h in enumerate(halves)]
        torch.cuda.synchronize()
        for i in range(steps):
            outs = []
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    f = featurizer(halves[k])
                    outs.append((f, models[k](f)))
            torch.cuda.synchronize()
            bad2 += sum(int(not (torch.equal(f, rf) and torch.equal(e, re))) for (f, e), (rf, re) in zip(outs, refs))
        # ... and a row's bits do not depend on the batch it sits in (featurizer.py:125-130 computes every row alone)
        same_rows = bool(torch.equal(refs[0][0], f0[:n1]) and torch.equal(refs[1][0], f0[n1:min(224, B)]))
    print(json.dumps({'model': name, 'two_streams_one_featurizer': {'steps': steps, 'mismatching_results': bad2, 'feature_rows_equal_full_batch': same_rows}}), flush=True)
    del model, featurizer, models
    torch.cuda.empty_cache()
