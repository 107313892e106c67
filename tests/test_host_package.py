"""CPU tests of the host-side mirror of the reference API: state_dict contract, CPU front-end, predictor plumbing
(BASELINE config 0: TDNN + Fbank-80 on the reference's sample audio), and the world_size-2 sharded path on gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import GOLDEN, cos_dist, load_case
from oracle import frontend, models as omodels, scoring

FB = dict(sample_frequency=16000, num_mel_bins=80)


@pytest.mark.parametrize('case', ['ecapa_tiny', 'ecapa_c512', 'ecapa_c1024', 'ecapa_mel128', 'campp', 'campp_c64', 'tdnn', 'eres2net_tiny',
                                  'eres2netv2_tiny', 'eres2net_m32', 'eres2netv2_m32', 'eres2netv2_w96s4'])
def test_state_dict_layout_equals_reference_manifest(case):
    import mvector.models as M
    with open(os.path.join(GOLDEN, f'manifest_{case}.json')) as f:
        man = json.load(f)
    m = getattr(M, man['model'])(**man['kwargs'])
    sd = m.state_dict()
    assert list(sd.keys()) == list(man['shapes'].keys())
    assert {k: list(v.shape) for k, v in sd.items()} == man['shapes']
    assert m.embd_dim == man['kwargs'].get('embd_dim', 192)


@pytest.mark.parametrize('case', ['ecapa_tiny', 'campp_short', 'tdnn', 'eres2net_tiny', 'eres2netv2_tiny'])
def test_cpu_module_forward_matches_reference_golden(case):
    import mvector.models as M
    man, sd, x, emb_ref, _ = load_case(case)
    m = getattr(M, man['model'])(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        assert torch.allclose(m(x), emb_ref, atol=1e-4, rtol=1e-4)


def test_build_model_and_config_objects():
    from mvector.models import build_model
    from mvector.utils.utils import dict_to_object
    cfg = dict_to_object(dict(model_conf=dict(model='EcapaTdnn', model_args=dict(embd_dim=192, pooling_type='ASP',
                                                                                  channels=[64, 64, 64, 64, 192]))))
    m = build_model(80, cfg)
    assert type(m).__name__ == 'EcapaTdnn' and m.embd_dim == 192
    assert cfg.model_conf.model_args.channels[-1] == 192
    with pytest.raises(AttributeError):
        build_model(80, dict_to_object(dict(model_conf=dict(model='NoSuchModel'))))
    import mvector.models as M
    for name in ('Res2Net', 'ResNetSE', 'SpeakerIdentification'):  # exported like the reference, outside the accelerated path
        assert hasattr(M, name)
        with pytest.raises(NotImplementedError, match='outside the MI355X embedding path'):
            build_model(80, dict_to_object(dict(model_conf=dict(model=name))))


def test_build_model_eres2net_family_and_constructor_contract():
    """configs/eres2net.yml style construction (model name + model_args), the reference's attribute surface, CPU forward."""
    from mvector.models import build_model
    from mvector.utils.utils import dict_to_object
    for name, extra in (('ERes2Net', {}), ('ERes2NetV2', dict(two_emb_layer=True))):
        cfg = dict_to_object(dict(model_conf=dict(model=name, model_args=dict(embd_dim=48, m_channels=16, num_blocks=[1, 1, 1, 1], **extra))))
        m = build_model(16, cfg).eval()
        assert type(m).__name__ == name and m.embd_dim == 48 and m.stats_dim == 2 * 16 * 8
        with torch.no_grad():
            e = m(torch.randn(2, 40, 16))
        assert e.shape == (2, 48) and torch.isfinite(e).all()
        ok, why = m._native_supported()
        assert ok, why
    import mvector.models as M
    assert M.ERes2Net(input_size=16, m_channels=24, num_blocks=[1, 1, 1, 1])._native_supported()[0] is False  # torch path only


def test_accuracy_helpers_match_the_reference_scan():
    """cal_accuracy_threshold: the reference scans thresholds 0.00 .. 0.99 and keeps the first best (mvector/utils/utils.py:59-71)."""
    from mvector.utils.utils import cal_accuracy, cal_accuracy_threshold, cosin_metric
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 2, 500)
    scores = (rng.normal(0.2, 0.2, 500) + 0.35 * labels).astype(np.float32)
    best_acc, best_thr = 0, 0
    for i in range(100):
        acc = np.mean(((scores >= i * 0.01) == labels).astype(int))
        if acc > best_acc:
            best_acc, best_thr = acc, i * 0.01
    acc, thr = cal_accuracy_threshold(scores, labels)
    assert abs(acc - best_acc) < 1e-12 and abs(thr - best_thr) < 1e-12
    assert abs(cal_accuracy(scores, labels, thr) - best_acc) < 1e-12
    a, b = rng.normal(size=8), rng.normal(size=8)
    assert abs(cosin_metric(a, b) - np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))) < 1e-12


def test_cpu_featurizer_matches_oracle_and_contract():
    from mvector.data_utils.featurizer import AudioFeaturizer
    fz = AudioFeaturizer('Fbank', method_args=FB)
    assert fz.feature_dim == 80 and AudioFeaturizer('MelSpectrogram').feature_dim == 128
    wav = frontend.synth_waveforms(3, 16000, seed=11)
    ratio = torch.tensor([1.0, 0.4, 0.77])
    out = fz(wav, ratio)
    ref = frontend.audio_featurizer(wav, ratio, 'Fbank', FB)
    assert out.shape == (3, 98, 80) and (out - ref).abs().max() < 1e-4
    assert fz(wav[0]).shape == (1, 98, 80)  # 1-D input is unsqueezed (featurizer.py:63-64)
    mel = AudioFeaturizer('MelSpectrogram', method_args={})
    refm = frontend.audio_featurizer(wav, ratio, 'MelSpectrogram', {})
    assert torch.allclose(mel(wav, ratio), refm, rtol=1e-4, atol=1e-5)
    with pytest.raises(Exception):
        AudioFeaturizer('NoSuchFeature')
    with pytest.raises(TypeError):
        AudioFeaturizer('Fbank', method_args=dict(not_an_argument=1))


def test_cpu_kaldi_fbank_module_matches_oracle():
    """the reference's KaldiFbank (featurizer.py:114-132): [Batch, Length] -> [Batch, Feature, Length], kaldi.fbank per row, no CMN"""
    from mvector.data_utils.featurizer import KaldiFbank
    m = KaldiFbank(**FB)
    wav = frontend.synth_waveforms(3, 16000, seed=12)
    out = m(wav)
    ref = torch.stack([frontend.kaldi_fbank(row.unsqueeze(0), **FB).transpose(0, 1) for row in wav])
    assert out.shape == (3, 80, 98) and (out - ref).abs().max() < 1e-4
    assert torch.equal(m(wav.unsqueeze(1)), out)      # rows as [1, Length]
    assert KaldiFbank(sample_frequency=16000)(wav[:1]).shape == (1, 23, 98)   # torchaudio's default 23 bins
    with pytest.raises(TypeError):
        KaldiFbank(not_an_argument=1)
    with pytest.raises(ValueError):
        m(wav[0])


def _write_model_and_audio(tmp_path):
    import scipy.io.wavfile as wavfile
    man, sd, _, _, _ = load_case('tdnn')
    model_dir = tmp_path / 'model'
    model_dir.mkdir()
    torch.save({'0.' + k: v for k, v in sd.items()}, str(model_dir / 'model.pth'))
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    paths = []
    for i, pcm in enumerate(z['pcm16']):
        p = str(tmp_path / f'u{i}.wav')
        wavfile.write(p, 16000, pcm)
        paths.append(p)
    cfg = dict(dataset_conf=dict(dataset=dict(min_duration=0.3, sample_rate=16000, use_dB_normalization=False, target_dB=-20),
                                 eval_conf=dict(batch_size=2)),
               preprocess_conf=dict(feature_method='Fbank', method_args=dict(sample_frequency=16000, num_mel_bins=80)),
               model_conf=dict(model='TDNN', model_args=dict(embd_dim=192)))
    return cfg, str(model_dir), paths, sd, z


def test_cpu_predictor_plumbing_config0(tmp_path):
    """BASELINE configs[0]: TDNN + Fbank-80, 4 x 1 s utterances from dataset/*.wav, CPU predict path."""
    from mvector.predict import MVectorPredictor
    cfg, model_dir, paths, sd, z = _write_model_and_audio(tmp_path)
    p = MVectorPredictor(cfg, model_path=model_dir, use_gpu=False)
    emb = p.predict_batch(paths)
    feats = torch.from_numpy(z['fbank'])
    ref = omodels.tdnn(sd, feats).numpy()
    assert emb.shape == (4, 192) and np.abs(emb - ref).max() < 1e-3
    golden = np.load(os.path.join(GOLDEN, 'tdnn.npz'))['emb']  # produced by the reference's own TDNN module
    assert np.abs(emb - golden).max() < 1e-3
    single = p.predict(paths[2])
    assert np.abs(single - ref[2]).max() < 1e-3
    c = p.contrast(paths[0], paths[3])
    assert abs(c - scoring.contrast(ref[0], ref[3])) < 1e-5
    pcm = z['pcm16'][1].astype(np.float32) / 32768.0
    assert np.abs(p.predict(pcm, sample_rate=16000) - ref[1]).max() < 1e-3  # ndarray input
    with pytest.raises(AssertionError):
        p.predict(np.zeros(100, dtype=np.float32))  # shorter than min_duration


def test_cpu_predictor_audio_db_register_recognise(tmp_path):
    from mvector.predict import MVectorPredictor
    cfg, model_dir, paths, sd, z = _write_model_and_audio(tmp_path)
    db = tmp_path / 'audio_db'
    p = MVectorPredictor(cfg, threshold=0.5, audio_db_path=str(db), model_path=model_dir, use_gpu=False)
    assert p.register(paths[0], 'alice') == (True, '注册成功')
    assert p.register(paths[2], 'bob')[0]
    name, score = p.recognition(paths[0])
    assert name == 'alice' and score > 0.99
    assert set(p.get_users()) == {'alice', 'bob'}
    assert os.path.exists(os.path.join(str(db), 'audio_indexes.bin'))
    p2 = MVectorPredictor(cfg, threshold=0.5, audio_db_path=str(db), model_path=model_dir, use_gpu=False)  # reload index
    assert p2.recognition(paths[2])[0] == 'bob'
    assert p2.remove_user('bob') and p2.get_users() == ['alice']
    assert p2.recognition(paths[2], threshold=0.999)[0] is None


def test_load_pretrained_drops_mismatched_shapes(tmp_path):
    from mvector.utils.checkpoint import load_pretrained
    from mvector.models import TDNN
    man, sd, _, _, _ = load_case('tdnn')
    state = {'0.' + k: v for k, v in sd.items()}
    state['0.linear.weight'] = torch.zeros(7, 3)  # wrong shape -> skipped, not fatal (checkpoint.py:33-38 intent)
    state['1.weight'] = torch.zeros(5, 192)       # classifier head of the training checkpoint -> ignored
    path = str(tmp_path / 'model.pth')
    torch.save(state, path)
    model = torch.nn.Sequential(TDNN(80))
    before = model[0].linear.weight.clone()
    load_pretrained(model, path, use_gpu=False)
    assert torch.equal(model[0].linear.weight, before)
    assert torch.equal(model[0].bn1.running_mean, sd['bn1.running_mean'])


GLOO_WORKER = r'''
import os, sys, json, numpy as np, torch, torch.distributed as dist
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "tests"), os.path.join(sys.argv[1], "voiceprintrecognition-pytorch_amd")]
from helpers import load_case
from mvector.models import EcapaTdnn
from mvector.data_utils.featurizer import AudioFeaturizer
from mvector import parallel
from oracle import frontend
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
man, sd, _, _, _ = load_case("ecapa_tiny")
model = EcapaTdnn(**man["kwargs"]); model.load_state_dict(sd); model.eval()
fz = AudioFeaturizer("Fbank", method_args=dict(sample_frequency=16000, num_mel_bins=80))
wav = frontend.synth_waveforms(6, 8000, seed=21)
lo, hi = parallel.shard_rows(6)
emb, emb_all, scores = parallel.embed_and_score(fz, model, wav[lo:hi])
np.savez(sys.argv[2] + f"/rank{dist.get_rank()}.npz", emb=emb.numpy(), emb_all=emb_all.numpy(), scores=scores.numpy(), lo=lo, hi=hi)
dist.destroy_process_group()
'''


BUCKET_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "tests"), os.path.join(sys.argv[1], "voiceprintrecognition-pytorch_amd")]
from helpers import load_case
from mvector.models import EcapaTdnn
from mvector.data_utils.featurizer import AudioFeaturizer
from mvector import parallel
from oracle import frontend
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
man, sd, _, _, _ = load_case("ecapa_tiny")
model = EcapaTdnn(**man["kwargs"]); model.load_state_dict(sd); model.eval()
fz = AudioFeaturizer("Fbank", method_args=dict(sample_frequency=16000, num_mel_bins=80))
lens = [int(v) for v in sys.argv[3].split(",")]
wav = frontend.synth_waveforms(len(lens), max(lens), seed=33)
emb = parallel.embed_bucketed(fz, model, [wav[i, :n] for i, n in enumerate(lens)], max_buckets=3, device=torch.device("cpu"))
np.save(sys.argv[2] + f"/bucketed_rank{dist.get_rank()}.npy", emb.numpy())
dist.destroy_process_group()
'''


def test_length_buckets_partition():
    from mvector.parallel import length_buckets
    lens = [16000, 160000, 16001, 90000, 52000, 51999, 160000, 30000]
    b = length_buckets(lens, 8)
    assert sorted(i for g in b for i in g) == list(range(len(lens))) and len(b) <= 8
    assert all(max(lens[i] for i in g) - min(lens[i] for i in g) <= (160000 - 16000) // 8 + 1 for g in b)
    assert [max(lens[i] for i in g) for g in b] == sorted(max(lens[i] for i in g) for g in b)  # shortest bucket first
    assert length_buckets([], 8) == [] and length_buckets([5, 5, 5], 4) == [[0, 1, 2]]
    assert len(length_buckets(lens, 1)) == 1


def test_bucketed_variable_length_world_size_2_gloo(tmp_path):
    """BASELINE config 4 layout on CPU: variable-length utterances, length buckets, every bucket sharded over 2 ranks (uneven
    shards), padded all-gather; result = per-bucket predict_batch semantics (padding to the bucket maximum, Q2)."""
    from mvector.parallel import length_buckets
    lens = [8000, 11000, 8400, 16000, 15000, 12000, 9000]
    script = tmp_path / 'bucket_worker.py'
    script.write_text(BUCKET_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), ','.join(map(str, lens))],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    man, sd, _, _, _ = load_case('ecapa_tiny')
    wav = frontend.synth_waveforms(len(lens), max(lens), seed=33)
    want = np.zeros((len(lens), 192), dtype=np.float32)
    for idx in length_buckets(lens, 3):
        longest = max(lens[i] for i in idx)
        padded = torch.zeros(len(idx), longest)
        for r, i in enumerate(idx):
            padded[r, :lens[i]] = wav[i, :lens[i]]
        ratio = torch.tensor([lens[i] / longest for i in idx])
        want[idx] = omodels.ecapa_tdnn(sd, frontend.audio_featurizer(padded, ratio, 'Fbank', FB)).numpy()
    for r in range(2):
        got = np.load(str(tmp_path / f'bucketed_rank{r}.npy'))
        assert cos_dist(got, want).max() < 1e-6 and np.abs(got - want).max() < 1e-3


def test_sharded_path_world_size_2_gloo(tmp_path):
    """N>1 layout on CPU: rows sharded over 2 ranks, all-gather of embeddings, each rank scores its rows vs all."""
    script = tmp_path / 'worker.py'
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    from mvector.models import EcapaTdnn
    man, sd, _, _, _ = load_case('ecapa_tiny')
    wav = frontend.synth_waveforms(6, 8000, seed=21)
    full = omodels.ecapa_tdnn(sd, frontend.audio_featurizer(wav, None, 'Fbank', FB)).numpy()
    sim = scoring.cosine_similarity(full, full)
    for r in range(2):
        z = np.load(str(tmp_path / f'rank{r}.npz'))
        lo, hi = int(z['lo']), int(z['hi'])
        assert (lo, hi) == ((0, 3), (3, 6))[r]
        assert np.abs(z['emb_all'] - full).max() < 1e-3          # every rank holds the whole embedding matrix
        assert np.abs(z['scores'] - sim[lo:hi]).max() < 1e-5     # and the score block of its own rows


def test_native_handle_cache_is_not_part_of_module_state(tmp_path):
    """A backbone that has built its native handle must still deep-copy, pickle and torch.save: the handle cache
    holds ctypes objects; copies build their own handle on first use.  In-place parameter edits invalidate the cache key."""
    import copy
    import ctypes
    import pickle
    from mvector.models import EcapaTdnn
    man, sd, _, _, _ = load_case('ecapa_tiny')
    m = EcapaTdnn(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval()

    class Handle:  # what _hip.Model holds: raw pointers and a CDLL
        def __init__(self):
            self.h, self.lib = ctypes.c_void_p(1234), ctypes.CDLL(None)
    tensors = m._forward_tensors(m.state_dict(keep_vars=True))   # the list the handle caches: the per-forward check walks it, not state_dict()
    v0 = m._params_version(tensors)
    assert v0 == m._params_version()
    m.__dict__['_native_handles'] = {0: (Handle(), v0, tensors)}
    c = copy.deepcopy(m)
    assert '_native_handles' not in c.__dict__ and '_native_handles' in m.__dict__
    assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), m.state_dict().values()))
    assert c.blocks[0].conv.conv.weight.data_ptr() != m.blocks[0].conv.conv.weight.data_ptr()
    r = pickle.loads(pickle.dumps(m))
    assert '_native_handles' not in r.__dict__ and torch.equal(r.fc.conv.weight, m.fc.conv.weight)
    torch.save(torch.nn.Sequential(m), str(tmp_path / 'whole_model.pt'))
    # an in-place edit under no_grad (EMA swap, weight surgery) changes the version key the handle was built under
    with torch.no_grad():
        m.fc.conv.weight.mul_(1.0)
    assert m._params_version() != v0 and m._params_version(tensors) != v0
    # gradient analysis with respect to the input needs the torch graph; plain inference (grad mode on, default
    # requires_grad parameters -- what the reference's predictor does) must stay on the native path
    x = torch.zeros(1, 20, man['kwargs']['input_size'])

    class FakeCuda(torch.Tensor):
        is_cuda = True
    assert m._use_native(x) is False                       # CPU tensor
    fx = x.as_subclass(FakeCuda)
    assert m._use_native(fx) is True
    fxg = x.clone().requires_grad_(True).as_subclass(FakeCuda)
    assert m._use_native(fxg) is False
    with torch.no_grad():
        assert m._use_native(fxg) is True
    m.train()
    assert m._use_native(fx) is False and '_native_handles' in m.__dict__ and not m.__dict__['_native_handles']


def test_device_gallery_sums_grow_and_remove_on_cpu_tensors():
    """DeviceGallery bookkeeping (the GPU-resident enrolment matrix of the HIP predictor), exercised on CPU tensors: per-user
    sums point the same way as the reference's per-user means, growth keeps rows, removal keeps the order of the others."""
    from mvector.infer_utils.gallery import DeviceGallery
    rng = np.random.default_rng(0)
    dg = DeviceGallery(torch.device('cpu'), 6, capacity=2)
    rows = {f'u{i}': [rng.normal(size=6).astype(np.float32) for _ in range(1 + i % 3)] for i in range(7)}
    for name, rs in rows.items():
        for r in rs:
            dg.add(name, torch.from_numpy(r))
    assert dg.users == list(rows) and dg.matrix().shape == (7, 6) and dg.sums.shape[0] >= 7
    for i, name in enumerate(rows):
        mean = np.mean(rows[name], axis=0)
        assert abs(scoring.contrast(dg.matrix()[i].numpy(), mean) - 1.0) < 1e-6
    assert dg.remove('u2') and not dg.remove('u2')
    assert dg.users == ['u0', 'u1', 'u3', 'u4', 'u5', 'u6']
    assert np.allclose(dg.matrix()[2].numpy(), np.sum(rows['u3'], axis=0), atol=1e-6)
    dg.add('u2', rows['u2'][0])  # numpy row: one upload, new last row
    assert dg.users[-1] == 'u2' and dg.uploads == 0  # same device: nothing to move


def test_embed_stream_cpu_is_the_plain_loop():
    """parallel.embed_stream on CPU tensors: batch by batch, int16 PCM scaled like the device path, ragged last batch."""
    from mvector import parallel
    from mvector.models import EcapaTdnn
    from mvector.data_utils.featurizer import AudioFeaturizer
    man, sd, _, _, _ = load_case('ecapa_tiny')
    m = EcapaTdnn(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval()
    fz = AudioFeaturizer('Fbank', method_args=FB)
    wav = frontend.synth_waveforms(7, 8000, seed=3)
    pcm = (wav * 32768).round().clamp(-32768, 32767).to(torch.int16)
    batches = [pcm[:3], pcm[3:6], pcm[6:]]
    outs = list(parallel.embed_stream(fz, m, batches, device='cpu'))
    assert [o.shape[0] for o in outs] == [3, 3, 1]
    with torch.no_grad():
        for o, b in zip(outs, batches):
            assert torch.equal(o, m(fz(b.float() / 32768.0)))


def test_embed_stream_cpu_applies_the_reference_db_normalisation():
    """the CPU branch of embed_stream ignored target_db.  int16 rows of different loudness must come out as the
    embeddings of the normalised waveforms (AudioSegment.normalize as called at predict.py:210-211: gain = target - 10 log10(mean x^2)),
    a silent row stays unscaled."""
    from mvector import parallel
    from mvector.models import EcapaTdnn
    from mvector.data_utils.featurizer import AudioFeaturizer
    man, sd, _, _, _ = load_case('ecapa_tiny')
    m = EcapaTdnn(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval()
    fz = AudioFeaturizer('Fbank', method_args=FB)
    wav = frontend.synth_waveforms(3, 8000, seed=5) * torch.tensor([0.05, 0.5, 0.0])[:, None]
    pcm = (wav * 32768).round().clamp(-32768, 32767).to(torch.int16)
    (out,) = list(parallel.embed_stream(fz, m, [pcm], device='cpu', target_db=-20.0))
    w = pcm.float() / 32768.0
    ref = w.clone()
    for i in range(2):
        rms_db = 10.0 * torch.log10((w[i] ** 2).mean())
        ref[i] = w[i] * 10.0 ** ((-20.0 - rms_db) / 20.0)
    with torch.no_grad():
        expect = m(fz(ref))
    assert torch.allclose(out, expect, atol=1e-5) and torch.isfinite(out).all()
    (plain,) = list(parallel.embed_stream(fz, m, [pcm], device='cpu'))
    assert not torch.allclose(plain[:2], expect[:2], atol=1e-3)   # the gain matters for these rows


def _bench(*argv, launcher=False):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    cmd = [sys.executable]
    if launcher:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port)]
    return subprocess.run(cmd + [os.path.join(ROOT, 'bench.py')] + list(argv), env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_bench_launches_its_own_ranks_and_fails_clearly_without_devices():
    """`python bench.py --gpus 2` with no launcher (the form of the driver's N = 1 command): the script starts 2 ranks itself; both rendezvous (gloo),
    meet again on a second group, and rank 0 prints ONE JSON line (--dry-launch: no device needed).  Without --dry-launch on a node with fewer than 2
    devices the run ends AFTER the rendezvous with a message that says so -- not with an assert.  Under torch.distributed.run
    (the driver's N > 1 command) the same ranks run main() directly."""
    r = _bench('--gpus', '2', '--dry-launch')
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {'dry_launch': True, 'n_gpus': 2, 'ranks_seen': [0, 1], 'ranks_seen_second_group': [0, 1],
                 'devices_per_rank': d['devices_per_rank'], 'launcher': 'self'}
    r = _bench('--gpus', '2', '--dry-launch', launcher=True)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['launcher'] == 'torch.distributed.run' and d['ranks_seen'] == [0, 1]
    if torch.cuda.device_count() < 2:
        r = _bench('--gpus', '2', '--steps', '1', '--warmup', '0')
        assert r.returncode != 0
        assert 'all 2 ranks met' in r.stderr and 'visible device(s)' in r.stderr, r.stderr[-2000:]
        assert 'AssertionError' not in r.stderr
    r = _bench('--dry-launch')   # default --gpus 1 without a launcher (this raised ValueError from env:// without MASTER_PORT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == [0] and d['launcher'] == 'self'
    r = _bench('--gpus', '3', '--dry-launch', launcher=True)   # launcher and flag disagree: said in words
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr


SYNC_WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "tests"), os.path.join(sys.argv[1], "voiceprintrecognition-pytorch_amd")]
from mvector.models import CAMPPlus, EcapaTdnn
from mvector import parallel
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
torch.manual_seed(0)
campp = torch.nn.Sequential(CAMPPlus(input_size=80, embd_dim=32)).eval()
if rank == 0:
    campp[0].head_precision = "f32"          # what rank 0's handle decided (here: pinned; no device in this container)
x = torch.randn(2, 60, 80)
n_fwd = 3 if rank == 0 else 0               # rank 1 has an empty shard / does not evaluate: forwards must not be collective
for _ in range(n_fwd):
    campp(x)
pinned = parallel.sync_native_choices(campp)  # the explicit, once-per-job collective
ecapa = torch.nn.Sequential(EcapaTdnn(input_size=80, channels=[64, 64, 64, 64, 192])).eval()
none = parallel.sync_native_choices(ecapa)    # nothing to agree on: no collective issued, returns {}
dist.barrier()
json.dump({"pinned": pinned, "head": campp[0].head_precision, "none": none}, open(sys.argv[2] + f"/sync_rank{rank}.json", "w"))
dist.destroy_process_group()
'''


def test_campp_head_choice_is_synchronised_explicitly_not_from_forward(tmp_path):
    """CAMPPlus.forward issued a dist.broadcast whenever a native handle was built -- ranks running different numbers of
    forwards (an empty shard, rank-0-only evaluation as in the reference's trainer.py:376) blocked each other.  Now: forward never touches the
    process group (rank 0 runs three forwards here, rank 1 none, nobody hangs), and parallel.sync_native_choices -- one explicit collective
    every rank calls -- pins rank 0's choice on all ranks."""
    import inspect
    from mvector.models import CAMPPlus
    from mvector.models._native import NativeBackbone
    # the build hook issues no collective (round 6: it only reads the handle's x-vector sensitivity report and warns)
    assert 'dist' not in inspect.getsource(CAMPPlus._native_created) and 'dist' not in inspect.getsource(NativeBackbone._native_created)
    assert 'dist.' not in inspect.getsource(CAMPPlus.forward) and 'dist.' not in inspect.getsource(NativeBackbone._native_handle_on)
    script = tmp_path / 'sync_worker.py'
    script.write_text(SYNC_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29633', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    for r in range(2):
        d = json.load(open(tmp_path / f'sync_rank{r}.json'))
        assert d == {'pinned': {'0': 'f32'}, 'head': 'f32', 'none': {}}, (r, d)


def test_cpu_featurizer_takes_the_wider_method_args():
    """AudioFeaturizer on CPU tensors (the DataLoader workers' path, reader.py:103,120) over the argument lists the device sweeps run
    (layer_checks.FBANK_ARG_CASES / MELSPEC_ARG_CASES): the batched torch front-end equals the oracle; what is not implemented says so"""
    import layer_checks as lc
    from mvector.data_utils.featurizer import AudioFeaturizer
    wav = frontend.synth_waveforms(3, 9000, seed=5)
    ratio = torch.tensor([1.0, 0.5, 0.8])
    for args, opt in lc.FBANK_ARG_CASES:
        if opt.get('cmn', True) and not opt.get('varlen'):
            out = AudioFeaturizer('Fbank', method_args=args)(wav, ratio)
            ref = frontend.audio_featurizer(wav, ratio, 'Fbank', args)
            scale = max(1.0, ref.abs().max().item()) if not args.get('use_log_fbank', True) else 1.0
            assert out.shape == ref.shape and (out - ref).abs().max().item() <= 1e-4 * scale, args
    for args in lc.MELSPEC_ARG_CASES:
        out = AudioFeaturizer('MelSpectrogram', method_args=args)(wav, ratio)
        ref = frontend.audio_featurizer(wav, ratio, 'MelSpectrogram', args)
        assert out.shape == ref.shape and (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), args
    for bad in (dict(dither=0.1), dict(round_to_power_of_two=False)):
        with pytest.raises(NotImplementedError):
            AudioFeaturizer('Fbank', method_args=dict(FB, **bad))
    for bad in (dict(high_freq=8100.0), dict(low_freq=7700.0, high_freq=-400.0), dict(num_mel_bins=3), dict(preemphasis_coefficient=-0.1)):   # torchaudio asserts on these
        with pytest.raises(AssertionError):
            AudioFeaturizer('Fbank', method_args=dict(FB, **bad))(wav, ratio)
    for bad in (dict(pad_mode='symmetric'), dict(onesided=False), dict(power=None)):
        with pytest.raises(NotImplementedError):
            AudioFeaturizer('MelSpectrogram', method_args=bad)
    for bad in (dict(norm='area'), dict(mel_scale='bark'), dict(normalized='sqrt')):
        with pytest.raises(ValueError):
            AudioFeaturizer('MelSpectrogram', method_args=bad)
    with pytest.raises(ValueError, match='Require f_min'):   # (torchaudio.transforms.MelScale.__init__)
        AudioFeaturizer('MelSpectrogram', method_args=dict(f_min=4000.0, f_max=3000.0))(wav, ratio)
    with pytest.raises(ValueError, match='Require f_min'):
        frontend.mel_spectrogram(wav, f_min=4000.0, f_max=3000.0)
