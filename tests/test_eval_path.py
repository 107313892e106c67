"""Evaluation path: metrics, batch assembly, MVectorTrainer.evaluate, variable-length Fbank.

CPU tests pin the oracle and the product's host logic to the reference's own metric functions (tests/golden/metrics.npz);
the GPU tests compare the HIP path with the oracle's per-utterance restatement of the reference flow."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_case
from oracle import evaluate as oeval, frontend

FB = dict(sample_frequency=16000, num_mel_bins=80)
DEV = 'cuda'


def _golden_metrics():
    return np.load(os.path.join(GOLDEN, 'metrics.npz'))


@pytest.mark.parametrize('name', ['small', 'large'])
def test_metrics_match_reference_golden(name):
    from mvector.metric import metrics
    z = _golden_metrics()
    sc, lb = z[f'{name}_scores'], z[f'{name}_labels']
    for impl in ('product', 'oracle'):
        if impl == 'product':
            fnr, fpr, thr = metrics.compute_fnr_fpr(sc, lb)
            eer, eer_thr = metrics.compute_eer(fnr, fpr, sc)
            dcf = metrics.compute_dcf(fnr, fpr)
        else:
            fnr, fpr, thr = oeval.fnr_fpr(sc, lb)
            eer, eer_thr = oeval.eer(fnr, fpr, sc)
            dcf = oeval.min_dcf(fnr, fpr)
        assert np.array_equal(fnr, z[f'{name}_fnr']) and np.array_equal(fpr, z[f'{name}_fpr']), impl  # index work: bit-exact
        assert np.array_equal(thr, z[f'{name}_thresholds']), impl
        assert eer == float(z[f'{name}_eer']) and eer_thr == float(z[f'{name}_eer_threshold']), impl
        assert dcf == float(z[f'{name}_min_dcf']), impl
    assert metrics.compute_eer(fnr, fpr) == float(z[f'{name}_eer'])  # scores are optional


def test_collate_pads_and_keeps_order():
    from mvector.data_utils.collate_fn import collate_fn, collate_waveforms
    g = torch.Generator().manual_seed(0)
    items = [(torch.randn(t, 5, generator=g), torch.tensor(lab)) for t, lab in ((4, 7), (9, 1), (2, 3))]
    feats, labels, lens = collate_fn(items)
    assert feats.shape == (3, 9, 5) and labels.tolist() == [7, 1, 3] and lens.tolist() == [4, 9, 2]
    for i, (f, _) in enumerate(items):
        assert torch.equal(feats[i, :f.size(0)], f) and feats[i, f.size(0):].abs().sum() == 0
    wav, labels, ns = collate_waveforms([(torch.ones(5), 2), (torch.ones(3) * 2, 0)])
    assert wav.tolist() == [[1, 1, 1, 1, 1], [2, 2, 2, 0, 0]] and ns.tolist() == [5, 3] and labels.tolist() == [2, 0]


def _make_eval_set(tmp_path, n_spk=3, per_spk=3, seed=5, num_workers=0):
    """wav files of different lengths + enrol / trial lists + a TDNN checkpoint; returns everything the oracle needs too."""
    import scipy.io.wavfile as wavfile
    man, sd, _, _, _ = load_case('tdnn')
    model_dir = tmp_path / 'model'
    model_dir.mkdir()
    torch.save({'0.' + k: v for k, v in sd.items()}, str(model_dir / 'model.pth'))
    rng = np.random.default_rng(seed)
    lines = {'enroll': [], 'trials': []}
    for spk in range(n_spk):
        base = frontend.synth_waveforms(1, 16000, seed=100 + spk)[0].numpy()
        for u in range(per_spk):
            n = int(rng.integers(9000, 16000))
            x = base[:n] + 0.02 * rng.standard_normal(n).astype(np.float32)
            pcm = np.clip(x * 20000, -32768, 32767).astype(np.int16)
            path = str(tmp_path / f's{spk}_u{u}.wav')
            wavfile.write(path, 16000, pcm)
            lines['enroll' if u == 0 else 'trials'].append(f'{path}\t{spk}\n')
    for k, v in lines.items():
        with open(str(tmp_path / f'{k}.txt'), 'w') as f:
            f.writelines(v)
    cfg = dict(dataset_conf=dict(dataset=dict(min_duration=0.3, sample_rate=16000, use_dB_normalization=True, target_dB=-20),
                                 eval_conf=dict(batch_size=4, max_duration=20), dataLoader=dict(num_workers=num_workers),
                                 enroll_list=str(tmp_path / 'enroll.txt'), trials_list=str(tmp_path / 'trials.txt')),
               preprocess_conf=dict(feature_method='Fbank', method_args=FB),
               model_conf=dict(model='TDNN', model_args=dict(embd_dim=192)))
    return cfg, str(model_dir), sd


def _oracle_eval(cfg, sd):
    """the reference flow on the same files: sorted shortest first, dB normalised, per-utterance features"""
    from mvector.data_utils.audio import AudioSegment  # host-side decode shared with the product (I/O, not the hot path)

    def read(list_path):
        rows = [l.strip().split('\t') for l in open(list_path) if l.strip()]
        segs = []
        for path, lab in rows:
            s = AudioSegment.from_file(path)
            dur = s.duration
            s.normalize(target_db=-20)
            segs.append((dur, torch.tensor(s.samples, dtype=torch.float32), int(lab)))
        order = np.argsort([d for d, _, _ in segs], kind='stable')
        return [segs[i][1] for i in order], np.array([segs[i][2] for i in order])

    ew, el = read(cfg['dataset_conf']['enroll_list'])
    tw, tl = read(cfg['dataset_conf']['trials_list'])
    return oeval.evaluate(sd, 'TDNN', ew, el, tw, tl, batch_size=4, method='Fbank', method_args=FB)


def test_trainer_evaluate_cpu_matches_oracle(tmp_path):
    from mvector.trainer import MVectorTrainer
    cfg, model_dir, sd = _make_eval_set(tmp_path, num_workers=2)
    eer, dcf, thr = MVectorTrainer(cfg, use_gpu=False).evaluate(resume_model=model_dir)
    o_eer, o_dcf, o_thr, _ = _oracle_eval(cfg, sd)
    assert abs(eer - o_eer) < 1e-6 and abs(dcf - o_dcf) < 1e-6 and abs(thr - o_thr) < 1e-4
    t = MVectorTrainer(cfg, use_gpu=False)
    for fn in (t.train, t.export, t.extract_features):
        with pytest.raises(NotImplementedError):
            fn()


def test_cpu_forward_varlen_equals_per_utterance_featurisation():
    from mvector.data_utils.featurizer import AudioFeaturizer
    fz = AudioFeaturizer('Fbank', method_args=FB)
    wav = frontend.synth_waveforms(3, 8000, seed=3)
    lens = torch.tensor([8000, 5000, 430])
    padded = wav.clone()
    for i, n in enumerate(lens):
        padded[i, n:] = 0
    out = fz.forward_varlen(padded, lens)
    for i, n in enumerate(lens):
        ref = frontend.audio_featurizer(wav[i, :n].unsqueeze(0), None, 'Fbank', FB)[0]
        assert (out[i, :ref.shape[0]] - ref).abs().max() < 1e-4 and out[i, ref.shape[0]:].abs().sum() == 0


@pytest.mark.gpu
def test_gpu_fbank_varlen_matches_per_utterance_oracle():
    from mvector import _hip
    fb = _hip.Fbank(FB)
    wav = frontend.synth_waveforms(6, 48000, seed=21)
    lens = torch.tensor([48000, 47999, 30123, 16000, 400, 399])  # full, ragged, exactly one frame, too short for a frame
    padded = wav.clone()
    for i, n in enumerate(lens):
        padded[i, n:] = 0.37  # anything behind the true length must be ignored
    out = fb(padded.to(DEV), None, lens.to(DEV)).cpu()
    assert out.shape == (6, 298, 80)
    for i, n in enumerate(lens.tolist()):
        if n < 400:
            assert out[i].abs().sum() == 0
            continue
        ref = frontend.audio_featurizer(wav[i, :n].unsqueeze(0), None, 'Fbank', FB)[0]
        d = (out[i, :ref.shape[0]] - ref).abs()
        assert d.max() < 2e-3 and d.mean() < 2e-5, (i, d.max().item())
        assert out[i, ref.shape[0]:].abs().sum() == 0


@pytest.mark.gpu
def test_gpu_trainer_evaluate_matches_oracle(tmp_path):
    from mvector.trainer import MVectorTrainer
    cfg, model_dir, sd = _make_eval_set(tmp_path, n_spk=4, per_spk=4, seed=9, num_workers=2)  # decode in forked workers, embed on the GPU
    eer, dcf, thr = MVectorTrainer(cfg, use_gpu=True).evaluate(resume_model=model_dir)
    o_eer, o_dcf, o_thr, _ = _oracle_eval(cfg, sd)
    # fp16 activations move individual scores by ~1e-4; the rank-based metrics move only if two trials swap order
    assert abs(eer - o_eer) < 2e-2 and abs(dcf - o_dcf) < 5e-2 and abs(thr - o_thr) < 5e-3, (eer, o_eer, dcf, o_dcf, thr, o_thr)
