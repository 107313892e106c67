"""Generate tests/golden/* by running the REFERENCE's own modules (oracle/test infrastructure only).

Run in the build container, where /root/reference exists:

    python oracle/make_golden.py

It imports ``mvector.models`` straight from /root/reference (``loguru`` is not installed, so a
no-op stub is injected; nothing else is patched), loads the seeded weights of oracle/weights.py
into the reference modules and stores

* ``manifest_<case>.json``  -- the reference ``state_dict`` keys -> shapes (the load contract),
* ``<case>.npz``            -- input features and the reference's output embedding (+ a few
                               intermediate activations for the small cases),
* ``cosine.npz``            -- sklearn ``cosine_similarity`` on seeded embeddings,
* ``metrics.npz``           -- the reference's ``compute_fnr_fpr`` / ``compute_eer`` / ``compute_dcf`` on seeded trial scores,
* ``frontend.npz``          -- front-end outputs.  torchaudio cannot be imported here, so these
                               come from the oracle restatement itself cross-checked (at
                               generation time, asserted below) against
                               ``transformers.audio_utils`` -- see oracle/__init__.py.

The GPU box has no /root/reference: tests only read the committed files.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'

CASES = {
    # name: (model class, ctor kwargs, input [B, T, F], weight seed)
    'ecapa_tiny': ('EcapaTdnn', dict(input_size=80, channels=[64, 64, 64, 64, 192]), (3, 50, 80), 3),
    'ecapa_c512': ('EcapaTdnn', dict(input_size=80), (2, 298, 80), 0),
    'ecapa_c1024': ('EcapaTdnn', dict(input_size=80, channels=[1024, 1024, 1024, 1024, 3072]), (2, 298, 80), 0),
    'ecapa_mel128': ('EcapaTdnn', dict(input_size=128), (2, 241, 128), 0),
    'campp': ('CAMPPlus', dict(input_size=80, embd_dim=192), (2, 298, 80), 0),
    'campp_short': ('CAMPPlus', dict(input_size=80, embd_dim=192), (3, 121, 80), 5),
    'tdnn': ('TDNN', dict(input_size=80), (4, 98, 80), 0),
}


def import_reference_models():
    stub = types.ModuleType('loguru')

    class _Logger:
        def __getattr__(self, _):
            return lambda *a, **k: None

    stub.logger = _Logger()
    sys.modules.setdefault('loguru', stub)
    sys.path.insert(0, REF)
    import mvector.models as ref_models  # noqa: E402  (the reference package)
    assert ref_models.__file__.startswith(REF)
    return ref_models


def main():
    sys.path.insert(0, ROOT)
    from oracle import frontend, weights, models as omodels
    os.makedirs(GOLDEN, exist_ok=True)
    ref_models = import_reference_models()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())

    # ---- front-end fixtures -------------------------------------------------------------
    from transformers.audio_utils import spectrogram, mel_filter_bank
    fb_args = dict(sample_frequency=16000, num_mel_bins=80)
    wav = frontend.synth_waveforms(4, 48000)
    feats = frontend.audio_featurizer(wav, None, 'Fbank', fb_args)
    mf = mel_filter_bank(257, 80, 20, 8000, 16000, norm=None, mel_scale='kaldi', triangularize_in_mel_space=True)
    win = frontend.povey_window(400).numpy().astype(np.float64)
    for b in range(2):
        hf = spectrogram(wav[b].numpy().astype(np.float64), win, frame_length=400, hop_length=160, fft_length=512,
                         power=2.0, center=False, preemphasis=0.97, mel_filters=mf, log_mel='log',
                         mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
        raw = frontend.kaldi_fbank(wav[b:b + 1], **fb_args).numpy()
        assert np.abs(raw - hf).max() < 1e-3, np.abs(raw - hf).max()
    # padded variable-length batch: exercises quirks Q1 (log floor), Q2 (CMN over padded frames), Q3 (round-half-even)
    lens = [48000, 24080, 30000, 16000 + 7]
    wav_var = torch.zeros(4, 48000)
    for i, n in enumerate(lens):
        wav_var[i, :n] = wav[i, :n] * (1e-4 if i == 3 else 1.0)
    ratio = torch.tensor([n / 48000 for n in lens], dtype=torch.float32)
    feats_var = frontend.audio_featurizer(wav_var, ratio, 'Fbank', fb_args)
    mel = frontend.audio_featurizer(wav, None, 'MelSpectrogram', {})
    mel_var = frontend.audio_featurizer(wav_var, ratio, 'MelSpectrogram', {})
    mel_readme_args = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000,
                           n_mels=64)  # README_en.md:263-269
    mel_readme = frontend.audio_featurizer(wav[:2], None, 'MelSpectrogram', mel_readme_args)
    np.savez_compressed(os.path.join(GOLDEN, 'frontend.npz'),
                        wav_seed=1234, fbank=feats.numpy(), lens=np.array(lens), ratio=ratio.numpy(),
                        fbank_var=feats_var.numpy(), mel=mel.numpy()[:2], mel_var=mel_var.numpy()[2:],
                        mel_readme=mel_readme.numpy()[:1])
    print('frontend.npz', feats.shape, feats_var.shape, mel.shape)

    # real audio (first 16000 samples of the reference's dataset/*.wav: SURVEY.md 8(d) config 1)
    import scipy.io.wavfile as wavfile
    real = []
    for name in ('a_1', 'a_2', 'b_1', 'b_2'):
        sr, d = wavfile.read(os.path.join(REF, 'dataset', f'{name}.wav'))
        assert sr == 16000 and d.dtype == np.int16
        real.append(d[:16000].copy())
    real = np.stack(real)
    real_f = torch.from_numpy(real.astype(np.float32) / 32768.0)
    real_feats = frontend.audio_featurizer(real_f, None, 'Fbank', fb_args)
    np.savez_compressed(os.path.join(GOLDEN, 'real_audio.npz'), pcm16=real, fbank=real_feats.numpy())

    # ---- model fixtures (reference modules are the source of truth) --------------------
    g = torch.Generator().manual_seed(99)
    for case, (cls, kwargs, in_shape, wseed) in CASES.items():
        model = getattr(ref_models, cls)(**kwargs)
        shapes = weights.shapes_of(model.state_dict())
        with open(os.path.join(GOLDEN, f'manifest_{case}.json'), 'w') as f:
            json.dump(dict(model=cls, kwargs=kwargs, seed=wseed, shapes={k: list(v) for k, v in shapes.items()}), f,
                      indent=0)
        sd = weights.make_state_dict(shapes, wseed)
        missing, unexpected = model.load_state_dict(sd, strict=True)
        model.eval()
        B, T, Fdim = in_shape
        if Fdim == 80 and T in (298, 98):
            x = (feats if T == 298 else real_feats)[:B].clone()
        else:
            x = torch.randn(in_shape, generator=g) * 2.0
            x = x - x.mean(1, keepdim=True)
        with torch.no_grad():
            emb = model(x)
            oemb = omodels.FORWARDS[cls](sd, x)
        err = (emb - oemb).abs().max().item()
        print(f'{case}: emb {tuple(emb.shape)} |emb| {emb.abs().mean():.4f} oracle-vs-reference max abs {err:.3e} '
              f'params {sum(v.numel() for v in sd.values()) / 1e6:.2f} M')
        extra = {}
        if case == 'ecapa_tiny':
            # intermediate activations of the reference modules, for layer-level kernel tests
            with torch.no_grad():
                h = x.transpose(1, 2)
                h0 = model.blocks[0](h)
                blk = model.blocks[1]
                t1 = blk.tdnn1(h0)
                r2 = blk.res2net_block(t1)
                t2 = blk.tdnn2(r2)
                b1 = blk(h0)
                extra = dict(l_block0=h0.numpy(), l_tdnn1=t1.numpy(), l_res2=r2.numpy(), l_tdnn2=t2.numpy(),
                             l_block1=b1.numpy())
        np.savez_compressed(os.path.join(GOLDEN, f'{case}.npz'), x=x.numpy(), emb=emb.numpy(), **extra)

    # ---- cosine fixture ----------------------------------------------------------------
    from sklearn.metrics.pairwise import cosine_similarity
    rng = np.random.default_rng(7)
    a = rng.normal(size=(37, 192)).astype(np.float32)
    b = rng.normal(size=(53, 192)).astype(np.float32) * 3.0
    np.savez_compressed(os.path.join(GOLDEN, 'cosine.npz'), a=a, b=b, sim=cosine_similarity(a, b))
    print('done ->', GOLDEN)


def save_metrics_golden():
    """mvector/metric/metrics.py of the reference, loaded by path (it only needs numpy / torch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_metrics', os.path.join(REF, 'mvector', 'metric', 'metrics.py'))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    rng = np.random.default_rng(7)
    out = {}
    for name, n, p_tgt in (('small', 60, 0.3), ('large', 20000, 0.05)):
        labels = (rng.random(n) < p_tgt).astype(np.int32)
        scores = (rng.normal(0.15, 0.18, n) + labels * 0.4).astype(np.float32)
        fnr, fpr, thr = rm.compute_fnr_fpr(scores, labels)
        eer, eer_thr = rm.compute_eer(fnr, fpr, scores)
        out.update({f'{name}_scores': scores, f'{name}_labels': labels, f'{name}_fnr': fnr, f'{name}_fpr': fpr,
                    f'{name}_thresholds': thr, f'{name}_eer': np.float64(eer), f'{name}_eer_threshold': np.float64(eer_thr),
                    f'{name}_min_dcf': np.float64(rm.compute_dcf(fnr, fpr))})
    np.savez_compressed(os.path.join(GOLDEN, 'metrics.npz'), **out)


if __name__ == '__main__':
    main()
    save_metrics_golden()
