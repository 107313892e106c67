"""GPU parity tests (-m gpu): the HIP path through the C ABI against the oracle and the committed golden
vectors.  Tolerances: Fbank features max-abs 2e-3 (both fp32 evaluations sit ~3e-4 from an fp64 evaluation near
the log floor) and mean-abs 2e-5; embeddings 1 - cos <= 1e-4 (north_star); cosine scores 2e-6."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import layer_checks as lc
from conftest import ROOT
from helpers import GOLDEN, cos_dist, load_case
from oracle import frontend, models as omodels, scoring

pytestmark = pytest.mark.gpu
FB = dict(sample_frequency=16000, num_mel_bins=80)
DEV = 'cuda'


def product_lib():
    from mvector import _hip
    return _hip.lib()


def test_gpu_library_is_the_hip_build():
    from mvector import _hip
    lib = product_lib()
    assert lib.mv_abi_version() == 5
    assert os.path.basename(_hip.LIB_PATH) == 'libmvector_hip.so'


@pytest.mark.parametrize('idx', range(len(lc.CONV2DS_CASES)))
def test_conv2ds(idx):
    """the split-fp16 form of the ERes2Net conv layers on S16 maps, through the C ABI"""
    lc.conv2ds_case(product_lib(), DEV, seed=idx, **lc.CONV2DS_CASES[idx])


def test_conv2ds_full_size_layers():
    """stage-1 / stage-4 shapes of the 54.9 M ERes2NetV2 (LDS-DMA rings under real latencies: many tiles, long K loops)"""
    lc.conv2ds_case(product_lib(), DEV, B=2, H=80, W=298, cin=48, cout=48, ks=3, with_sum=True, seed=30)
    lc.conv2ds_case(product_lib(), DEV, B=2, H=80, W=298, cin=192, cout=192, ks=1, with_res=True, seed=31)
    lc.conv2ds_case(product_lib(), DEV, B=3, H=10, W=38, cin=1280, cout=1536, ks=1, with_res=True, seed=32)
    lc.conv2ds_case(product_lib(), DEV, B=3, H=10, W=38, cin=320, cout=320, ks=3, seed=33)
    lc.conv2ds_case(product_lib(), DEV, B=2, H=20, W=75, cin=768, cout=1536, ks=3, stride=2, lo=-65504.0, hi=65504.0, seed=34)


def test_tstp_and_first_conv():
    lc.tstp_case(product_lib(), DEV)
    lc.conv2d_first_case(product_lib(), DEV, B=3, T=298, F_=80, C=32)
    lc.tstp_case(product_lib(), DEV, s16=True)
    lc.conv2d_first_case(product_lib(), DEV, B=3, T=298, F_=80, C=32, s16=True)


@pytest.mark.parametrize('cfg', [dict(B=2, T=70, cout=64), dict(B=3, T=300, cout=512), dict(B=5, T=298, cout=1024, F_=128, tile=256)])
def test_conv1d_window(cfg):
    lc.conv1d_window_case(product_lib(), DEV, **cfg)


@pytest.mark.parametrize('cfg', [dict(k=1, dil=1, cin=128, cout=256, T=150, B=2, tile=256), dict(k=1, dil=1, cin=1024, cout=1024, T=298, B=64),
                                 dict(k=1, dil=1, cin=128, cout=256, T=150, B=2, tile=256, post_act=1),
                                 dict(k=3, dil=2, cin=64, cout=128, T=70, B=2), dict(k=1, dil=1, cin=64, cout=64, T=33, B=1)])
def test_gpu_conv1d_saturates_at_the_fp16_range(cfg):
    """BatchNorm scales of 1e5: fp16 outputs saturate at +-65504 and never become infinite -- the ring kernel through MODE.FP16_OVFL
    (tools/fp16_ovfl_probe.hip), every other kernel through its v_med3"""
    lc.conv1d_case(product_lib(), DEV, seed=5, out_gain=1.0e5, **cfg)


def test_gpu_profile_classes_ring_is_a_subset_of_conv1d():
    lc.profile_classes_case(product_lib(), DEV)


@pytest.mark.parametrize('idx', range(len(lc.CONV_CASES)))
def test_gpu_conv1d(idx):
    lc.conv1d_case(product_lib(), DEV, seed=idx, **lc.CONV_CASES[idx])


def test_gpu_conv1d_persistent_walks_many_tiles():
    """MvConv1dDesc.persist_blocks_hint = 8: eight resident workgroups walk 5-10 tiles each (full tiles, a ragged last one, 2-3 channel tiles), so
    the tile-boundary machinery of the persistent kernels (epilogue inside the next tile's first stage, transfers requested ahead of the output
    stores and the counted wait behind them, parameter reload when the channel tile changes) runs on the device: dense 1x1 layers on the ring
    kernel, a k = 3 layer on the double-buffer kernel."""
    for seed, cfg in enumerate([dict(k=1, dil=1, cin=256, cout=512, T=298, B=16, tile=256),
                                dict(k=1, dil=1, cin=1024, cout=768, T=300, B=11, tile=256),
                                dict(k=3, dil=2, cin=256, cout=512, T=298, B=16, tile=256),
                                dict(k=1, dil=1, cin=128, cout=256, T=64, B=40, tile=256, affine=False)]):
        lc.conv1d_case(product_lib(), DEV, seed=seed, persist_blocks=8, **cfg)


@pytest.mark.parametrize('cfg,blocks,launches', [
    (dict(k=1, dil=1, cin=512, cout=512, T=298, B=19, tile=256), 40, 2),     # 6 of 46 tiles as 24 quarters (ragged last rows)
    (dict(k=1, dil=1, cin=512, cout=768, T=300, B=21, tile=256), 64, 2),     # 11 of 75 tiles as 44 quarters
    (dict(k=1, dil=1, cin=512, cout=512, T=300, B=14, tile=256), 32, 2),     # 2 of 34 tiles as 32 sixteenths
    (dict(k=1, dil=1, cin=3072, cout=3072, T=300, B=128), 0, 2),              # half the headline batch through the MFA layer on the whole chip: 1800 tiles = 7 rounds + 8 tiles as 128 sixteenths
    (dict(k=1, dil=1, cin=1024, cout=1024, T=300, B=256), 0, 1),              # a K = 1024 layer of the headline batch: 176 tiles in the last round, not split
])
def test_gpu_ring_tail_sub_tiles_carry_the_same_bits(cfg, blocks, launches):
    """conv1d_launch runs the ring GEMM walk's last partial round as 64 x 64 or 128 x 128 sub-tiles when they fit one round of the chip: same bits as the
    unsplit launch, the profile classes add up to the layer's FLOPs"""
    n3, n0, w3, w0 = lc.ring_tail_case(product_lib(), DEV, blocks=blocks, **cfg)
    assert (n3, n0) == (1, launches) and (w3 < w0) == (launches == 2)


def test_gpu_conv1d_input_statistics_two_launch_forms_agree():
    """the ASP hidden layer's conv: fused statistics kernel (70 utterances) vs stand-alone statistics + 64 x 64 tiles (1 utterance): same bits"""
    lc.in_stats_forms_case(product_lib(), DEV, B_big=70, T=298, cin=3072)
    lc.in_stats_forms_case(product_lib(), DEV, B_big=72, T=161, cin=1536, seed=1)


@pytest.mark.parametrize('shape', [(5, 100, 37, 1), (17, 64, 16, 0), (1, 7, 3, 3), (256, 6144, 192, 0), (256, 1024, 128, 2), (19, 4180, 40, 0), (250, 4180, 40, 0), (2048, 6144, 192, 0), (256, 6144, 128, 0), (300, 3072, 192, 1), (33, 2500, 17, 2)])
def test_gpu_linear(shape):
    lc.linear_case(product_lib(), DEV, *shape)


def test_gpu_cosine_matches_sklearn_golden():
    z = np.load(os.path.join(GOLDEN, 'cosine.npz'))
    lc.cosine_case(product_lib(), DEV, z['a'], z['b'], z['sim'])


@pytest.mark.parametrize('unbiased,eps', [(0, 1e-12), (1, 0.0)])
def test_gpu_time_stats(unbiased, eps):
    lc.time_stats_case(product_lib(), DEV, unbiased=unbiased, eps=eps)
    lc.time_stats_case(product_lib(), DEV, B=7, T=298, C=3072, ld=3072, unbiased=unbiased, eps=eps, seed=2)


def test_gpu_fbank_long_utterances_chunked_and_single_workgroup_forms():
    """utterances beyond the LDS block with fewer utterances than CUs: several workgroups per utterance + the finish pass (caller workspace), or one
    workgroup per utterance (no workspace); 10 s x 3 with ragged lengths, 30 s x 1, and 128 x 6 s (half a chip of utterances) against the oracle and
    BIT-IDENTICAL to each other (the time sum of an utterance is formed in an order fixed by its own length, csrc/fbank.hip FbankArgs)"""
    from mvector import _hip
    FB = dict(sample_frequency=16000, num_mel_bins=80)
    wav = frontend.synth_waveforms(3, 400 + 160 * 999, seed=31)
    ratio = torch.tensor([0.41, 1.0, 0.77])
    lc.fbank_case(product_lib(), DEV, wav, ratio, FB)
    lc.fbank_case(product_lib(), DEV, frontend.synth_waveforms(1, 480000, seed=32), None, FB)
    big = (0.1 * torch.randn(128, 96000, generator=torch.Generator().manual_seed(33))).to(DEV)
    fb = _hip.Fbank(FB)
    a = fb(big)
    b = fb(big, workspace=False)
    assert torch.equal(a, b)
    w, r = wav.to(DEV), ratio.to(DEV)
    assert torch.equal(fb(w, r), fb(w, r, workspace=False))


def test_gpu_fbank_row_bits_do_not_depend_on_the_batch_size():
    """featurizer.py:125-130 computes every row on its own: row i of a batch must carry the same bits at B = 1 / 32 / 128 / 256 (the forms the
    launcher picks by batch size -- one workgroup per utterance on a full chip, about CUs / B workgroups per utterance below -- sum the time
    mean in one order)"""
    from mvector import _hip
    fb = _hip.Fbank(FB)
    wav = frontend.synth_waveforms(256, 48000, seed=5).to(DEV)
    full = fb(wav)
    for nb in (1, 32, 128, 255):
        assert torch.equal(fb(wav[:nb]), full[:nb]), nb
        assert torch.equal(fb(wav[256 - nb:]), full[256 - nb:]), nb
    ratio = torch.linspace(0.3, 1.0, 256).to(DEV)
    fullr = fb(wav, ratio)
    for nb in (1, 32, 128):
        assert torch.equal(fb(wav[:nb], ratio[:nb]), fullr[:nb]), nb
    nsamp = torch.full((32,), 48000, dtype=torch.int64, device=DEV)
    assert torch.equal(fb(wav[:32], num_samples=nsamp), full[:32])   # the variable-length entry point too
    long = frontend.synth_waveforms(40, 400 + 160 * 700, seed=6).to(DEV)   # 7 s: beyond the LDS block of the one-workgroup form
    fl = fb(long)
    for nb in (1, 7):
        assert torch.equal(fb(long[:nb]), fl[:nb]), nb


def test_gpu_fbank_one_handle_two_streams_is_reentrant():
    """include/mvector_hip.h: handles are immutable after create -- ONE Fbank handle driven from two streams at once (sub-chip batches of 3 s
    utterances: the several-workgroups form with its per-call scratch) gives, 100 times over, the bits of the serial run"""
    from mvector import _hip
    fb = _hip.Fbank(FB)
    wa = frontend.synth_waveforms(32, 48000, seed=11).to(DEV)
    wb = frontend.synth_waveforms(48, 48000 + 160 * 300, seed=12).to(DEV)
    ra, rb = fb(wa).clone(), fb(wb).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for it in range(100):
        with torch.cuda.stream(s1):
            oa = fb(wa)
        with torch.cuda.stream(s2):
            ob = fb(wb)
        s1.synchronize()
        s2.synchronize()
        bad += int(not torch.equal(oa, ra)) + int(not torch.equal(ob, rb))
    assert bad == 0, bad


def test_gpu_bn_relu_rows():
    lc.bn_relu_rows_case(product_lib(), DEV)
    lc.bn_relu_rows_case(product_lib(), DEV, rows=38144, C=1024, ldx=1024, ldy=1024, seed=3)


def test_gpu_fbank_golden_fixed_and_ragged():
    z = np.load(os.path.join(GOLDEN, 'frontend.npz'))
    wav = frontend.synth_waveforms(4, 48000)
    from mvector import _hip
    fb = _hip.Fbank(FB)
    out = fb(wav.to(DEV)).cpu().numpy()
    d = np.abs(out - z['fbank'])
    assert d.max() < 2e-3 and d.mean() < 2e-5, (d.max(), d.mean())
    lens = z['lens']
    wav_var = torch.zeros(4, 48000)
    for i, n in enumerate(lens):
        wav_var[i, :n] = wav[i, :n] * (1e-4 if i == 3 else 1.0)
    outv = fb(wav_var.to(DEV), torch.from_numpy(z['ratio']).to(DEV)).cpu().numpy()
    dv = np.abs(outv - z['fbank_var'])
    assert dv.max() < 2e-3 and dv.mean() < 2e-5, (dv.max(), dv.mean())
    for i, n in enumerate(lens):  # frames beyond round_half_even(ratio*T) are exactly zero (Q2/Q3)
        ml = int(torch.round(torch.tensor(n / 48000, dtype=torch.float32) * 298).item())
        assert np.all(outv[i, ml:] == 0) and np.any(outv[i, ml - 1] != 0)


def test_gpu_featurizer_matches_reference_wrapper_golden():
    """AudioFeaturizer.forward on the HIP path against tests/golden/featurizer_ref.npz = the REFERENCE's own featurizer.py wrapper
    (KaldiFbank loop, transposes, CMN, torch.round mask) run by oracle/make_golden.py under a stub torchaudio (a3 pin)."""
    from mvector.data_utils.featurizer import AudioFeaturizer
    z = np.load(os.path.join(GOLDEN, 'featurizer_ref.npz'))
    wav = frontend.synth_waveforms(4, 48000)
    wav_var = torch.zeros(4, 48000)
    for i, n in enumerate(z['lens']):
        wav_var[i, :n] = wav[i, :n] * (1e-4 if i == 3 else 1.0)
    ratio, half = torch.from_numpy(z['ratio']).to(DEV), torch.from_numpy(z['half']).to(DEV)
    fz = AudioFeaturizer('Fbank', method_args=FB)

    def close(got, want):
        d = np.abs(got - want)
        assert d.max() < 2e-3 and d.mean() < 2e-5, (d.max(), d.mean())
        # masked frames are exactly zero, frame by frame, on both sides (Q3: the edge sits where torch.round puts it)
        assert np.array_equal(np.all(got == 0, axis=-1), np.all(want == 0, axis=-1))
    out = fz(wav[:2].to(DEV))
    assert out.is_cuda and out.dtype == torch.float32
    close(out.cpu().numpy(), z['fbank'])
    close(fz(wav_var.to(DEV), ratio).cpu().numpy(), z['fbank_var'])
    close(fz(wav_var.to(DEV), half).cpu().numpy()[:, :24], z['fbank_half'])
    close(fz(wav[1, :16000].to(DEV)).cpu().numpy(), z['fbank_1d'])  # 1-D input is unsqueezed (featurizer.py:63-64)
    mz = AudioFeaturizer('MelSpectrogram', method_args={})
    scale = float(np.abs(z['mel']).max())
    for got, want in ((mz(wav[:2].to(DEV)), z['mel']), (mz(wav_var.to(DEV), ratio)[2:], z['mel_var']),
                      (mz(wav_var[:1].to(DEV), half[:1]), z['mel_half'])):
        got = got.cpu().numpy()
        assert np.abs(got - want).max() < 2e-4 * scale
        assert np.array_equal(np.all(got == 0, axis=-1), np.all(want == 0, axis=-1))
    readme = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000, n_mels=64)
    got = AudioFeaturizer('MelSpectrogram', method_args=readme)(wav[:1].to(DEV)).cpu().numpy()
    assert np.abs(got - z['mel_readme']).max() < 2e-4 * float(np.abs(z['mel_readme']).max())


def test_gpu_fbank_ragged_strides_and_edges():
    wav = frontend.synth_waveforms(3, 16000 + 37, seed=3)
    wav[2, 9000:] = 0
    lc.fbank_case(product_lib(), DEV, wav, torch.tensor([1.0, 0.61, 9000 / 16037]), FB)
    from mvector import _hip
    fb = _hip.Fbank(FB)
    assert fb(torch.zeros(2, 399, device=DEV)).shape == (2, 0, 80)
    assert fb(torch.zeros(1, 1200, device=DEV)).abs().max().item() < 1e-5
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    real = torch.from_numpy(z['pcm16'].astype(np.float32) / 32768.0).to(DEV)
    assert np.abs(fb(real).cpu().numpy() - z['fbank']).max() < 2e-3


@pytest.mark.parametrize('kernel', ['generic', 'tile'])
def test_gpu_fbank_both_kernels_on_the_committed_goldens(kernel):
    """fbank_kernel and fbank_tile_kernel (MvFbankCfg.kernel) each against the committed fixtures: frontend.npz (fixed + ragged), the real-audio
    golden, and the full 256 x 3 s batch against the fp64 arbiter at the stated 1e-3 (layer_checks.fbank_within_stated_bar)"""
    from mvector import _hip
    fb = _hip.Fbank(FB, kernel=kernel)
    assert fb.info()['tile_kernel'] == (kernel == 'tile')
    z = np.load(os.path.join(GOLDEN, 'frontend.npz'))
    wav = frontend.synth_waveforms(4, 48000)
    d = np.abs(fb(wav.to(DEV)).cpu().numpy() - z['fbank'])
    assert d.max() < 2e-3 and d.mean() < 2e-5, (d.max(), d.mean())
    wav_var = torch.zeros(4, 48000)
    for i, n in enumerate(z['lens']):
        wav_var[i, :n] = wav[i, :n] * (1e-4 if i == 3 else 1.0)
    dv = np.abs(fb(wav_var.to(DEV), torch.from_numpy(z['ratio']).to(DEV)).cpu().numpy() - z['fbank_var'])
    assert dv.max() < 2e-3 and dv.mean() < 2e-5, (dv.max(), dv.mean())
    zr = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    real = torch.from_numpy(zr['pcm16'].astype(np.float32) / 32768.0).to(DEV)
    assert np.abs(fb(real).cpu().numpy() - zr['fbank']).max() < 2e-3
    big = frontend.synth_waveforms(256, 48000)
    out = fb(big.to(DEV)).cpu()
    ref64 = frontend.audio_featurizer_fbank_f64(big, None, FB)
    e = lc.fbank_within_stated_bar(out, ref64)
    print(f'{kernel}: 256 x 3 s |HIP - f64| max {e.max().item():.3e} mean {e.mean().item():.3e} n>1e-3 {int((e > 1e-3).sum())}')


@pytest.mark.parametrize('idx', range(len(lc.FBANK_ARG_CASES)))
def test_gpu_fbank_arguments(idx):
    """HIP vs the fp32 oracle AND the fp64 arbiter over the kaldi.fbank keyword arguments featurizer.py:128 forwards (tests/layer_checks.py::
    FBANK_ARG_CASES: frame_length 20 / 24 / 25 / 30 / 32 ms on BOTH kernels -- fbank_tile_kernel<13> and <16>, fbank_kernel --, frame_shift 10 / 12.5,
    23 / 40 / 64 / 80 / 128 bins, 8 / 11.025 / 16 / 22.05 kHz, band edges, use_power / use_log_fbank / remove_dc_offset / preemphasis switches,
    the five window types, snip_edges=False, subtract_mean, min_duration, VTLN warps; bare kaldi.fbank rows and true-length batches): 5 x 3 s with a ragged
    mask, and 260 x 0.5 s (more utterances than CUs)"""
    lc.fbank_arguments_case(product_lib(), DEV, idx, B=5, seconds=3.0)
    lc.fbank_arguments_case(product_lib(), DEV, idx, B=260, seconds=0.5, seed=7, check_rows=[0, 1, 2, 3, 4, 130, 255, 256, 257, 258, 259])


def test_gpu_fbank_clip_of_exactly_min_duration_keeps_its_frames():
    """L == min_duration * sample_frequency is featurised (torchaudio's double compare), L - 1 is not -- both kernels, device"""
    lc.fbank_min_duration_edge(product_lib(), DEV)


def test_gpu_kaldi_fbank_module_matches_oracle():
    """KaldiFbank (featurizer.py:114-132: kaldi.fbank per utterance, [B, F, T], NO time mean) on CUDA tensors against oracle.frontend.kaldi_fbank;
    [B, 1, L] rows as the reference's own caller passes them; the default 23 bins (generic kernel) and 80 bins (tile kernel)"""
    from mvector.data_utils.featurizer import KaldiFbank
    wav = frontend.synth_waveforms(6, 48000, seed=41)
    for kwargs in (FB, dict(sample_frequency=16000), dict(sample_frequency=8000, num_mel_bins=40, frame_length=30)):
        mod = KaldiFbank(**kwargs)
        out = mod(wav.to(DEV))
        F_ = kwargs.get('num_mel_bins', 23)
        ref = torch.stack([frontend.kaldi_fbank(w.unsqueeze(0), **kwargs).t() for w in wav])
        ref64 = torch.stack([frontend.kaldi_fbank_f64(w.unsqueeze(0), **kwargs).t() for w in wav])
        assert out.is_cuda and out.shape == ref.shape == (6, F_, ref.shape[2])
        assert (out.cpu() - ref).abs().max().item() < 2e-3
        lc.fbank_within_stated_bar(out.cpu(), ref64)
        assert torch.equal(mod(wav.to(DEV).unsqueeze(1)), out)
        assert torch.allclose(mod(wav), ref, atol=1e-4)   # CPU tensors: the batched torch restatement


def test_gpu_fbank_full_batch_properties():
    """BASELINE size (256 x 3 s): size-independent properties + EVERY row against the oracle."""
    from mvector import _hip
    fb = _hip.Fbank(FB)
    wav = frontend.synth_waveforms(256, 48000).to(DEV)
    out = fb(wav)
    assert out.shape == (256, 298, 80)
    assert out.mean(1).abs().max().item() < 2e-4            # CMN: zero time-mean per (utterance, bin)
    assert torch.equal(out, fb(wav))                         # deterministic
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).to(DEV)
    assert torch.equal(fb(wav[perm]), out[perm])             # utterances are independent
    ref = frontend.audio_featurizer(wav.cpu(), None, 'Fbank', FB)
    err = (out.cpu() - ref).abs()
    # 6.1 M log energies.  The stated bar is max-abs <= 1e-3.  Against the fp64 ARBITER of the same algorithm
    # (oracle.frontend.kaldi_fbank_f64) the kernel meets it; the torch-fp32 oracle itself sits 1.12e-3 from the arbiter on this batch (two
    # near-floor bins above 1e-3), which is why kernel-vs-fp32-oracle (two fp32 evaluations of a small difference of fp32 spectra) shows
    # up to 2.5e-3 on a handful of bins: that distance is asserted as a distribution only.
    ref64 = frontend.audio_featurizer_fbank_f64(wav.cpu(), None, FB)
    e_hip, e_o32 = (out.cpu().double() - ref64).abs(), (ref.double() - ref64).abs()
    print(f'fbank 256 x 3 s: |HIP - f64| max {e_hip.max().item():.3e} mean {e_hip.mean().item():.3e} n>1e-3 {int((e_hip > 1e-3).sum())}; '
          f'|oracle32 - f64| max {e_o32.max().item():.3e} mean {e_o32.mean().item():.3e} n>1e-3 {int((e_o32 > 1e-3).sum())}; '
          f'|HIP - oracle32| max {err.max().item():.3e} mean {err.mean().item():.3e} n>2e-3 {int((err > 2e-3).sum())}')
    assert e_hip.max().item() <= 1e-3, e_hip.max().item()
    assert e_hip.mean().item() <= 1e-5
    assert err.mean().item() < 2e-5 and (err > 2e-3).float().mean().item() < 1e-5 and err.max().item() < 5e-3, (err.mean().item(), err.max().item())


def test_gpu_fbank_gain_invariance_full_batch():
    """log-mel + time-mean subtraction: a gain on the waveform shifts every log energy by the same constant, which the CMN
    removes -- except where the quieter copy falls onto the log floor (Q1: [-1, 1]-scaled input, a few low bins), which
    also moves that bin's time mean.  So: most (utterance, bin) columns are invariant, and where they are not the kernel
    moves exactly as the oracle does."""
    from mvector import _hip
    fb = _hip.Fbank(FB)
    wav = 0.1 * torch.randn(256, 48000, generator=torch.Generator().manual_seed(4))
    a, b = fb(wav.to(DEV)).cpu(), fb((wav * 0.25).to(DEV)).cpu()
    col = (a - b).abs().amax(1)                     # [256, 80] worst frame per (utterance, bin)
    assert (col < 2e-3).float().mean().item() > 0.9
    ra = frontend.audio_featurizer(wav[:8], None, 'Fbank', FB)
    rb = frontend.audio_featurizer(wav[:8] * 0.25, None, 'Fbank', FB)
    assert ((a[:8] - b[:8]) - (ra - rb)).abs().max().item() < 3e-3


@pytest.mark.parametrize('cfg', [dict(), dict(normalize=False), dict(B=256, L=48000, seed=2)])
def test_gpu_wave_prepare_int16(cfg):
    lc.wave_prepare_case(product_lib(), DEV, **cfg)


def test_gpu_cosine_properties_full_size():
    """config 4 scoring shape: [2048, 192] x [2048, 192]"""
    from mvector import _hip
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2048, 192, generator=g).to(DEV)
    s = _hip.cosine(x, x)
    assert (s.diagonal() - 1).abs().max().item() < 2e-6
    assert (s - s.t()).abs().max().item() < 2e-6                       # symmetric
    assert (_hip.cosine(3.0 * x, 0.5 * x) - s).abs().max().item() < 2e-6  # scale invariant
    ref = scoring.cosine_similarity(x[:64].cpu().numpy(), x.cpu().numpy())
    assert np.abs(s[:64].cpu().numpy() - ref).max() < 2e-6


@pytest.mark.parametrize('case', ['ecapa_tiny', 'ecapa_c512', 'ecapa_c1024', 'ecapa_mel128', 'tdnn', 'campp', 'campp_short', 'campp_c64'])
def test_gpu_native_model_matches_reference_golden(case):
    # (campp_c64: CAMPPlus(init_channels=64) -- the first dense block is narrower than the per-block kernel takes: per-layer kernel)
    cd, rel = lc.model_case(product_lib(), DEV, case)
    assert cd < 1e-4 and rel < 1.4e-2, (cd, rel)


@pytest.mark.parametrize('case', ['ecapa_c1024', 'tdnn', 'campp', 'eres2netv2_tiny'])
def test_gpu_embedding_bits_do_not_depend_on_the_batch_size(case):
    """the reference embeds every utterance on its own data: whatever launch forms the batch size selects (64 x 64 / 128 x 128 / 256 x 256 conv
    tiles -- one accumulation order --, fused or stand-alone ASP input statistics, Fbank chunk form), row i carries the same bits alone, in a
    small batch and on a full chip"""
    from mvector.data_utils.featurizer import AudioFeaturizer
    import mvector.models as M
    man, sd, _, _, _ = load_case(case)
    m = getattr(M, man['model'])(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    fz = AudioFeaturizer('Fbank', method_args=dict(sample_frequency=16000, num_mel_bins=man['kwargs']['input_size']))
    wav = frontend.synth_waveforms(256, 48000, seed=8).to(DEV)
    with torch.no_grad():
        full = m(fz(wav))
        for nb in (1, 8, 40, 130):
            e = m(fz(wav[:nb]))
            assert torch.equal(e, full[:nb]), (case, nb, (e - full[:nb]).abs().max().item())


def test_gpu_fp16_backbone_stress_golden_ecapa():
    """fp16 stress: conv output channels scaled over 10^-1.5 .. 10^0.5, BatchNorm gains up to 3, running
    statistics calibrated by the reference in training mode (running_var from 0 -- dead ReLU channels -- to ~80, median ~0.1).
    The embeddings come from the reference modules (fp32 vs fp64 of the reference itself: 1e-10).  EcapaTdnn stays inside the bar."""
    cd, rel = lc.model_case(product_lib(), DEV, 'ecapa_stress', tol=1e-4)
    print(f'ecapa_stress: 1 - cos = {cd:.3e}, max rel err {rel:.3e}')


def test_gpu_campp_stress_golden_takes_the_fp32_head():
    """The same stress on CAM++.  The FCM head of this checkpoint amplifies a relative perturbation ~100 x: with
    fp16 maps and tap matrices the embedding lands at 1 - cos = 4e-4, every one of the head's ~20 rounding sites contributing
    (tests/budget_campp.py, profiles/r06_campp_error_budget.log).  The handle measures that at creation -- both heads embed three probe
    utterances -- and takes the fp32 head (conv2d kernels): the golden is met at the north-star bar.  Pinned onto the fp16 head
    (MvCamppCfg.head_precision) the same checkpoint shows the miss the calibration saw."""
    info = {1: None, 2: None, 3: None, 4: None, 5: None}
    cd, rel = lc.model_case(product_lib(), DEV, 'campp_stress', tol=1e-4, info=info)
    print(f'campp_stress: head fp32 = {info[1]}, calibration 1 - cos = {info[2]:.3e} (probes {info[3]:.2e} {info[4]:.2e} {info[5]:.2e}); '
          f'golden 1 - cos = {cd:.3e}, max rel err {rel:.3e}')
    assert info[1] == 1.0 and info[2] > 5e-6 and info[2] == max(info[3], info[4], info[5])
    info16 = {1: None, 2: None}
    cd16, _ = lc.model_case(product_lib(), DEV, 'campp_stress', tol=1e-3, info=info16, head=1)
    assert info16[1] == 0.0 and info16[2] == -1.0 and cd16 > cd
    print(f'campp_stress pinned onto the fp16 head: 1 - cos = {cd16:.3e}')
    # (r12b: probes 7.4e-5 / 8.3e-5 / 4.4e-6 against 6.6e-4 on the test input -- an 8 x under-read, the worst seen; the 5e-6 threshold leaves the
    #  1e-4 bar a factor 20)
    assert info[2] > cd16 / 12, 'the probes under-read the miss on the test input by more than the threshold allows for'


@pytest.mark.parametrize('case', ['campp_mid_a', 'campp_mid_b'])
def test_gpu_campp_borderline_checkpoints_meet_the_bar_under_the_automatic_choice(case):
    """checkpoints BETWEEN the trained-like and the stress golden (BatchNorm statistics interpolated between independent draws and the
    train-mode calibration, oracle/make_golden.py `mid`): the fp16 head misses the embedding by a figure around the 5e-6 threshold (a) and just
    below the 1e-4 bar (b).  Whatever head the probes pick, the golden must be met at 1e-4; pinned to either head the figures are printed."""
    info = {1: None, 2: None, 3: None, 4: None, 5: None}
    cd, _ = lc.model_case(product_lib(), DEV, case, tol=1e-4, info=info)
    cd16, _ = lc.model_case(product_lib(), DEV, case, tol=1e-3, head=1)
    cd32, _ = lc.model_case(product_lib(), DEV, case, tol=1e-4, head=2)
    print(f'{case}: automatic head fp32 = {info[1]} (probes {info[3]:.2e} {info[4]:.2e} {info[5]:.2e}) 1 - cos {cd:.2e}; fp16 head {cd16:.2e}, fp32 head {cd32:.2e}')
    assert info[1] == 1.0 or cd16 < 1e-4


def test_gpu_model_info_is_model_specific():
    """mv_model_info: the CAM++ keys do not exist on other backbones (error, not a silent zero)"""
    info = {1: None}
    with pytest.raises(RuntimeError, match='no such key'):
        lc.model_case(product_lib(), DEV, 'ecapa_tiny', info=info)


@pytest.mark.parametrize('case', ['campp', 'campp_short'])
def test_gpu_campp_well_conditioned_checkpoints_keep_the_fp16_head(case):
    """trained-like BatchNorm gains: the calibration difference is ~1e-7 on every probe, the handle keeps the fp16 head (the fast one); pinned
    onto the fp32 head the golden is met as well (the two heads are the same function)"""
    info = {1: None, 2: None}
    cd, _ = lc.model_case(product_lib(), DEV, case, info=info)
    assert info[1] == 0.0 and 0.0 <= info[2] < 5e-6, info
    info32 = {1: None}
    cd32, _ = lc.model_case(product_lib(), DEV, case, info=info32, head=2)
    assert info32[1] == 1.0 and cd32 < 1e-5, (cd32, info32)
    print(f'{case}: fp16 head 1 - cos {cd:.2e} (calibration {info[2]:.2e}), fp32 head {cd32:.2e}')


@pytest.mark.parametrize('case', ['eres2net_tiny', 'eres2netv2_tiny', 'eres2net_m32', 'eres2netv2_m32', 'eres2netv2_w96s4'])
def test_gpu_eres2net_matches_reference_golden(case):
    """ERes2Net / ERes2NetV2: fp32 operands, so far inside the 1e-4 bar."""
    cd, rel = lc.model_case(product_lib(), DEV, case)
    assert cd < 1e-6 and rel < 2e-3, (cd, rel)


def test_gpu_eres2netv2_variable_length_bucketed():
    """BASELINE config 4 shape: ERes2NetV2 on 1-10 s utterances with length bucketing (<= 8 buckets, padded inside a bucket,
    predict_batch semantics); the shortest and the longest bucket are checked in full against the oracle."""
    import mvector.models as M
    from mvector.data_utils.featurizer import AudioFeaturizer
    from mvector.parallel import embed_bucketed, length_buckets
    man, sd, _, _, _ = load_case('eres2netv2_m32')
    m = M.ERes2NetV2(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    fz = AudioFeaturizer('Fbank', method_args=FB)
    g = torch.Generator().manual_seed(77)
    lens = [int(v) for v in torch.randint(16000, 160001, (24,), generator=g)]
    lens[0], lens[1] = 16000, 160000
    wav = frontend.synth_waveforms(len(lens), max(lens), seed=5)
    waves = [wav[i, :n] for i, n in enumerate(lens)]
    emb = embed_bucketed(fz, m, waves, max_buckets=8, device=torch.device(DEV)).cpu()
    assert emb.shape == (24, 192) and torch.isfinite(emb).all() and m.__dict__.get('_native_handles')
    # round 6: the buckets run on two HIP streams in turn (one native handle, one workspace per stream); the same rows on ONE stream carry the same
    # bits, also when the passes are repeated back to back (a workspace shared by two forwards in flight would show here)
    one = embed_bucketed(fz, m, waves, max_buckets=8, device=torch.device(DEV), streams=1).cpu()
    assert torch.equal(emb, one)
    for _ in range(3):
        assert torch.equal(embed_bucketed(fz, m, waves, max_buckets=8, device=torch.device(DEV), streams=2).cpu(), one)
    (h, _, _), = m.__dict__['_native_handles'].values()
    assert len(h._ws) >= 2, 'two streams, two workspaces'
    buckets = length_buckets(lens, 8)
    for idx in (buckets[0], buckets[-1]):
        longest = max(lens[i] for i in idx)
        padded = torch.zeros(len(idx), longest)
        for r, i in enumerate(idx):
            padded[r, :lens[i]] = wav[i, :lens[i]]
        ratio = torch.tensor([lens[i] / longest for i in idx])
        ref = omodels.eres2netv2(sd, frontend.audio_featurizer(padded, ratio, 'Fbank', FB))
        assert cos_dist(emb[idx], ref).max() < 1e-4, cos_dist(emb[idx], ref).max()  # front-end fp32 rounding x the depth of the net


def test_gpu_eres2net_full_batch_properties():
    """ERes2NetV2 m32 at 64 x 3 s through the module API: native path taken, finite, batch-invariant, spot parity vs the oracle."""
    import mvector.models as M
    man, sd, x, emb_ref, _ = load_case('eres2netv2_m32')
    m = M.ERes2NetV2(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(64, 298, 80, generator=g) * 2.0
    feats = feats - feats.mean(1, keepdim=True)
    emb = m(feats.to(DEV))
    assert m.__dict__.get('_native_handles'), 'native handle was not created: forward did not take the HIP path'
    assert emb.shape == (64, 192) and torch.isfinite(emb).all()
    small = m(feats[40:42].to(DEV))
    assert cos_dist(small.cpu(), emb[40:42].cpu()).max() < 1e-9
    ref = omodels.eres2netv2(sd, feats[:1])
    assert cos_dist(emb[:1].cpu(), ref).max() < 1e-6


@pytest.mark.parametrize('case', ['ecapa_tiny', 'tdnn', 'campp_short', 'eres2net_tiny'])
def test_gpu_module_forward_uses_native_and_tracks_weights(case):
    import mvector.models as M
    man, sd, x, emb_ref, _ = load_case(case)
    m = getattr(M, man['model'])(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    emb = m(x.to(DEV))
    assert m.__dict__.get('_native_handles'), 'native handle was not created: forward did not take the HIP path'
    assert cos_dist(emb.cpu(), emb_ref).max() < 1e-4
    # loading different weights must invalidate the derived native copy
    from oracle import weights
    sd2 = weights.make_state_dict(man['shapes'], man['seed'] + 11)
    m.load_state_dict(sd2)
    emb2 = m(x.to(DEV)).cpu()
    ref2 = omodels.FORWARDS[man['model']](sd2, x)
    assert cos_dist(emb2, ref2).max() < 1e-4


def test_gpu_end_to_end_waveform_to_embedding_full_batch():
    """Config 2 shape (EcapaTdnn c=1024, bs=256, 3 s): EVERY row against the oracle (front-end + backbone on the CPU, ~10 s) +
    batch-invariance property."""
    from mvector.data_utils.featurizer import AudioFeaturizer
    from mvector.models import EcapaTdnn
    man, sd, _, _, _ = load_case('ecapa_c1024')
    model = EcapaTdnn(**man['kwargs'])
    model.load_state_dict(sd)
    model.eval().to(DEV)
    fz = AudioFeaturizer('Fbank', method_args=FB)
    wav = frontend.synth_waveforms(256, 48000)
    emb = model(fz(wav.to(DEV)))
    assert emb.shape == (256, 192) and torch.isfinite(emb).all()
    torch.set_num_threads(min(32, os.cpu_count()))
    with torch.no_grad():
        ref = torch.cat([omodels.ecapa_tdnn(sd, frontend.audio_featurizer(wav[i:i + 32], None, 'Fbank', FB)) for i in range(0, 256, 32)])
    assert cos_dist(emb.cpu(), ref).max() < 1e-4
    small = model(fz(wav[100:104].to(DEV)))
    assert cos_dist(small.cpu(), emb[100:104].cpu()).max() < 1e-6  # an utterance's embedding does not depend on its batch


def test_gpu_predictor_matches_cpu_predictor(tmp_path):
    """MVectorPredictor(use_gpu=True) vs (use_gpu=False) on the reference's sample audio (config 1: TDNN + Fbank)."""
    import scipy.io.wavfile as wavfile
    from mvector.predict import MVectorPredictor
    from mvector.models import TDNN
    man, sd, _, _, _ = load_case('tdnn')
    model_dir = tmp_path / 'model'
    model_dir.mkdir()
    torch.save({'0.' + k: v for k, v in sd.items()}, str(model_dir / 'model.pth'))
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    paths = []
    for i, pcm in enumerate(z['pcm16']):
        p = str(tmp_path / f'u{i}.wav')
        wavfile.write(p, 16000, pcm[: 16000 - 1500 * i])
        paths.append(p)
    cfg = dict(dataset_conf=dict(dataset=dict(min_duration=0.3, sample_rate=16000, use_dB_normalization=True, target_dB=-20),
                                 eval_conf=dict(batch_size=2)),
               preprocess_conf=dict(feature_method='Fbank', method_args=dict(sample_frequency=16000, num_mel_bins=80)),
               model_conf=dict(model='TDNN', model_args=dict(embd_dim=192)))
    gpu = MVectorPredictor(cfg, model_path=str(model_dir), use_gpu=True)
    e_gpu = gpu.predict_batch(paths)
    assert gpu._last_batch_path == 'pcm16'  # 16-bit mono WAVs at the target rate: int16 upload, scaled / normalised on the device
    floats = [z['pcm16'][i][: 16000 - 1500 * i].astype(np.float32) / 32768.0 for i in range(4)]
    e_host = gpu.predict_batch(floats)      # ndarray input: the general host path
    assert gpu._last_batch_path == 'host' and cos_dist(e_host, e_gpu).max() < 1e-6
    one = gpu.predict(paths[1])
    c_gpu = gpu.contrast(paths[0], paths[2])
    cpu = MVectorPredictor(cfg, model_path=str(model_dir), use_gpu=False)
    e_cpu = cpu.predict_batch(paths)
    assert e_gpu.shape == (4, 192)
    assert cos_dist(e_gpu, e_cpu).max() < 1e-4
    assert cos_dist(one, cpu.predict(paths[1])).max() < 1e-4
    assert abs(c_gpu - cpu.contrast(paths[0], paths[2])) < 1e-3


def _predictor_fixture(tmp_path, ragged=True):
    import scipy.io.wavfile as wavfile
    man, sd, _, _, _ = load_case('tdnn')
    model_dir = tmp_path / 'model'
    model_dir.mkdir()
    torch.save({'0.' + k: v for k, v in sd.items()}, str(model_dir / 'model.pth'))
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    paths, pcms = [], []
    for i, pcm in enumerate(z['pcm16']):
        p = str(tmp_path / f'u{i}.wav')
        cut = pcm[: 16000 - (1500 * i if ragged else 0)]
        wavfile.write(p, 16000, cut)
        paths.append(p)
        pcms.append(cut)
    cfg = dict(dataset_conf=dict(dataset=dict(min_duration=0.3, sample_rate=16000, use_dB_normalization=True, target_dB=-20),
                                 eval_conf=dict(batch_size=2)),
               preprocess_conf=dict(feature_method='Fbank', method_args=dict(sample_frequency=16000, num_mel_bins=80)),
               model_conf=dict(model='TDNN', model_args=dict(embd_dim=192)))
    return cfg, str(model_dir), paths, pcms, sd


def _oracle_predict_batch(sd, pcms, target_db=-20.0):
    """the reference's predict_batch semantics restated with oracle pieces: int16 -> float, dB normalisation over the true
    length, zero padding to the batch maximum, length ratios, Fbank + CMN over the padded frames + mask, TDNN forward
    (predict.py:185-212, 231-265)"""
    n = torch.tensor([len(p) for p in pcms])
    longest = int(n.max())
    staged = torch.zeros(len(pcms), longest, dtype=torch.int16)
    for i, p in enumerate(pcms):
        staged[i, :len(p)] = torch.from_numpy(np.ascontiguousarray(p))
    wav, _ = frontend.wave_prepare(staged, n, target_db)
    ratio = n.float() / longest
    return omodels.tdnn(sd, frontend.audio_featurizer(wav, ratio, 'Fbank', FB)).numpy()


def test_gpu_predictor_matches_oracle(tmp_path):
    """MVectorPredictor(use_gpu=True) against the ORACLE directly (not against the package's CPU predictor): predict_batch on a
    ragged batch (int16 upload path and host path), predict, contrast."""
    from mvector.predict import MVectorPredictor
    cfg, model_dir, paths, pcms, sd = _predictor_fixture(tmp_path)
    gpu = MVectorPredictor(cfg, model_path=model_dir, use_gpu=True)
    want = _oracle_predict_batch(sd, pcms)
    got = gpu.predict_batch(paths)
    assert gpu._last_batch_path == 'pcm16' and got.shape == (4, 192)
    assert cos_dist(got, want).max() < 1e-4, cos_dist(got, want).max()
    floats = [p.astype(np.float32) / 32768.0 for p in pcms]
    assert cos_dist(gpu.predict_batch(floats), want).max() < 1e-4 and gpu._last_batch_path == 'host'
    single = _oracle_predict_batch(sd, [pcms[1]])[0]
    assert cos_dist(gpu.predict(paths[1]), single).max() < 1e-4
    a, b = _oracle_predict_batch(sd, [pcms[0]])[0], _oracle_predict_batch(sd, [pcms[2]])[0]
    assert abs(gpu.contrast(paths[0], paths[2]) - scoring.contrast(a, b)) < 1e-3


def test_gpu_register_recognition_remove_user_device_gallery(tmp_path):
    """f4: enrolment matrix resident on the GPU (per-user sums, updated in place), recognition through mv_cosine_f32 without
    re-uploading it; register -> recognition -> remove_user against the oracle's retrieval (predict.py:169-183, 281-363)."""
    from mvector.predict import MVectorPredictor
    cfg, model_dir, paths, pcms, sd = _predictor_fixture(tmp_path, ragged=False)
    db = tmp_path / 'audio_db'
    p = MVectorPredictor(cfg, threshold=0.3, audio_db_path=str(db), model_path=model_dir, use_gpu=True)
    dg = p._device_gallery
    assert dg is not None and dg.matrix().is_cuda and dg.matrix().shape == (0, 192)
    emb = [_oracle_predict_batch(sd, [pcm])[0] for pcm in pcms]
    assert p.register(paths[0], 'alice') == (True, '注册成功') and p.register(paths[2], 'bob')[0] and p.register(paths[1], 'alice')[0]
    assert dg.users == ['alice', 'bob'] and dg.uploads == 0   # embeddings went from the backbone into the matrix on the device
    means = np.stack([(emb[0] + emb[1]) / 2, emb[2]])
    for q in (1, 3):
        want = scoring.retrieval(emb[q][None], means, ['alice', 'bob'], 0.3)[0]
        got = p.recognition(paths[q])
        assert got[0] == want[0] and abs(got[1] - want[1]) < 2e-3, (got, want)
    assert dg.uploads == 0                                     # recognition re-uploaded nothing
    sums = dg.matrix().cpu().numpy()
    assert cos_dist(sums, means).max() < 1e-4                  # sums and means point the same way
    assert p.remove_user('alice') and dg.users == ['bob'] and p.get_users() == ['bob']
    got = p.recognition(paths[0])
    want = scoring.retrieval(emb[0][None], means[1:], ['bob'], 0.3)[0]
    assert got[0] == want[0] and (got[1] is None or abs(got[1] - want[1]) < 2e-3)
    # a fresh predictor rebuilds the resident matrix from audio_indexes.bin with ONE upload
    p2 = MVectorPredictor(cfg, threshold=0.3, audio_db_path=str(db), model_path=model_dir, use_gpu=True)
    assert p2._device_gallery.users == ['bob'] and p2._device_gallery.uploads == 1
    assert p2.recognition(paths[2])[0] == 'bob'


def test_gpu_cosine_more_than_a_million_rows():
    """evaluation score matrices beyond 65535 row tiles: rows go in chunks inside mv_cosine_f32"""
    from mvector import _hip
    g = torch.Generator().manual_seed(5)
    a = torch.randn(65535 * 16 + 40, 8, generator=g).to(DEV)
    b = torch.randn(3, 8, generator=g).to(DEV)
    s = _hip.cosine(a, b)
    ref = torch.nn.functional.normalize(a, dim=1) @ torch.nn.functional.normalize(b, dim=1).t()
    assert s.shape == (65535 * 16 + 40, 3) and (s - ref).abs().max().item() < 2e-6


def test_gpu_melspec_golden_and_variants():
    z = np.load(os.path.join(GOLDEN, 'frontend.npz'))
    wav = frontend.synth_waveforms(4, 48000)
    from mvector import _hip
    ms = _hip.MelSpec({})
    out = ms(wav[:2].to(DEV)).cpu().numpy()
    assert out.shape == (2, 241, 128)
    assert np.abs(out - z['mel']).max() <= 2e-4 * np.abs(z['mel']).max()
    lens = z['lens']
    wav_var = torch.zeros(4, 48000)
    for i, n in enumerate(lens):
        wav_var[i, :n] = wav[i, :n] * (1e-4 if i == 3 else 1.0)
    outv = ms(wav_var.to(DEV), torch.from_numpy(z['ratio']).to(DEV)).cpu().numpy()[2:]
    assert np.abs(outv - z['mel_var']).max() <= 2e-4 * np.abs(z['mel_var']).max()
    readme = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000, n_mels=64)
    from mvector import _hip
    assert _hip.MelSpec(readme).info()['kernel'] == 'melspec_pow2_kernel'   # the README's run takes the FFT kernel (round 3), not the dense DFT
    lc.melspec_case(product_lib(), DEV, wav[:1], None, readme)
    lc.melspec_case(product_lib(), DEV, frontend.synth_waveforms(5, 48000, seed=8), torch.tensor([1.0, 0.4, 0.77, 0.5, 0.9]), readme)
    for n_fft, extra in ((512, dict(hop_length=128)), (256, dict(win_length=200, hop_length=80, n_mels=40, f_min=50, f_max=7000)),
                         (128, dict(hop_length=64, n_mels=32)), (512, dict(center=False, n_mels=80))):
        args = dict(n_fft=n_fft, **extra)
        assert _hip.MelSpec(args).info()['kernel'] == 'melspec_pow2_kernel'
        lc.melspec_case(product_lib(), DEV, frontend.synth_waveforms(3, 20000 + 13, seed=n_fft), torch.tensor([1.0, 0.35, 0.8]), args)
    lc.melspec_case(product_lib(), DEV, frontend.synth_waveforms(2, 160000, seed=9), None, readme)   # 10 s: 501 frames, beyond the LDS tile
    lc.melspec_case(product_lib(), DEV, frontend.synth_waveforms(3, 16000 + 37, seed=5), torch.tensor([1.0, 0.3, 0.81]), {})


def test_gpu_ecapa_on_melspectrogram_end_to_end():
    """Config 3 shape: EcapaTdnn(128) on MelSpectrogram defaults, waveform -> embedding vs the oracle."""
    from mvector.data_utils.featurizer import AudioFeaturizer
    from mvector.models import EcapaTdnn
    man, sd, _, _, _ = load_case('ecapa_mel128')
    model = EcapaTdnn(**man['kwargs'])
    model.load_state_dict(sd)
    model.eval().to(DEV)
    fz = AudioFeaturizer('MelSpectrogram', method_args={})
    wav = frontend.synth_waveforms(8, 48000, seed=77)
    emb = model(fz(wav.to(DEV)))
    ref = omodels.ecapa_tdnn(sd, frontend.audio_featurizer(wav[:3], None, 'MelSpectrogram', {}))
    assert cos_dist(emb[:3].cpu(), ref).max() < 1e-4


@pytest.mark.parametrize('cfg', [dict(), dict(online=True), dict(B=5, T=9, C=72, A=64, ldx=80, centred=False),
                                 dict(B=2, T=33, C=64, A=128, online=True, wscale=1.0), dict(B=9, T=298, C=3072, A=128),
                                 dict(B=6, T=298, C=1536, A=128, online=True), dict(B=3, T=1, C=64, A=128),
                                 dict(B=2, T=45, C=256), dict(B=2, T=100, C=512, A=64), dict(B=1, T=16, C=256), dict(B=1, T=1, C=256),
                                 dict(B=2, T=17, C=256, centred=False), dict(B=3, T=1000, C=1024, ldx=1032), dict(B=300, T=50, C=256)])
def test_gpu_asp_pool(cfg):
    lc.asp_pool_case(product_lib(), DEV, **cfg)


FCM_GPU_CASES = lc.FCM_CASES + [
    dict(B=9, Fin=80, T=298, sf=2, mode2=0),            # the shipped geometry: FCM layer1.block0.conv1, NI = 5
    dict(B=7, Fin=40, T=298, sf=1, mode2=1, sf2=2),     # layer1.block0.conv2 + shortcut
    dict(B=7, Fin=40, T=298, sf=1, mode2=2),            # layer1.block1.conv2 + identity
    dict(B=300, Fin=20, T=298, sf=2, mode2=0, strided_out=True),  # more workgroups than CUs, FCM.conv2 layout
    dict(B=2, Fin=20, T=1000, sf=1, mode2=2),           # 10 s: four time tiles
]


@pytest.mark.parametrize('idx', range(len(FCM_GPU_CASES)))
@pytest.mark.parametrize('impl', ['band', 'row'])
def test_gpu_fcm_conv3x3(idx, impl):
    """the band kernel (16-byte aligned output rows) and the one-row-per-workgroup kernel that takes every other output layout (here: rows
    72 bytes apart)"""
    cfg = dict(FCM_GPU_CASES[idx])
    if impl == 'row':
        if cfg.pop('strided_out', False):
            pytest.skip('the padded layout of the row-kernel arm replaces the [B, T, F, 32] layout of this case')
        cfg['padded_out'] = True
    lc.fcm_conv_case(product_lib(), DEV, seed=idx, **cfg)


FCM_BLOCK_GPU_CASES = lc.FCM_BLOCK_CASES + [
    dict(B=9, Fin=80, T=298, sf=2),                      # the shipped geometry: layer1.block0, NT = 5, bands (9 < CUs)
    dict(B=7, Fin=40, T=298, sf=1),                      # layer1.block1 (identity), ring 3 steps ahead
    dict(B=300, Fin=40, T=298, sf=2),                    # more workgroups than CUs, one band each: layer2.block0
    dict(B=2, Fin=20, T=1000, sf=1),                     # 10 s: four time tiles
    dict(B=260, Fin=20, T=150, sf=1, strided_out=True),  # NT = 3: the deepest ring
    dict(B=3, Fin=10, T=200, sf=2),                      # NT = 4
]


@pytest.mark.parametrize('idx', range(len(FCM_BLOCK_GPU_CASES)))
def test_gpu_fcm_block(idx):
    """BasicResBlock in one launch (fcmblock.hip) against the fp64 reference with an fp16-rounded intermediate map"""
    lc.fcm_block_case(product_lib(), DEV, seed=idx, **FCM_BLOCK_GPU_CASES[idx])


@pytest.mark.parametrize('idx', range(len(lc.FCM_BLOCK_C1_CASES) + 2))
def test_gpu_fcm_block_with_first_conv(idx):
    """head.conv1 evaluated inside the first block's kernel: the emulator's cases + the product shape (80 bins x 298 frames) and a ragged one"""
    cases = lc.FCM_BLOCK_C1_CASES + [dict(B=5, F=80, T=298), dict(B=2, F=41, T=621)]
    lc.fcm_block_c1_case(product_lib(), DEV, seed=60 + idx, **cases[idx])


@pytest.mark.parametrize('cfg', [dict(width=64, T=45, dil=3), dict(width=128, T=298, dil=4, B=5), dict(width=64, T=298, dil=2, B=3),
                                 dict(width=128, T=320, dil=3, B=2), dict(width=128, T=17, dil=2, B=2),
                                 dict(width=128, T=321, dil=4, B=3), dict(width=128, T=998, dil=2, B=2), dict(width=64, T=600, dil=3, B=2), dict(width=128, T=600, dil=3, B=128)])   # > 320 frames: chunks with halo rows
def test_gpu_res2net_fused_chain(cfg):
    lc.res2_chain_case(product_lib(), DEV, **cfg)


@pytest.mark.parametrize('cfg', [dict(B=2, T=45, width=64, dil=3), dict(B=1, T=100, width=128, dil=2), dict(B=3, T=170, width=128, dil=4), dict(B=300, T=298, width=128, dil=3)])
def test_gpu_res2net_chain_saturates_at_the_fp16_range(cfg):
    """BatchNorm scales of 3e4: step outputs and next-input sums leave the fp16 range; the chain saturates at +-65504 through MODE.FP16_OVFL
    (tools/fp16_ovfl_probe.hip) on every launch form -- ring, direct, 5-tile chunks, a batch that fills the chip"""
    lc.res2_chain_case(product_lib(), DEV, seed=11, gain=3.0e4, **cfg)


@pytest.mark.parametrize('cfg', [dict(T=298, dil=2), dict(T=298, dil=3), dict(T=298, dil=4), dict(T=150, dil=3), dict(T=500, dil=2), dict(T=81, dil=4)])
def test_gpu_res2net_chain_small_batch_form_carries_the_same_bits(cfg):
    """small batches run the chain as <= 160-frame chunks on the 5-tile form of the kernel (one 3 s utterance: three workgroups instead of one); the
    same utterances in a batch that fills the chip take one workgroup each: identical bits, checked row by row"""
    lc.res2_chain_case(product_lib(), DEV, width=128, B=200, alone_rows=2, seed=7, **cfg)


@pytest.mark.parametrize('case', ['campp_short', 'ecapa_tiny', 'tdnn'])
def test_gpu_model_forward_is_hipgraph_capturable(case):
    """a model forward is a fixed launch sequence on the caller's stream over the caller's workspace (no allocation, no
    synchronisation) -- so it can be captured in a hipGraph and replayed on new inputs (CAM++ issues ~110 launches per forward: at
    batch 1 the eager path is bound by the host's launch rate).  Capture after one warm-up (lazy one-time setup), replay on two inputs."""
    from mvector import models as pmodels
    man, sd, x, _, _ = load_case(case)
    model = getattr(pmodels, man['model'])(**man['kwargs'])
    model.load_state_dict(sd)
    model.eval().to(DEV)
    xs = [x.to(DEV), (x * 0.5 + 0.1).to(DEV)]
    with torch.no_grad():
        eager = [model(v).clone() for v in xs]
        static_in = xs[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(static_in)  # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_out = model(static_in)
        for v, ref in zip(xs, eager):
            static_in.copy_(v)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_out, ref), (static_out - ref).abs().max().item()


@pytest.mark.parametrize('case,T', [('ecapa_c1024', 305), ('ecapa_c1024', 321), ('ecapa_c1024', 998), ('ecapa_c512', 998), ('ecapa_c512', 9),
                                    ('campp', 998), ('campp', 23), ('campp', 322), ('campp', 640), ('campp', 803), ('tdnn', 998), ('tdnn', 30),
                                    ('campp', 3001), ('ecapa_c512', 3001), ('ecapa_c1024', 1501)])   # 30 s / 15 s: many chunks per utterance
def test_gpu_backbones_long_and_short_utterances(case, T):
    """Frame counts outside the golden fixtures (1-10 s is the range BASELINE configs[4] names): T = 305 / 321 leave the Res2Net
    chain's direct form / fused form (its limits are 304 / 320 frames), 998 frames = 10 s takes the multi-tile paths of the ASP
    pooling and the CAM dense layers (T/2 > 160 strided frames: two launches per layer over 160-frame chunks -- 322 -> one full chunk + one
    frame, 640 -> exactly two chunks, 803 -> three chunks with the segment boundaries inside them), the short ones are close to the
    reflect-padding minimum.  Reference = the oracle on the CPU."""
    from mvector import models as pmodels
    man, sd, x, _, _ = load_case(case)
    g = torch.Generator().manual_seed(T)
    feats = torch.randn(2, T, x.shape[2], generator=g) * x.std() + x.mean()
    ref = omodels.FORWARDS[man['model']](sd, feats)
    model = getattr(pmodels, man['model'])(**man['kwargs'])
    model.load_state_dict(sd)
    model.eval().to(DEV)
    with torch.no_grad():
        emb = model(feats.to(DEV)).cpu()
    d = cos_dist(emb, ref).max().item()
    assert d < 1e-4, d


def test_gpu_campp_long_utterance_two_launch_dense_layers():
    """CAM++ beyond 3.2 s: the two-launch dense layers over even chunks (camdense.hip) against the oracle, in a batch and alone (the chunk
    count follows the batch size; the embeddings agree to rounding)"""
    from mvector import models as pmodels
    man, sd, x, _, _ = load_case('campp')
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(3, 700, x.shape[2], generator=g) * x.std() + x.mean()
    ref = omodels.FORWARDS[man['model']](sd, feats)
    model = getattr(pmodels, man['model'])(**man['kwargs'])
    model.load_state_dict(sd)
    model.eval().to(DEV)
    with torch.no_grad():
        out = model(feats.to(DEV)).cpu()
        one = model(feats[:1].to(DEV)).cpu()
    assert cos_dist(out, ref).max().item() < 1e-4
    assert cos_dist(one, out[:1]).max().item() < 1e-6


def test_gpu_rccl_one_rank_group_runs_the_exchange_step():
    """The multi-GPU exchange is one all_gather_into_tensor on the nccl (= RCCL) backend.  The GPU boxes of
    this build have one device, so this runs the very call in a one-rank group: it proves that the RCCL stack of the image loads,
    that the process-group bootstrap on 127.0.0.1 works and that the sharded step (embed -> all-gather -> cosine block) runs on
    device buffers.  The N > 1 layout itself is covered by the world_size-2 gloo tests."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')\n"
        "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1)\n"
        "from helpers import load_case, cos_dist\n"
        "from oracle import frontend, models as omodels, scoring\n"
        "from mvector import parallel\n"
        "from mvector.data_utils.featurizer import AudioFeaturizer\n"
        "from mvector.models import EcapaTdnn\n"
        "man, sd, _, _, _ = load_case('ecapa_tiny')\n"
        "m = EcapaTdnn(**man['kwargs']); m.load_state_dict(sd); m.eval().cuda()\n"
        "FB = dict(sample_frequency=16000, num_mel_bins=80)\n"
        "wav = frontend.synth_waveforms(6, 8000, seed=21)\n"
        "lo, hi = parallel.shard_rows(6)\n"
        "assert (lo, hi) == (0, 6)\n"
        "with torch.no_grad():\n"
        "    emb = m(AudioFeaturizer('Fbank', method_args=FB)(wav.cuda()))\n"
        "    allemb = parallel.all_gather_embeddings(emb, always=True)\n"
        "    scores = parallel.cosine_block(emb, allemb)\n"
        "torch.cuda.synchronize()\n"
        "assert allemb.data_ptr() != emb.data_ptr() and torch.equal(allemb, emb)\n"
        "ref = omodels.ecapa_tdnn(sd, frontend.audio_featurizer(wav, None, 'Fbank', FB))\n"
        "assert cos_dist(allemb.cpu(), ref).max().item() < 1e-4\n"
        "sim = scoring.cosine_similarity(ref.numpy(), ref.numpy())\n"
        "assert abs(scores.cpu().numpy() - sim).max() < 1e-3\n"
        "dist.destroy_process_group()\n"
        "print('rccl one-rank exchange ok')\n"
    ) % (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gpu_embed_stream_pipeline_matches_batch_by_batch():
    """parallel.embed_stream: uploads on a copy stream, int16 PCM converted on the device, downloads behind the next batch's compute;
    seven batches (two shapes, a ragged last one) must come back in order and equal to the sequential path, with and without the
    reference's dB normalisation."""
    from mvector import _hip, parallel
    from mvector.models import EcapaTdnn
    from mvector.data_utils.featurizer import AudioFeaturizer
    man, sd, _, _, _ = load_case('ecapa_tiny')
    m = EcapaTdnn(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    fz = AudioFeaturizer('Fbank', method_args=FB)
    wav = frontend.synth_waveforms(50, 12000, seed=5) * torch.linspace(0.2, 1.0, 50)[:, None]
    pcm = (wav * 32768).round().clamp(-32768, 32767).to(torch.int16)
    batches = [pcm[0:8].pin_memory(), pcm[8:16].pin_memory(), pcm[16:24, :9000].contiguous().pin_memory(), pcm[24:32].pin_memory(),
               pcm[32:40, :9000].contiguous().pin_memory(), pcm[40:48].pin_memory(), pcm[48:50].pin_memory()]
    for target_db in (None, -20.0):
        got = [o.clone() for o in parallel.embed_stream(fz, m, batches, device=DEV, target_db=target_db)]
        with torch.no_grad():
            for o, b in zip(got, batches):
                w, _ = _hip.wave_prepare(b.to(DEV), target_db=target_db)
                assert torch.equal(o, m(fz(w)).cpu())
    # a stream in which every batch has its own length (a length-sorted evaluation list) must not allocate one ring of
    # device buffers per shape: the slots only grow to the largest batch
    torch.cuda.synchronize()
    ragged = [pcm[0:8, :L].contiguous().pin_memory() for L in range(4000, 12000, 500)]
    base = torch.cuda.memory_allocated()
    peak = 0
    got = []
    for o in parallel.embed_stream(fz, m, ragged, device=DEV):
        got.append(o.clone())
        peak = max(peak, torch.cuda.memory_allocated() - base)
    assert peak < 6 * ragged[-1].numel() * 4 + (64 << 20), peak   # two int16 slots + one batch of float32 intermediates, not 16 rings
    with torch.no_grad():
        for o, b in zip(got, ragged):
            w, _ = _hip.wave_prepare(b.to(DEV))
            assert torch.equal(o, m(fz(w)).cpu())
    fl = [b.float().div(32768.0).pin_memory() for b in batches[:3]]  # float32 waveforms take the same pipeline
    got = [o.clone() for o in parallel.embed_stream(fz, m, fl, device=DEV)]
    with torch.no_grad():
        for o, b in zip(got, fl):
            assert torch.equal(o, m(fz(b.to(DEV))).cpu())


@pytest.mark.parametrize('name', ['ecapa1024', 'campp'])
def test_gpu_repeated_full_batches_are_bit_identical(name):
    """Race detector for the hand-synchronised kernels (counted s_waitcnt rings, the barrier-free K loop of the Res2Net chain, cross-wave
    LDS hand-overs): the path has no atomics, so the same 256-utterance batch must give bit-identical features and embeddings on every run,
    also with a different batch going through the same workspace in between (tools/stress_determinism.py runs the long version)."""
    sys.path.insert(0, ROOT)
    import bench
    featurizer, model, _ = bench.build(name, torch.device(DEV))
    g = torch.Generator().manual_seed(99)
    wav = (0.1 * torch.randn([256, bench.SAMPLES], generator=g)).clamp(-1, 1).to(DEV)
    wav2 = wav.flip(0).contiguous() * 0.5
    with torch.no_grad():
        f0 = featurizer(wav).clone()
        e0 = model(f0).clone()
        for _ in range(12):
            model(featurizer(wav2))
            f = featurizer(wav)
            assert torch.equal(f, f0)
            assert torch.equal(model(f), e0)


@pytest.mark.parametrize('B', [250, 300])
def test_gpu_ecapa1024_batches_with_ragged_last_tiles(B):
    """B * 298 rows is not a multiple of the 256-row GEMM tile for these batch sizes (and 300 utterances make more tiles than resident
    workgroups can take in equal shares): the persistent kernels' clamped tail rows and uneven walks inside the real model, checked on the
    first, a middle and the last utterance against the oracle."""
    from mvector.models import EcapaTdnn
    from mvector.data_utils.featurizer import AudioFeaturizer
    man, sd, _, _, _ = load_case('ecapa_c1024')
    m = EcapaTdnn(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    wav = frontend.synth_waveforms(B, 48000, seed=B)
    with torch.no_grad():
        emb = m(AudioFeaturizer('Fbank', method_args=FB)(wav.to(DEV))).cpu()
    rows = [0, B // 2, B - 1]
    ref = omodels.ecapa_tdnn(sd, frontend.audio_featurizer(wav[rows], None, 'Fbank', FB))
    d = cos_dist(emb[rows], ref).max().item()
    assert d < 1e-4, d


S16_RANGE_CASES = [
    dict(cin=32, cout=32, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, peak=True),                                   # peak below the range: reported exactly
    dict(cin=32, cout=32, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, peak=True, x_scale=200.0, seed=2),            # outputs beyond 1023.5: clamped AND reported
    dict(cin=32, cout=48, ks=1, H=6, W=20, B=2, lo=-3.0e38, hi=3.0e38, peak=True, x_scale=200.0, with_res=True, seed=3),
    dict(cin=16, cout=16, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, nan_at=(0, 2, 5, 3)),                          # a NaN input reaches its 9 x 16 outputs as NaN
    dict(cin=32, cout=32, ks=1, H=6, W=20, B=1, epi=1, nan_at=(0, 2, 5, 3), peak=True),
    dict(cin=32, cout=16, ks=1, H=4, W=20, B=1, epi=2, nan_at=(0, 1, 7, 30), peak=True),
]


@pytest.mark.parametrize('idx', range(len(S16_RANGE_CASES)))
def test_gpu_conv2ds_reports_its_peak_and_keeps_nans(idx):
    """S16 maps saturate at |value| = 1023.5 and the clamps turned NaNs into finite bounds.  MvConv2dsDesc.peak reports the largest
    value a launch wanted to store (the CAM++ handle picks its exact head's gain from it and exposes saturation on real inputs); NaNs travel on."""
    lc.conv2ds_case(product_lib(), DEV, **S16_RANGE_CASES[idx])


def test_gpu_campp_hot_head_golden_runs_the_exact_head_inside_its_range():
    """tests/golden/campp_hot (oracle/make_golden.py::save_hot_golden): the ill-conditioned campp_stress checkpoint with every inner FCM-head map
    2^11 times larger -- 2e3 .. 4e4, beyond the +-1023.5 S16 maps hold at their scale of 64.  The handle measures the probes' largest map value at
    create and runs the exact head at a gain of 2^k (exact: the head is positively homogeneous) that puts it 16 x below the bound; the golden is
    met at the north-star bar, nothing saturates on the test input, and the choice is visible (mv_model_info / Model.campp_head())."""
    info = {1: None, 2: None, 6: None, 7: None, 8: None, 9: None}
    cd, rel = lc.model_case(product_lib(), DEV, 'campp_hot', tol=1e-4, info=info)
    print(f'campp_hot: head fp32 = {info[1]}, calibration {info[2]:.3e}, gain 2^{info[6]:.0f}, probe peak {info[7]:.4g}, peak on the test input {info[8]:.4g}, '
          f'saturated {info[9]}; 1 - cos = {cd:.3e}, max rel err {rel:.3e}')
    assert info[1] == 1.0 and info[6] <= -5 and info[7] > 2e3 and info[8] > 2e3 and info[9] == 0.0
    assert info[7] * 64.0 * 2.0 ** info[6] <= 65504.0 / 16.0 * 1.0001
    info0 = {6: None, 7: None}
    lc.model_case(product_lib(), DEV, 'campp_stress', tol=1e-4, info=info0, head=2)
    assert info0[6] == 0.0 and info0[7] < 1023.5 / 16.0 * 1.0001, info0   # an ordinary checkpoint keeps the scale of 64


def test_gpu_campp_xvector_sensitivity_report_fires():
    """VERDICT r5 item 4b.  tests/golden/campp_c64_s31 (reference modules, oracle/make_golden.py campp_variants): CAMPPlus(init_channels=64) on the weight
    seed whose embedding moves by 3.3e-4 under the fp16 rounding of the x-vector WEIGHTS alone (tests/budget_campp.py) -- a miss the head probes cannot
    see (they compare the two FCM heads) and no head choice repairs.  The handle now evaluates the x-vector part of its three probes once more in exact
    fp32 at create and reports 1 - cos against the shipped path (mv_model_info 10-13): it must exceed MV_CAMPP_XVEC_WARN on this checkpoint, stay an
    order of magnitude below it on the ordinary goldens, and the Python surface must say so (Model.campp_head(), a RuntimeWarning from the module)."""
    import warnings
    from mvector import _hip
    WARN = _hip.Model.XVEC_WARN
    info = {10: None, 11: None, 12: None, 13: None}
    cd, _ = lc.model_case(product_lib(), DEV, 'campp_c64_s31', tol=2e-3, info=info)
    print(f'campp_c64_s31: golden 1 - cos {cd:.3e} (the 1e-4 contract does NOT hold on this checkpoint), x-vector sensitivity {info[10]:.3e}, probes '
          f'{info[11]:.2e} / {info[12]:.2e} / {info[13]:.2e}')
    assert cd > 1e-4, 'the fixture is meant to miss the bar'
    assert info[10] > WARN and info[10] == max(info[11], info[12], info[13])
    for case in ('campp', 'campp_c64', 'campp_short'):
        ok = {10: None}
        cd, _ = lc.model_case(product_lib(), DEV, case, tol=1e-4, info=ok)
        print(f'{case}: golden 1 - cos {cd:.3e}, x-vector sensitivity {ok[10]:.3e}')
        assert 0.0 <= ok[10] < WARN / 2, (case, ok)
    off = {10: None}
    lc.model_case(product_lib(), DEV, 'campp_short', tol=1e-4, info=off, xvec_probe=False)
    assert off[10] == -1.0   # MvCamppCfg.xvector_probe = OFF: not measured
    # the module: one RuntimeWarning per handle build, figures through native_head()
    from helpers import load_case
    from mvector.models import CAMPPlus
    man, sd, x, emb_ref, _ = load_case('campp_c64_s31')
    m = CAMPPlus(**man['kwargs'])
    m.load_state_dict(sd)
    m.eval().to(DEV)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        with torch.no_grad():
            m(x.to(DEV))
    assert any(issubclass(w.category, RuntimeWarning) and 'x-vector part' in str(w.message) for w in caught), [str(w.message) for w in caught]
    rep = m.native_head()
    assert rep['xvector_warning'] and rep['xvector_sensitivity'] > WARN and len(rep['xvector_probes']) == 3


@pytest.mark.parametrize('idx', range(len(lc.MELSPEC_ARG_CASES)))
def test_gpu_melspec_arguments(idx):
    """HIP MelSpectrogram vs the oracle over the keyword arguments beyond the shipped configurations (tests/layer_checks.py::MELSPEC_ARG_CASES):
    Slaney mel scale / norm (the oracle's filterbanks agree with transformers' independent ones to 2e-7), normalized = True / "window" /
    "frame_length", window_fn + wkwargs, power 1 / 1.5 / 3, centre off, 8 / 22.05 kHz -- on 5 x 3 s with a ragged mask and 260 x 0.5 s"""
    lc.melspec_arguments_case(product_lib(), DEV, idx, B=5, seconds=3.0)
    lc.melspec_arguments_case(product_lib(), DEV, idx, B=260, seconds=0.5)


def test_gpu_bench_under_a_launcher_runs_its_collectives_on_the_rccl_group():
    """bench.py under torch.distributed.run with ONE rank: gloo rendezvous group + device-count exchange, `new_group(backend='nccl')`, and the step's
    all-gather / barriers / max-reduce on that group -- the N > 1 code path of the driver's SCALE runs on a 1-GPU box (the 8-GPU curve itself has
    never been measured)."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='16')   # (the launcher's default of one OpenMP thread would slow the 4-row oracle check)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-other-configs'],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['ranks'] == {'seen_at_rendezvous': 1, 'rccl_world_size': 1, 'launcher': 'torch.distributed.run'}
    assert d['value'] > 10000 and d['parity']['max_one_minus_cos'] < 1e-4
    # round 6: the box yard-stick travels with every line -- streaming copy, bare MFMA rate and clock, and the clock INSIDE a ring-GEMM launch
    # (MvConv1dDesc.clock_probe): sane ranges for an MI355X, and the in-kernel clock at or below the bare-MFMA one
    box = d['box']
    assert 'error' not in box and 'ring_k3072_error' not in box, box
    assert 1500 < box['copy_gbs'] < 8000 and 800 < box['mfma_f16_tflops'] < 2600, box
    assert 0.8 < box['ring_k3072_clock_ghz'] <= box['short_launch_clock_ghz'] + 0.05 and 0.8 < box['mfma_clock_ghz'] < 2.6, box
    assert 0.2 < box['ring_k3072_frac_of_2p5pf'] < box['ring_k3072_frac_at_sustained_clock'] < 1.0, box
