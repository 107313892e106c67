"""CPU tests that execute the HIP kernel SOURCES under the SIMT emulator (tests/emu/hip_emu.h): indexing,
MFMA fragment layouts, LDS addressing and the native model orchestration are checked against the oracle / golden
vectors without a GPU.  The emulator library is test infrastructure; the product never loads it."""
import os

import numpy as np
import pytest
import torch

import layer_checks as lc
from emu_lib import emu_cdll
from mvector import _hip
from helpers import GOLDEN
from oracle import frontend

FB = dict(sample_frequency=16000, num_mel_bins=80)


@pytest.mark.parametrize('idx', range(len(lc.CONV2DS_CASES)))
def test_conv2ds_emu(idx):
    """split-fp16 conv2d on S16 maps (conv2ds.hip): producer / consumer roles, patch geometry, swizzle, K tails, epilogues"""
    if idx in (9,) and os.environ.get('MV_SLOW_EMU') != '1':
        pytest.skip('slow under the emulator; set MV_SLOW_EMU=1 (covered on the GPU by test_gpu_parity)')
    lc.conv2ds_case(emu_cdll(), 'cpu', seed=idx, **lc.CONV2DS_CASES[idx])


def test_tstp_and_first_conv_emu():
    lc.tstp_case(emu_cdll(), 'cpu')
    lc.conv2d_first_case(emu_cdll(), 'cpu')
    lc.tstp_case(emu_cdll(), 'cpu', s16=True)
    lc.conv2d_first_case(emu_cdll(), 'cpu', s16=True)


@pytest.mark.parametrize('cfg', [dict(B=2, T=70, cout=64), dict(B=1, T=150, cout=256, tile=256)])
def test_conv1d_window_emu(cfg):
    lc.conv1d_window_case(emu_cdll(), 'cpu', **cfg)


@pytest.mark.parametrize('idx', range(len(lc.CONV_CASES)))
def test_emu_conv1d(idx):
    lc.conv1d_case(emu_cdll(), 'cpu', seed=idx, **lc.CONV_CASES[idx])


@pytest.mark.parametrize('cfg', [dict(k=1, dil=1, cin=128, cout=256, T=150, B=2, tile=256),                  # ring kernel: MODE.FP16_OVFL
                                 dict(k=1, dil=1, cin=128, cout=256, T=150, B=2, tile=256, post_act=1),       # post-activation: the double-buffer kernel (v_med3)
                                 dict(k=3, dil=2, cin=64, cout=128, T=70, B=2), dict(k=1, dil=1, cin=64, cout=64, T=33, B=1)])   # 128 / 64 tiles
def test_emu_conv1d_saturates_at_the_fp16_range(cfg):
    lc.conv1d_case(emu_cdll(), 'cpu', seed=5, out_gain=1.0e5, **cfg)


def test_emu_profile_classes_ring_is_a_subset_of_conv1d():
    lc.profile_classes_case(emu_cdll(), 'cpu')


@pytest.mark.parametrize('cfg,blocks,launches', [
    (dict(k=1, dil=1, cin=512, cout=512, T=250, B=5, tile=256), 8, 2),     # 5 x 2 = 10 tiles: one round of 8 + 2 tiles (the ragged last rows) as 8 quarters
    (dict(k=1, dil=1, cin=512, cout=256, T=290, B=15, tile=256), 16, 2),   # 17 tiles: one round of 16 + the tile of the ragged last rows as 16 sixteenths
    (dict(k=1, dil=1, cin=256, cout=512, T=250, B=5, tile=256), 8, 1),     # four K stages: below the tail's threshold
    (dict(k=1, dil=1, cin=512, cout=512, T=250, B=6, tile=256), 8, 1),     # 4 tiles in the partial round = 16 quarters > 8 workgroups: not split
])
def test_emu_ring_tail_sub_tiles_carry_the_same_bits(cfg, blocks, launches):
    n3, n0, w3, w0 = lc.ring_tail_case(emu_cdll(), 'cpu', blocks=blocks, **cfg)
    assert (n3, n0) == (1, launches) and (w3 < w0) == (launches == 2)


def test_emu_conv1d_input_statistics_two_launch_forms_agree():
    """fused kernel (3 utterances on the emulator's 8-CU chip) vs stand-alone statistics + small-tile conv (1 utterance): identical bits"""
    lc.in_stats_forms_case(emu_cdll(), 'cpu', B_big=3, T=298, cin=192)
    lc.in_stats_forms_case(emu_cdll(), 'cpu', B_big=5, T=161, cin=136, seed=1)


def test_emu_conv1d_rejects_bad_arguments():
    with pytest.raises(RuntimeError, match='reflect padding'):
        lc.conv1d_case(emu_cdll(), 'cpu', T=3, k=3, dil=4)


@pytest.mark.parametrize('shape', [(5, 100, 37, 1), (17, 64, 16, 0), (1, 7, 3, 3), (33, 1024, 128, 2), (3, 2100, 20, 0), (4, 4180, 18, 0), (150, 4100, 20, 0),
                                   (40, 6144, 192, 0), (33, 2500, 17, 2), (70, 2050, 40, 3)])   # (B >= 32 and K >= 2048: also the split-K workspace form)
def test_emu_linear(shape):
    B, K, O, act = shape
    lc.linear_case(emu_cdll(), 'cpu', B, K, O, act)


def test_emu_cosine_matches_sklearn_golden():
    z = np.load(os.path.join(GOLDEN, 'cosine.npz'))
    lc.cosine_case(emu_cdll(), 'cpu', z['a'], z['b'], z['sim'])


@pytest.mark.parametrize('unbiased,eps', [(0, 1e-12), (1, 0.0)])
def test_emu_time_stats(unbiased, eps):
    lc.time_stats_case(emu_cdll(), 'cpu', unbiased=unbiased, eps=eps)


def test_emu_bn_relu_rows():
    lc.bn_relu_rows_case(emu_cdll(), 'cpu')
    lc.bn_relu_rows_case(emu_cdll(), 'cpu', rows=5, C=1024, ldx=1024, ldy=1024, seed=1)


def test_emu_fbank_fixed_and_ragged():
    wav = frontend.synth_waveforms(3, 16000 + 37, seed=3)  # odd stride -> scalar-load path
    wav[2, 9000:] = 0
    ratio = torch.tensor([1.0, 0.61, 9000 / 16037])
    lc.fbank_case(emu_cdll(), 'cpu', wav, ratio, FB)
    wav2 = frontend.synth_waveforms(2, 8000, seed=4)        # aligned float2 path, no mask
    lc.fbank_case(emu_cdll(), 'cpu', wav2, None, FB)


def test_emu_fbank_real_audio_golden():
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    wav = torch.from_numpy(z['pcm16'][:2].astype(np.float32) / 32768.0)
    fb = lc._hip.Fbank(FB, cdll=emu_cdll())
    out = fb(wav)
    assert np.abs(out.numpy() - z['fbank'][:2]).max() < 2e-3


def test_emu_fbank_edge_cases():
    fb = lc._hip.Fbank(FB, cdll=emu_cdll())
    assert fb(torch.zeros(2, 399)).shape == (2, 0, 80)           # shorter than one window: empty
    out = fb(torch.zeros(1, 400 + 160 * 5))                       # silence: log floor everywhere -> zero after CMN
    assert out.shape == (1, 6, 80) and out.abs().max() < 1e-5
    fb23 = lc._hip.Fbank(dict(sample_frequency=16000), cdll=emu_cdll())  # torchaudio default 23 bins
    w = frontend.synth_waveforms(1, 4000, seed=9)
    ref = frontend.audio_featurizer(w, None, 'Fbank', dict(sample_frequency=16000))
    assert (fb23(w) - ref).abs().max() < 2e-3
    with pytest.raises(RuntimeError, match='512 points'):   # 40 ms at 16 kHz = 640 samples: kaldi pads to 1024
        lc._hip.Fbank(dict(sample_frequency=16000, num_mel_bins=40, frame_length=40), cdll=emu_cdll())
    with pytest.raises(RuntimeError, match='fbank_tile_kernel is instantiated'):
        lc._hip.Fbank(dict(sample_frequency=16000, num_mel_bins=40), cdll=emu_cdll(), kernel='tile')
    for bad in (dict(dither=1.0), dict(round_to_power_of_two=False)):
        with pytest.raises(NotImplementedError):
            lc._hip.Fbank(dict(FB, **bad), cdll=emu_cdll())
    with pytest.raises(RuntimeError, match='bad VTLN options'):   # (torchaudio asserts on these)
        lc._hip.Fbank(dict(FB, vtln_warp=1.1, vtln_low=10.0), cdll=emu_cdll())
    for bad in (dict(high_freq=8100.0), dict(low_freq=-1.0), dict(low_freq=7700.0, high_freq=-400.0), dict(sample_frequency=11025, high_freq=7600.0)):
        with pytest.raises(RuntimeError, match='bad band'):      # (get_mel_banks asserts 0 <= low < nyquist, 0 < high <= nyquist, low < high)
            lc._hip.Fbank(dict(FB, **bad), cdll=emu_cdll())
    with pytest.raises(RuntimeError, match='preemphasis_coefficient must be in'):
        lc._hip.Fbank(dict(FB, preemphasis_coefficient=1.5), cdll=emu_cdll())
    with pytest.raises(RuntimeError, match=r'\[4, 128\]'):         # (... and num_bins > 3)
        lc._hip.Fbank(dict(FB, num_mel_bins=3), cdll=emu_cdll())
    lc._hip.Fbank(dict(FB, high_freq=8000.0, low_freq=0.0, num_mel_bins=4), cdll=emu_cdll())   # the edges themselves are valid
    with pytest.raises(TypeError):
        lc._hip.Fbank(dict(FB, n_fft=512), cdll=emu_cdll())
    with pytest.raises(Exception, match='Invalid window type'):
        lc._hip.Fbank(dict(FB, window_type='kaiser'), cdll=emu_cdll())
    snip = lc._hip.Fbank(dict(FB, snip_edges=False), cdll=emu_cdll())
    assert snip.num_frames(16000) == 100 and snip.num_frames(79) == 0 and snip.num_frames(80) == 1
    with pytest.raises(RuntimeError, match='too short to be mirrored'):   # (torchaudio raises on these as well: its mirrored signal ends)
        snip(torch.zeros(1, 100))


@pytest.mark.skipif(os.environ.get('MV_SLOW_EMU') != '1', reason='~3 min under the emulator; set MV_SLOW_EMU=1 (covered on the GPU by test_gpu_backbones_long_and_short_utterances)')
def test_emu_campp_long_utterance_two_launch_dense_layers():
    """T = 372 frames -> 186 strided frames: two chunks of 160, segments 0 / 1 split over the chunks (camdense.hip, long utterances)"""
    cd, rel = lc.model_case(emu_cdll(), 'cpu', 'campp_short', frames=372, head=1)
    assert rel < 1e-2


@pytest.mark.parametrize('case', ['eres2net_tiny', 'eres2netv2_tiny'])
def test_emu_eres2net_tiny_end_to_end(case):
    cd, rel = lc.model_case(emu_cdll(), 'cpu', case, tol=1e-8, max_batch=1)  # fp32 operands: far inside the 1e-4 bar
    assert rel < 1e-4


def test_emu_eres2net_rejects_bad_arguments():
    from helpers import load_case
    man, sd, x, _, _ = load_case('eres2netv2_tiny')
    cfg = lc._hip.MvEres2Cfg()
    cfg.version, cfg.input_size, cfg.embd_dim = 2, 16, 64
    for i, n in enumerate([1, 1, 2, 1]):
        cfg.num_blocks[i] = n
    cfg.m_channels, cfg.mul_channel, cfg.expansion, cfg.base_width, cfg.scale, cfg.two_emb_layer = 16, 1, 2, 26, 2, 1
    m = lc._hip.Model('eres2net', cfg, sd, cdll=emu_cdll())
    with pytest.raises(RuntimeError, match='at least 9 frames'):
        m.forward(x[:, :8].contiguous())
    cfg.m_channels = 24  # not a multiple of 16
    with pytest.raises(RuntimeError, match='multiple of 16'):
        lc._hip.Model('eres2net', cfg, sd, cdll=emu_cdll())
    cfg.m_channels, cfg.expansion = 16, 4
    with pytest.raises(RuntimeError, match='expansion 2'):
        lc._hip.Model('eres2net', cfg, sd, cdll=emu_cdll())
    cfg.expansion = 2
    sd_missing = {k: v for k, v in sd.items() if k != 'fuse34.local_att.3.weight'}
    with pytest.raises(RuntimeError, match='fuse34.local_att.3.weight'):
        lc._hip.Model('eres2net', cfg, sd_missing, cdll=emu_cdll())


@pytest.mark.parametrize('idx', range(len(lc.FCM_BLOCK_CASES)))
def test_emu_fcm_block(idx):
    """one BasicResBlock per launch (fcmblock.hip): strided + shortcut tap, identity, bands, two time tiles, ragged sizes"""
    lc.fcm_block_case(emu_cdll(), 'cpu', seed=idx, **lc.FCM_BLOCK_CASES[idx])


@pytest.mark.parametrize('idx', range(len(lc.FCM_BLOCK_C1_CASES)))
def test_emu_fcm_block_with_first_conv(idx):
    """the head's first conv evaluated by the producers of the first block's kernel (no [B, F, T, 32] map in memory)"""
    lc.fcm_block_c1_case(emu_cdll(), 'cpu', seed=50 + idx, **lc.FCM_BLOCK_C1_CASES[idx])


@pytest.mark.parametrize('idx', range(len(lc.FCM_CASES)))
@pytest.mark.parametrize('impl', ['band', 'row'])
def test_emu_fcm_conv3x3(idx, impl):
    """the band kernel (LDS ring of input rows; 16-byte aligned output rows) and the one-row-per-workgroup kernel every other output layout
    falls to (here: rows 72 bytes apart)"""
    cfg = dict(lc.FCM_CASES[idx])
    if impl == 'row':
        if cfg.pop('strided_out', False):
            pytest.skip('the padded layout of the row-kernel arm replaces the [B, T, F, 32] layout of this case')
        cfg['padded_out'] = True
    lc.fcm_conv_case(emu_cdll(), 'cpu', seed=idx, **cfg)


def test_emu_ecapa_tiny_end_to_end():
    cd, rel = lc.model_case(emu_cdll(), 'cpu', 'ecapa_tiny')
    assert rel < 5e-3


@pytest.mark.skipif(os.environ.get('MV_SLOW_EMU') != '1', reason='~90 s under the emulator; set MV_SLOW_EMU=1 (covered on the GPU by test_gpu_parity)')
def test_emu_campp_short_end_to_end():
    """fp16 head (pinned: the creation-time calibration would run six more forwards under the emulator), then the fp32 head through the
    conv2d kernels (frequency-only stride, residual epilogue, rows cast)"""
    cd, rel = lc.model_case(emu_cdll(), 'cpu', 'campp_short', head=1)
    assert rel < 1e-2
    info = {1: None}
    cd32, _ = lc.model_case(emu_cdll(), 'cpu', 'campp_short', max_batch=1, info=info, head=2)
    assert info[1] == 1.0 and cd32 < 1e-5


@pytest.mark.skipif(os.environ.get('MV_SLOW_EMU') != '1', reason='~50 s under the emulator; set MV_SLOW_EMU=1 (covered on the GPU by test_gpu_native_model_matches_reference_golden[campp_c64])')
def test_emu_campp_64_initial_channels_runs_the_per_layer_dense_kernel():
    """CAMPPlus(init_channels=64): block 1 is 64 -> 448 channels wide, below what cam_dense_block_kernel takes, so its twelve layers run
    cam_dense_layer_kernel (camdense.hip); blocks 2 / 3 (224 -> 992, 496 -> 1008: channel counts that are no multiples of 64) the block kernel"""
    cd, rel = lc.model_case(emu_cdll(), 'cpu', 'campp_c64', head=1)
    assert cd < 2e-5 and rel < 1e-2, (cd, rel)


def test_emu_melspec_fft_kernel_and_dft_kernel():
    """default geometry (n_fft 400, 128 mels) = melspec_tile_kernel: edge frames with reflect padding, a masked row, feature
    rows beyond the LDS block (T = 241 > 212 rows) and fewer; every n_fft that is neither 400 nor a power of two runs the dense-DFT kernels"""
    assert _hip.MelSpec({}, cdll=emu_cdll()).info()['tile_kernel']
    assert _hip.MelSpec(dict(n_fft=512), cdll=emu_cdll()).info()['kernel'] == 'melspec_pow2_kernel'
    assert not _hip.MelSpec(dict(n_fft=600, win_length=600), cdll=emu_cdll()).info()['tile_kernel']   # neither 400 nor a power of two: dense DFT
    with pytest.raises(RuntimeError, match='Require f_min <= f_max'):   # (MelScale.__init__ raises ValueError on it)
        _hip.MelSpec(dict(f_min=4000.0, f_max=3000.0), cdll=emu_cdll())
    wav = frontend.synth_waveforms(2, 48000, seed=31)
    lc.melspec_case(emu_cdll(), 'cpu', wav, torch.tensor([0.71, 1.0]), {})          # T = 241: 212 rows in LDS, 29 through global
    lc.melspec_case(emu_cdll(), 'cpu', wav[:1, :5000 + 3], None, {})                # T = 26, odd length
    lc.melspec_case(emu_cdll(), 'cpu', wav[:1, :9000], None, dict(hop_length=160))  # other hop
    lc.melspec_case(emu_cdll(), 'cpu', wav[:1, :6000], None, dict(n_fft=600, win_length=600))   # dense DFT, default hop / mels


def test_emu_melspec_default_and_masked():
    wav = frontend.synth_waveforms(2, 2000, seed=12)
    lc.melspec_case(emu_cdll(), 'cpu', wav, torch.tensor([1.0, 0.45]), {})


def test_emu_melspec_other_geometry():
    """n_fft 256 with a shorter window = melspec_pow2_kernel on the zero-extended frame (every fourth bin of the 1024-point transform);
    n_fft 600 = dense DFT"""
    wav = frontend.synth_waveforms(1, 1500, seed=13)
    args = dict(sample_rate=16000, n_fft=256, win_length=200, hop_length=80, f_min=50, f_max=7000, n_mels=40)
    assert _hip.MelSpec(args, cdll=emu_cdll()).info()['kernel'] == 'melspec_pow2_kernel'
    lc.melspec_case(emu_cdll(), 'cpu', wav, None, args)
    lc.melspec_case(emu_cdll(), 'cpu', wav, None, dict(sample_rate=16000, n_fft=600, win_length=600, hop_length=150, n_mels=32))


def test_emu_melspec_pow2_kernel_readme_geometry():
    """the reference README's MelSpectrogram run (n_fft 1024, hop 320, 64 mels, f_max above Nyquist: README_en.md:263-269) and
    n_fft 512 with 128 mels (two mel passes), both on melspec_pow2_kernel: edge frames (reflect padding), a masked row, a ragged last
    quad of frames"""
    readme = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000, n_mels=64)
    assert _hip.MelSpec(readme, cdll=emu_cdll()).info()['kernel'] == 'melspec_pow2_kernel'
    wav = frontend.synth_waveforms(2, 7000, seed=14)
    lc.melspec_case(emu_cdll(), 'cpu', wav, torch.tensor([1.0, 0.6]), readme)
    lc.melspec_case(emu_cdll(), 'cpu', wav[:1, :3333], None, dict(n_fft=512, hop_length=128))
    lc.melspec_case(emu_cdll(), 'cpu', wav[:1, :2000], None, dict(n_fft=512, win_length=400, hop_length=200, center=False, n_mels=80))


def test_emu_melspec_pow2_kernel_matches_reference_wrapper_golden():
    z = np.load(os.path.join(GOLDEN, 'featurizer_ref.npz'))
    wav = frontend.synth_waveforms(4, 48000)[:1]   # the waveforms oracle/make_golden.py fed to the reference's wrapper
    readme = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000, n_mels=64)
    out = _hip.MelSpec(readme, cdll=emu_cdll())(wav).numpy()
    assert np.abs(out - z['mel_readme']).max() < 2e-4 * float(np.abs(z['mel_readme']).max())


@pytest.mark.parametrize('cfg', [dict(), dict(online=True), dict(B=5, T=9, C=72, A=64, ldx=80, centred=False),
                                 dict(B=2, T=33, C=64, A=128, online=True, wscale=1.0),
                                 # the ring kernel (C a multiple of 256, bounded logits): 3 / 7 / 1 / 2 tiles, A = 128 and 64, no centre
                                 dict(B=2, T=45, C=256), dict(B=2, T=100, C=512, A=64), dict(B=1, T=16, C=256), dict(B=1, T=1, C=256),
                                 dict(B=2, T=17, C=256, centred=False), dict(B=1, T=130, C=256, ldx=264)])
def test_emu_asp_pool(cfg):
    lc.asp_pool_case(emu_cdll(), 'cpu', **cfg)


@pytest.mark.parametrize('cfg', [dict(width=64, T=45, dil=3), dict(width=128, T=33, dil=4, B=1), dict(width=64, T=170, dil=2, B=1), dict(width=128, T=75, dil=3, B=2),
                                 dict(width=64, T=400, dil=2, B=1), dict(width=128, T=331, dil=4, B=1)])   # > 320 frames: two chunks with halo rows
def test_emu_res2net_fused_chain(cfg):
    lc.res2_chain_case(emu_cdll(), 'cpu', **cfg)


@pytest.mark.parametrize('cfg', [dict(B=2, T=45, width=64, dil=3), dict(B=1, T=100, width=128, dil=2), dict(B=3, T=170, width=128, dil=4)])
def test_emu_res2net_chain_saturates_at_the_fp16_range(cfg):
    """BatchNorm scales of 3e4: step outputs and next-input sums leave the fp16 range; the chain saturates at +-65504 (ring, direct and 5-tile forms)"""
    lc.res2_chain_case(emu_cdll(), 'cpu', seed=11, gain=3.0e4, **cfg)


def test_emu_res2net_chain_small_batch_form_carries_the_same_bits():
    """one utterance alone (three workgroups of the 5-tile form on <= 160-frame chunks with halo rows) against the same utterance in a batch
    that takes one workgroup per utterance: identical bits (the emulator's chip has 8 CUs: 5 utterances are not a small batch)"""
    lc.res2_chain_case(emu_cdll(), 'cpu', width=128, T=298, dil=3, B=5, alone_rows=1)
    lc.res2_chain_case(emu_cdll(), 'cpu', width=128, T=130, dil=4, B=5, alone_rows=1, seed=2)


def test_emu_fbank_variable_length_batch():
    """mv_fbank_forward_varlen: own frame count and time mean per row, zero rows behind it (reader.py:100-106 + collate_fn.py)"""
    wav = frontend.synth_waveforms(4, 2400, seed=17)
    lens = torch.tensor([2400, 1999, 400, 120])
    padded = wav.clone()
    for i, n in enumerate(lens):
        padded[i, n:] = -0.5
    out = _hip.Fbank(FB, cdll=emu_cdll())(padded, None, lens)
    assert out.shape == (4, 13, 80)
    for i, n in enumerate(lens.tolist()):
        if n < 400:
            assert out[i].abs().sum() == 0
            continue
        ref = frontend.audio_featurizer(wav[i, :n].unsqueeze(0), None, 'Fbank', FB)[0]
        assert (out[i, :ref.shape[0]] - ref).abs().max() < 2e-3 and out[i, ref.shape[0]:].abs().sum() == 0


@pytest.mark.parametrize('normalize', [True, False])
def test_emu_wave_prepare_int16(normalize):
    lc.wave_prepare_case(emu_cdll(), 'cpu', normalize=normalize)


def test_emu_fbank_odd_window_length():
    args = dict(sample_frequency=11025, num_mel_bins=40)  # 275-sample window (odd), 110-sample shift
    w = frontend.synth_waveforms(2, 3000, seed=21)
    lc.fbank_case(emu_cdll(), 'cpu', w, None, args)


def test_emu_fbank_tile_kernel_long_utterance_and_generic_kernel():
    """80 bins run fbank_tile_kernel: feature block in LDS up to 292 frames, second pass over global memory beyond that (311 frames
    here); other mel geometries (23 bins, 40 bins) run fbank_kernel.  All against the oracle."""
    wav = frontend.synth_waveforms(2, 400 + 160 * 310, seed=23)
    ratio = torch.tensor([0.83, 1.0])
    assert _hip.Fbank(FB, cdll=emu_cdll()).info() == {'tile_kernel': True, 'pass_steps': (28, 12)}
    assert not _hip.Fbank(dict(sample_frequency=16000), cdll=emu_cdll()).info()['tile_kernel']  # 23 bins: generic kernel
    lc.fbank_case(emu_cdll(), 'cpu', wav, ratio, FB)              # tile kernel, no LDS block (T = 311)
    lc.fbank_case(emu_cdll(), 'cpu', wav[:, :48000], ratio, FB)   # tile kernel, LDS block at its largest (T = 298)
    lc.fbank_case(emu_cdll(), 'cpu', wav[:1, :48160], None, FB)   # one frame more: back to the global second pass
    FB40 = dict(sample_frequency=16000, num_mel_bins=40)
    assert not _hip.Fbank(FB40, cdll=emu_cdll()).info()['tile_kernel']
    lc.fbank_case(emu_cdll(), 'cpu', wav[:1, :20000], ratio[:1], FB40)


def test_emu_fbank_long_utterance_chunked_and_single_workgroup_forms():
    """utterances beyond the LDS block on a chip the batch does not fill: several workgroups per utterance + the finish pass (caller workspace);
    without a workspace one workgroup per utterance.  Both against the oracle (1000 frames = 4 chunks; ragged lengths through the ratio mask) and
    BIT-IDENTICAL to each other and across batch sizes: an utterance's time sum is formed in one order (wave slot w adds its quads w, w + 8, ...;
    the chunk form hands the per-quad sums to the finish pass, which adds them in that order)."""
    wav = frontend.synth_waveforms(3, 400 + 160 * 999, seed=31)
    ratio = torch.tensor([0.41, 1.0, 0.77])
    lc.fbank_case(emu_cdll(), 'cpu', wav, ratio, FB)
    lc.fbank_case(emu_cdll(), 'cpu', wav[:1], None, FB)
    fb = _hip.Fbank(FB, cdll=emu_cdll())
    chunked = fb(wav, ratio)
    single = fb(wav, ratio, workspace=False)
    assert torch.equal(chunked, single)
    assert torch.equal(fb(wav[1:2], ratio[1:2]), chunked[1:2]) and torch.equal(fb(wav[2:3], ratio[2:3], workspace=False), chunked[2:3])
    # 3 s utterances, the ordinary sub-chip batch: the launcher cuts every utterance into about CUs / B chunks
    wav3 = frontend.synth_waveforms(9, 48000, seed=32)   # the emulator's chip has 8 CUs: 9 rows take the one-workgroup form, fewer the chunk form
    full = fb(wav3)
    for nb in (1, 2, 5):
        assert torch.equal(fb(wav3[:nb]), full[:nb])
    nsamp = torch.tensor([48000, 48000, 48000])
    assert torch.equal(fb(wav3[:3], num_samples=nsamp), fb(wav3[:3], torch.ones(3), workspace=False))   # the variable-length entry sums the same way


def test_emu_fbank_without_time_mean_is_the_reference_kaldi_fbank_module():
    """KaldiFbank.forward (featurizer.py:114-132 of the reference): kaldi.fbank per utterance, no mean subtraction -- the kernel with
    subtract_time_mean off, in the one-workgroup form, with rows beyond the LDS block (T > 292) and in the several-workgroups form"""
    fb = lc._hip.Fbank(dict(sample_frequency=16000, num_mel_bins=80), subtract_time_mean=False, cdll=emu_cdll())
    for B, L, ws in ((2, 4000, True), (1, 52000, True), (1, 52000, False), (3, 16001, True)):
        wav = frontend.synth_waveforms(B, L, seed=40 + B)
        ref = torch.stack([frontend.kaldi_fbank(row.unsqueeze(0), sample_frequency=16000, num_mel_bins=80) for row in wav])
        out = fb(wav, workspace=ws)
        assert out.shape == ref.shape and (out - ref).abs().max() < 2e-3, (B, L, ws, (out - ref).abs().max())


def test_emu_ecapa_without_global_context():
    """EcapaTdnn(global_context=False) (ecapa_tdnn.py:159, pooling.py:105): the attention conv sees the frames alone (C instead of 3C input channels, no
    per-utterance bias from the global mean / std) -- a constructor argument no golden uses; reference = the oracle's restatement of that branch"""
    import mvector.models as M
    from oracle import weights, models as omodels
    from helpers import cos_dist
    kw = dict(input_size=80, channels=[64, 64, 64, 64, 192], global_context=False)
    m = M.EcapaTdnn(**kw)
    sd = weights.make_state_dict(weights.shapes_of(m.state_dict()), 7)
    x = torch.randn(2, 50, 80, generator=torch.Generator().manual_seed(3)) * 2
    ref = omodels.ecapa_tdnn(sd, x, global_context=False)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        assert cos_dist(m(x), ref).max().item() < 1e-10      # the host package's torch forward
    cfg = lc._hip.MvEcapaCfg()
    cfg.input_size, cfg.embd_dim = 80, 192
    for i in range(5):
        cfg.channels[i], cfg.kernel_sizes[i], cfg.dilations[i] = kw['channels'][i], [5, 3, 3, 3, 1][i], [1, 2, 3, 4, 1][i]
    cfg.attention_channels, cfg.res2net_scale, cfg.se_channels, cfg.global_context = 128, 8, 128, 0
    emb = lc._hip.Model('ecapa', cfg, sd, cdll=emu_cdll()).forward(x).cpu()
    assert cos_dist(emb, ref).max().item() < 1e-5


@pytest.mark.parametrize('cls,kw,T', [
    ('EcapaTdnn', dict(input_size=80, channels=[64, 64, 64, 64, 192], res2net_scale=4, attention_channels=64, se_channels=64, embd_dim=256), 50),
    ('EcapaTdnn', dict(input_size=40, channels=[64, 64, 64, 64, 192], kernel_sizes=[5, 5, 3, 3, 1], dilations=[1, 2, 2, 2, 1]), 50),
    ('TDNN', dict(input_size=40, channels=256, embd_dim=128), 60),
    ('ERes2Net', dict(input_size=16, m_channels=16, num_blocks=[1, 1, 1, 1], embd_dim=128), 41),
    ('ERes2NetV2', dict(input_size=16, m_channels=16, num_blocks=[1, 1, 1, 1], embd_dim=64, base_width=32, scale=4), 41),
])
def test_emu_models_with_other_constructor_arguments(cls, kw, T):
    """constructor arguments the goldens do not use (widths, Res2Net scale, kernel sizes / dilations, embedding size, base width): the native
    handle built from the module's own configuration against the host package's torch forward of the same module (that forward and the
    state_dict layout are pinned to the reference at the goldens' arguments)"""
    import mvector.models as M
    from oracle import weights
    from helpers import cos_dist
    m = getattr(M, cls)(**kw)
    sd = weights.make_state_dict(weights.shapes_of(m.state_dict()), 9)
    x = torch.randn(2, T, kw['input_size'], generator=torch.Generator().manual_seed(4)) * 2
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        ref = m(x)
    ok, why = m._native_supported()
    assert ok, why
    kind = {'EcapaTdnn': 'ecapa', 'TDNN': 'tdnn', 'ERes2Net': 'eres2net', 'ERes2NetV2': 'eres2net'}[cls]
    emb = lc._hip.Model(kind, m._native_cfg(), sd, cdll=emu_cdll()).forward(x).cpu()
    assert cos_dist(emb, ref).max().item() < 1e-5


def test_unsupported_constructor_arguments_are_named_not_approximated():
    """what the native path does not build says so (and the CUDA forward raises NotImplementedError with that text): other pooling types, other
    CAM++ growth rates"""
    import mvector.models as M
    ok, why = M.EcapaTdnn(input_size=80, channels=[64, 64, 64, 64, 192], pooling_type='TSP')._native_supported()
    assert not ok and 'pooling_type' in why
    ok, why = M.CAMPPlus(input_size=80, growth_rate=16)._native_supported()
    assert not ok and 'growth_rate' in why


@pytest.mark.parametrize('frame_length', [20.0, 24.0, 25.0, 26.0, 27.0, 28.0, 30.0, 32.0])
def test_emu_fbank_other_frame_lengths_with_80_bins(frame_length):
    """kaldi.fbank's frame_length is a method argument (featurizer.py:128).  Windows of 385 .. 416 samples (25 / 26 ms at 16 kHz) run the 13-group instantiation of
    fbank_tile_kernel, every other even window the 16-group one -- which gave wrong features for windows of <= 384 samples until round 5 (20 ms: 9.9 off; its LDS window
    table was 448 taps, groups 14 / 15 read behind it; found by tools/emu_fuzz.py).  Both kernels, every window, against the oracle."""
    args = dict(sample_frequency=16000, num_mel_bins=80, frame_length=frame_length)
    assert lc._hip.Fbank(args, cdll=emu_cdll()).info()['tile_kernel']
    assert not lc._hip.Fbank(args, cdll=emu_cdll(), kernel='generic').info()['tile_kernel']
    wav = frontend.synth_waveforms(2, 24080, seed=3)
    ratio = torch.tensor([1.0, 0.6])
    lc.fbank_case(emu_cdll(), 'cpu', wav, ratio, args)
    lc.fbank_case(emu_cdll(), 'cpu', wav, ratio, args, kernel='generic')


@pytest.mark.parametrize('idx', range(len(lc.FBANK_ARG_CASES)))
def test_emu_fbank_arguments(idx):
    """the kaldi.fbank keyword arguments featurizer.py:128 forwards (frame length / shift, bin counts, sample rates incl. kaldi's 256-point FFT at 8 kHz, band edges,
    magnitude / linear outputs, DC / pre-emphasis switches, the five window types, snip_edges=False, subtract_mean, min_duration, VTLN warps) x (kernel, bare rows, true lengths):
    tests/layer_checks.py::FBANK_ARG_CASES, the list the device sweep (test_gpu_fbank_arguments) runs at 3 s"""
    lc.fbank_arguments_case(emu_cdll(), 'cpu', idx)


def test_emu_fbank_clip_of_exactly_min_duration_keeps_its_frames():
    lc.fbank_min_duration_edge(emu_cdll(), 'cpu')


S16_RANGE_CASES = [
    dict(cin=32, cout=32, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, peak=True),                                   # peak below the range: reported exactly
    dict(cin=32, cout=32, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, peak=True, x_scale=200.0, seed=2),            # outputs beyond 1023.5: clamped AND reported
    dict(cin=32, cout=48, ks=1, H=6, W=20, B=2, lo=-3.0e38, hi=3.0e38, peak=True, x_scale=200.0, with_res=True, seed=3),
    dict(cin=16, cout=16, ks=3, H=6, W=20, B=1, lo=0.0, hi=3.0e38, nan_at=(0, 2, 5, 3)),                          # a NaN input reaches its 9 x 16 outputs as NaN
    dict(cin=32, cout=32, ks=1, H=6, W=20, B=1, epi=1, nan_at=(0, 2, 5, 3), peak=True),
    dict(cin=32, cout=16, ks=1, H=4, W=20, B=1, epi=2, nan_at=(0, 1, 7, 30), peak=True),
]


@pytest.mark.parametrize('idx', range(len(S16_RANGE_CASES)))
def test_emu_conv2ds_reports_its_peak_and_keeps_nans(idx):
    """S16 maps saturate at |value| = 1023.5 and the clamps turned NaNs into finite bounds.  MvConv2dsDesc.peak reports the largest
    value a launch wanted to store (the CAM++ handle picks its exact head's gain from it and exposes saturation on real inputs); NaNs travel on."""
    lc.conv2ds_case(emu_cdll(), 'cpu', **S16_RANGE_CASES[idx])


@pytest.mark.parametrize('idx', range(len(lc.MELSPEC_ARG_CASES)))
def test_emu_melspec_arguments(idx):
    """MelSpectrogram(**method_args) beyond the shipped configurations (featurizer.py:41-42): Slaney mel scale / area normalisation, the three
    `normalized` modes, window_fn (+ wkwargs) evaluated once like torchaudio does, any positive power, centre off, other rates -- all of them tables
    of the same kernels (tests/layer_checks.py::MELSPEC_ARG_CASES; the device runs the list at 3 s: test_gpu_melspec_arguments)"""
    lc.melspec_arguments_case(emu_cdll(), 'cpu', idx)
