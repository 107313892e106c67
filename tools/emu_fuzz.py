"""Random launch geometries through the kernels on the SIMT emulator (no GPU), each checked against the same fp32 / fp64 evaluation the layer
tests use (tests/layer_checks.py) -- the hand-picked emulator cases cover the shapes somebody thought of, this covers the ones nobody did:
odd sizes around tile / ring / chunk boundaries, single rows and columns, ragged channel counts, operand combinations.

    python tools/emu_fuzz.py conv2ds 200                  # 200 random conv2ds cases, plain emulator
    python tools/emu_fuzz.py all 60 --mode lazy-dma       # every family, 60 cases each, under a checking mode of tools/emu_check.py
    python tools/emu_fuzz.py conv1d 100 --mode asan --seed 7 --jobs 8
    python tools/emu_fuzz.py conv2ds --replay "dict(B=1, H=3, ...)"      # one case again (the line a failure prints)
    python tools/emu_fuzz.py all 300 --device gpu --jobs 1 # the same generators against the product library on cuda:0 (a GPU box)

MV_FUZZ_STREAM=1: the workers print their own RESULT / FAIL / progress lines instead of the per-family summary (for runs under a time limit).
A case the launcher REFUSES (RuntimeError carrying the library's message) counts as "refused", not as a failure: refusing loudly is the contract
(include/mvector_hip.h); wrong values, NaNs, sanitizer reports, deadlocks and crashes are failures.  Modes: see tools/emu_check.py.
"""
import argparse
import glob
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]


# ---- generators: one random kwargs dict for the family's case function --------------------------------------------------------------------
BIG = False   # --device gpu: also the sizes the emulator would take minutes for (full-height maps, 10 s utterances, batches that fill the chip)


def _pick(r, small, big):
    return r.choice(small + big) if BIG else r.choice(small)


def g_conv2ds(r):
    for _ in range(100):
        kw = _g_conv2ds(r)
        if kw['B'] * kw['H'] * kw['W'] * kw['cin'] * kw['cout'] * kw['ks'] ** 2 <= 3e9:   # (the fp64 reference runs on the host)
            return kw
    return kw


def _g_conv2ds(r):
    ks = r.choice([1, 3, 3])
    stride = r.choice([1, 1, 2])
    ch = lambda: r.choice([13, 16, 26, 32, 39, 48, 64, 80, 96, 104, 128, 160, 208])
    kw = dict(B=_pick(r, [1, 1, 2, 3], [4, 16, 33]), H=_pick(r, [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 16, 17, 21, 24], [40, 41, 80]),
              W=_pick(r, [1, 2, 7, 15, 16, 17, 31, 32, 33, 40, 50, 65], [75, 149, 298, 301]), cin=ch(), cout=ch(), ks=ks, stride=stride, seed=r.randrange(1000))
    if stride == 2 and r.random() < 0.3:
        kw['stride_w'] = 1
    mode = r.random()
    if mode < 0.2:
        kw['with_res'] = True
    elif mode < 0.35:
        kw['with_sum'] = True
    elif mode < 0.45 and ks == 1:
        kw['epi'] = r.choice([1, 2])
    elif mode < 0.55 and ks == 1:
        kw.update(concat=True, epi=1, cin=2 * r.choice([16, 32, 48, 64]))
    if r.random() < 0.2:
        kw.update(hi=65504.0, lo=-65504.0)
    if r.random() < 0.15:
        kw['ring'] = r.choice([2, 3])
        kw['wgs'] = 1
    if r.random() < 0.15 and ks == 3:
        kw['rows'] = r.choice([1, 2, 3, 5, 8])
    if r.random() < 0.15:
        kw['spw'] = r.choice([2, 4, 8])
    if r.random() < 0.1:
        kw['nbw'] = r.choice([1, 2, 3])
    return kw


def g_conv1d(r):
    k = r.choice([1, 1, 3, 5])
    kw = dict(B=_pick(r, [1, 2, 3, 5, 7], [16, 64, 130]), T=_pick(r, [1, 2, 9, 31, 37, 63, 64, 65, 100, 127, 129, 160, 161, 200, 298, 305], [600, 998, 1501]),
              cin=r.choice([8, 16, 24, 64, 72, 80, 128, 136, 192, 320]),
              cout=r.choice([8, 16, 20, 40, 64, 128, 200, 256, 512]), k=k, dil=r.choice([1, 2, 3, 4]) if k > 1 else 1, seed=r.randrange(1000))
    if k > 1 and r.random() < 0.3:
        kw['pad_mode'] = 'zero'
        if r.random() < 0.4:
            kw['valid'] = True
    if r.random() < 0.2:
        kw.update(stride=2, pad_mode='zero')
    if r.random() < 0.2:
        kw['x_f32'] = True
    if r.random() < 0.2:
        kw['y_f32'] = True
    if r.random() < 0.15:
        kw['with_x2'] = True
    if r.random() < 0.2:
        kw.update(in_affine=True, pre_act=0)
    if r.random() < 0.2:
        kw.update(affine=False, pre_act=0)
    if r.random() < 0.2:
        kw.update(row_bias=True, post_act=2)
    if r.random() < 0.15:
        kw['gate_seg'] = r.choice([10, 25, 100])
    if r.random() < 0.35:
        kw['tile'] = r.choice([160, 256, 256])
    if kw.get('tile') == 256 and k == 1 and r.random() < 0.4:
        kw['stats'] = r.choice([1, 2])
    if k == 1 and 'tile' not in kw and r.random() < 0.15:
        kw.update(in_stats=True, y_f32=True, pre_act=0, affine=False)
    if r.random() < 0.2:
        kw['extra_ld'] = r.choice([0, 8, 56])
    if 'tile' in kw:   # the forced wide tiles belong to the plain fp16 path (anything else is refused -- loudly, but it would not test a kernel)
        for key in ('x_f32', 'with_x2', 'in_affine', 'in_stats'):
            kw.pop(key, None)
        if kw['tile'] == 256:
            kw['cout'] = r.choice([256, 512, 768])
            if r.random() < 0.5:   # persistent walks of several rounds on a few resident workgroups (dense tile ids; with >= 8 K stages the ring kernel's last
                #                    partial round leaves as 64 x 64 / 128 x 128 sub-tiles)
                kw['persist_blocks'] = r.choice([8, 16, 24, 40])
                if k == 1 and r.random() < 0.6:
                    kw['cin'] = r.choice([512, 576, 1024])
    if kw.get('in_stats'):
        kw['cout'] = r.choice([64, 128])
        kw.pop('stride', None)
    # reflect padding needs T > pad (F.pad's rule, and the reference's)
    pad = kw['dil'] * (k - 1) // 2
    if kw.get('pad_mode', 'reflect') == 'reflect' and kw['T'] <= pad:
        kw['T'] = pad + 1 + r.randrange(5)
    if kw.get('valid') and kw['T'] <= kw['dil'] * (k - 1):
        kw['T'] = kw['dil'] * (k - 1) + 1 + r.randrange(5)
    return kw


def g_res2(r):
    width = r.choice([64, 128])
    dil = r.choice([2, 3, 4])
    return dict(B=_pick(r, [1, 2, 3], [8, 40, 130]), T=_pick(r, [9, 17, 33, 45, 75, 100, 150, 160, 161, 170, 200], [298, 304, 305, 320, 321, 400, 998]), width=width, dil=dil, groups=8,
                seed=r.randrange(1000),
                alone_rows=r.choice([0, 0, 1]))


def g_asp(r):
    kw = dict(B=_pick(r, [1, 2, 3, 5], [16, 130]), T=_pick(r, [1, 2, 9, 45, 64, 65, 100, 160, 161, 298], [600, 998, 1501]), C=_pick(r, [72, 192, 256, 384], [1536, 3072]), A=r.choice([64, 128]),
              seed=r.randrange(1000))
    if r.random() < 0.3:
        kw['online'] = True
    if r.random() < 0.3:
        kw['ldx'] = kw['C'] + 8
    if r.random() < 0.3 and kw['T'] >= 9:
        kw['centred'] = False   # (the uncentred moments are the option for callers without a global mean; on one or two frames E[x^2] - mean^2 cancels to
        #                          ~1e-3 where the centred form gives the clamp -- the models always pass the mean, and never pool fewer than 9 frames)
    return kw


def g_time_stats(r):
    C = r.choice([8, 72, 520, 1024])
    return dict(B=_pick(r, [1, 2, 3, 7], [64, 256]), T=_pick(r, [1, 2, 29, 64, 150, 298], [998, 3001]), C=C, ld=C + r.choice([0, 8]), unbiased=r.choice([0, 1]), eps=r.choice([1e-12, 0.0]),
                seed=r.randrange(1000))


def g_linear(r):
    return dict(B=r.choice([1, 2, 5, 17, 33, 150]), K=r.choice([1, 7, 64, 100, 1024, 2100, 4100]), O=r.choice([1, 3, 16, 37, 128, 192]), act=r.choice([0, 1, 2, 3]), seed=r.randrange(1000))


def g_fbank(r):
    # samples per utterance around the frame / quad / chunk boundaries (25 ms window = 400, shift 160); a few utterances, ragged ratios or none
    n = _pick(r, [400, 401, 559, 560, 561, 1040, 4000, 8000, 16000, 16001, 24080, 48000, 52000], [112000, 160000, 480000])
    B = _pick(r, [1, 2, 3], [8, 40, 130, 256])
    if B * n > 16e6:
        B = 3
    kw = dict(B=B, L=n, ragged=r.random() < 0.5, bins=r.choice([80, 80, 80, 40, 23, 64, 128, 17]), seed=r.randrange(1000))
    kw['kernel'] = r.choice(['auto', 'auto', 'generic'])   # (80 bins at 16 kHz: auto = fbank_tile_kernel)
    if r.random() < 0.15:
        kw['cmn'] = False     # bare kaldi.fbank rows (KaldiFbank)
    elif r.random() < 0.15:
        kw['varlen'] = True   # true lengths per row (the evaluation path)
    if r.random() < 0.5:   # kaldi.fbank arguments other than the defaults (featurizer.py:128 passes method_args through)
        extra = {}
        if r.random() < 0.5:
            extra['frame_length'] = r.choice([10.0, 20.0, 24.0, 25.0, 26.0, 30.0, 32.0])
        if r.random() < 0.2:
            extra['sample_frequency'] = r.choice([8000, 11025, 16000, 22050])
        if r.random() < 0.3:
            extra['window_type'] = r.choice(['hamming', 'hanning', 'rectangular', 'blackman', 'povey'])
        if r.random() < 0.2:
            extra['snip_edges'] = False
        if r.random() < 0.15:
            extra['subtract_mean'] = True
        if r.random() < 0.1:
            extra['min_duration'] = r.choice([0.02, 0.5, 1.0])
        if r.random() < 0.5:
            extra['frame_shift'] = r.choice([8.0, 10.0, 12.5, 16.0])
        if r.random() < 0.4:
            extra['low_freq'] = r.choice([0.0, 20.0, 100.0])
        if r.random() < 0.4:
            extra['high_freq'] = r.choice([0.0, -400.0, 7600.0])
        if r.random() < 0.3:
            extra['preemphasis_coefficient'] = r.choice([0.0, 0.9])
        if r.random() < 0.3:
            extra['remove_dc_offset'] = False
        if r.random() < 0.2:
            extra['use_power'] = False
        if r.random() < 0.2:
            extra['use_log_fbank'] = False
        if r.random() < 0.2:   # the log-energy column
            extra['use_energy'] = True
            if r.random() < 0.5:
                extra['raw_energy'] = False
            if r.random() < 0.5:
                extra['energy_floor'] = r.choice([0.0, 1e-4, 10.0])
            if r.random() < 0.5:
                extra['htk_compat'] = True
        # more mel bins than a quarter of the transform's bins: filters narrower than two FFT bins, where a log energy IS one bin's power and the
        # per-bin rounding of a 512-point fp32 transform shows undamped (device fuzz r14b: 128 bins on kaldi's 128-point FFT at 8 kHz / 10 ms,
        # 29 of 832 k values 1e-3 .. 3e-3 from the fp64 arbiter where torch's 128-point fp32 transform stays within 7.4e-4) -- not a
        # configuration anybody featurises with; the generator keeps to at most half as many filters as bins
        sf = extra.get('sample_frequency', 16000)
        if extra.get('high_freq', 0.0) > 0.5 * sf:   # (beyond Nyquist: torchaudio's get_mel_banks asserts, the library refuses -- r15bt drew 7600 Hz at 11.025 kHz)
            extra['high_freq'] = 0.45 * sf
        size = int(sf * extra.get('frame_length', 25.0) * 0.001)
        padded = 1 << max(1, (size - 1).bit_length())
        if kw['bins'] > padded // 4:   # (at most half as many filters as the transform has bins)
            kw['bins'] = max(4, padded // 4 // 4 * 4)
        kw['extra'] = extra
    return kw


def g_melspec(r):
    # torchaudio.transforms.MelSpectrogram(**method_args) geometries (featurizer.py:41-42): the n_fft = 400 FFT kernel, the power-of-two FFT kernel
    # (n_fft <= 1024), the dense DFT for everything else; windows shorter than n_fft, hops, mel counts and band edges, centre on / off
    n_fft = r.choice([128, 200, 256, 320, 400, 400, 512, 512, 600, 1024])
    args = dict(n_fft=n_fft)
    if r.random() < 0.4:
        args['win_length'] = r.choice([n_fft, n_fft // 2, max(16, n_fft - 56)])
    if r.random() < 0.6:
        args['hop_length'] = r.choice([80, 100, 128, 160, 200, 320])
    if r.random() < 0.5:
        args['n_mels'] = r.choice([23, 32, 40, 64, 80, 128])
    if r.random() < 0.3:
        args.update(f_min=r.choice([0.0, 20.0, 50.0]), f_max=r.choice([3800.0, 7000.0, 8000.0, 14000.0]))
    if r.random() < 0.2:
        args['center'] = False
    if r.random() < 0.2:
        args['power'] = r.choice([1.0, 2.0])
    if r.random() < 0.2:   # zeros around the signal / the centre extension in torch.stft's other modes (the extended signal goes through the workspace)
        args['pad'] = r.choice([1, 7, 100, n_fft])
    if r.random() < 0.2:
        args['pad_mode'] = r.choice(['constant', 'replicate', 'circular', 'reflect'])
    L = _pick(r, [n_fft, n_fft + 1, 1500, 2000, 3333, 5003, 9000], [48000, 160000])
    if L < n_fft:
        L = n_fft
    return dict(B=_pick(r, [1, 2, 3], [8, 64, 256]) if L < 100000 else 2, L=L, ragged=r.random() < 0.4, args=args, seed=r.randrange(1000))


def g_fcm_block(r):
    sf = r.choice([1, 2])
    kw = dict(B=_pick(r, [1, 2, 3, 9], [32, 256]), Fin=r.choice([1, 2, 3, 4, 5, 6, 9, 12, 20]), T=_pick(r, [5, 16, 33, 45, 62, 70, 100, 318, 319, 330], [298, 998]), sf=sf,
              seed=r.randrange(1000))
    if sf == 2 and r.random() < 0.3:
        kw['strided_out'] = True
    return kw


def g_model(r):
    # whole tiny backbones on one utterance of a random length against the oracle: the host side's choices (tile shapes, chunk forms, fused or
    # stand-alone statistics, ring / direct Res2Net forms) all follow the frame count
    case = r.choice(['ecapa_tiny', 'ecapa_tiny', 'tdnn', 'eres2net_tiny', 'eres2netv2_tiny'])
    T = _pick(r, [20, 23, 31, 33, 47, 64, 65, 97, 100, 127, 128, 129, 159, 160, 161, 163, 200, 257], [298, 305, 321, 640, 998])
    return dict(case=case, frames=T)


def g_fcm_conv(r):
    # the single 3x3 FCM convs (band kernel): plain / strided in frequency, with the strided 1x1 shortcut tap or the identity residual
    mode2 = r.choice([0, 0, 1, 2])
    kw = dict(B=_pick(r, [1, 2, 3], [32]), Fin=r.choice([1, 2, 3, 5, 6, 9, 12, 20]), T=_pick(r, [5, 16, 33, 45, 70, 100, 319, 320, 321, 330], [298, 998]), sf=r.choice([1, 2]) if mode2 == 0 else 1,
              mode2=mode2, seed=r.randrange(1000))
    if mode2 == 1:
        kw['sf2'] = r.choice([1, 2])
    if mode2 == 0 and kw['sf'] == 2 and r.random() < 0.4:
        kw['strided_out'] = True
    return kw


def g_fcm_c1(r):
    return dict(B=_pick(r, [1, 2, 3], [64]), F=r.choice([3, 5, 9, 12, 40, 80]), T=_pick(r, [5, 16, 45, 70, 131, 319, 330], [298, 998]), seed=r.randrange(1000), outlier=False)


def g_window(r):
    kw = dict(B=_pick(r, [1, 2, 3], [64]), T=_pick(r, [9, 33, 70, 150, 300], [298, 998]), F_=r.choice([16, 40, 80]), k=5, cout=r.choice([64, 256, 512]), seed=r.randrange(1000))
    if kw['cout'] % 256 == 0 and r.random() < 0.5:
        kw['tile'] = 256
    return kw


def g_small(r):
    # the small element-wise / reduction kernels of the paths: TSTP, the ERes2Net stem, BN + ReLU rows, int16 wave preparation
    what = r.choice(['tstp', 'first', 'bn_relu', 'wave'])
    if what == 'tstp':
        s16 = r.random() < 0.5   # (the S16 map needs rows of whole 16-channel units: the case's ld is C + 8)
        return dict(what=what, B=r.choice([1, 2, 3]), H=r.choice([1, 2, 5, 10]), W=_pick(r, [2, 3, 17, 38, 75], [298]), C=r.choice([24, 72, 120]) if s16 else r.choice([16, 72, 128]), s16=s16,
                    seed=r.randrange(1000))
    if what == 'first':
        return dict(what=what, B=r.choice([1, 2]), T=_pick(r, [9, 33, 50, 131], [298]), F_=r.choice([16, 80]), C=r.choice([16, 32]), s16=r.random() < 0.5, seed=r.randrange(1000))
    if what == 'bn_relu':
        C = r.choice([8, 72, 520, 1024])
        return dict(what=what, rows=_pick(r, [1, 5, 77, 300], [76288]), C=C, ldx=C + r.choice([0, 8]), ldy=C + r.choice([0, 24]), seed=r.randrange(1000))
    return dict(what=what, B=r.choice([2, 3, 5]), L=_pick(r, [1, 7, 300, 5000, 16001], [480000]), normalize=r.random() < 0.5, seed=r.randrange(1000))


def _budget(gen, cost, limit):
    """resample until the host-side reference of the case is affordable"""
    def g(r):
        for _ in range(200):
            kw = gen(r)
            if cost(kw) <= limit:
                break
        return kw
    return g


g_conv1d = _budget(g_conv1d, lambda k: k['B'] * k['T'] * k['cin'] * k['cout'] * k['k'], 4e9)
g_res2 = _budget(g_res2, lambda k: k['B'] * k['T'] * k['width'] ** 2 * 3 * 7, 6e9)
g_asp = _budget(g_asp, lambda k: k['B'] * k['T'] * k['C'] * k['A'], 3e9)
g_time_stats = _budget(g_time_stats, lambda k: k['B'] * k['T'] * k['C'], 3e8)
g_fcm_block = _budget(g_fcm_block, lambda k: k['B'] * k['T'] * k['Fin'] * 32 * 32 * 9 * 2, 4e9)
g_melspec = _budget(g_melspec, lambda k: k['B'] * k['L'], 4e6)

FAMILIES = {'conv2ds': g_conv2ds, 'conv1d': g_conv1d, 'res2': g_res2, 'asp_pool': g_asp, 'time_stats': g_time_stats, 'linear': g_linear,
            'fbank': g_fbank, 'melspec': g_melspec, 'fcm_block': g_fcm_block, 'model': g_model, 'fcm_conv': g_fcm_conv, 'fcm_c1': g_fcm_c1, 'window': g_window,
            'small': g_small}


DEVICE = 'cpu'   # 'cpu' = the emulator build of the kernels; 'cuda' = the product library on the GPU


def run_case(family, kw):
    import torch
    import layer_checks as lc
    if DEVICE == 'cpu':
        from emu_lib import emu_cdll
        cdll = emu_cdll()
    else:
        from mvector import _hip
        cdll = _hip.lib()
    if family == 'conv2ds':
        lc.conv2ds_case(cdll, DEVICE, **kw)
    elif family == 'conv1d':
        lc.conv1d_case(cdll, DEVICE, **kw)
    elif family == 'res2':
        lc.res2_chain_case(cdll, DEVICE, **kw)
    elif family == 'asp_pool':
        lc.asp_pool_case(cdll, DEVICE, **kw)
    elif family == 'time_stats':
        lc.time_stats_case(cdll, DEVICE, **kw)
    elif family == 'linear':
        lc.linear_case(cdll, DEVICE, **kw)
    elif family == 'fbank':
        from oracle import frontend
        wav = frontend.synth_waveforms(kw['B'], kw['L'], seed=kw['seed'])
        ratio = None
        if kw['ragged'] and kw['B'] > 1:
            g = torch.Generator().manual_seed(kw['seed'])
            ratio = torch.rand(kw['B'], generator=g) * 0.8 + 0.2
            ratio[0] = 1.0
        args = dict(dict(sample_frequency=16000, num_mel_bins=kw['bins']), **kw.get('extra', {}))
        cmn, ns = kw.get('cmn', True), None
        if kw.get('varlen'):
            g = torch.Generator().manual_seed(kw['seed'] + 1)
            ns = (torch.rand(kw['B'], generator=g) * kw['L']).long().clamp(min=1)
            ns[0] = kw['L']
            ratio = None
        if not cmn:
            ratio = None
        if not args.get('snip_edges', True):   # rows the reference raises on (its mirrored signal ends before the last frame): refused / zero rows here
            size, shift = int(args['sample_frequency'] * args.get('frame_length', 25.0) * 0.001), int(args['sample_frequency'] * args.get('frame_shift', 10.0) * 0.001)

            def mirrors(n):
                m, pad = (n + shift // 2) // shift, size // 2 - shift // 2
                return m == 0 or (pad <= n and (m - 1) * shift - pad + size <= 2 * n)
            if not all(mirrors(int(n)) for n in (ns.tolist() if ns is not None else [kw['L']])):
                raise RuntimeError('mv_fbank_forward: (fuzzer) snip_edges=False on a signal too short to mirror')
        lc.fbank_case(cdll, DEVICE, wav, ratio, args, kernel=kw.get('kernel', 'auto'), cmn=cmn, num_samples=ns)
    elif family == 'melspec':
        from oracle import frontend
        wav = frontend.synth_waveforms(kw['B'], kw['L'], seed=kw['seed'])
        ratio = None
        if kw['ragged'] and kw['B'] > 1:
            g = torch.Generator().manual_seed(kw['seed'])
            ratio = torch.rand(kw['B'], generator=g) * 0.8 + 0.2
            ratio[0] = 1.0
        lc.melspec_case(cdll, DEVICE, wav, ratio, kw['args'])
    elif family == 'fcm_block':
        lc.fcm_block_case(cdll, DEVICE, **kw)
    elif family == 'fcm_conv':
        lc.fcm_conv_case(cdll, DEVICE, **kw)
    elif family == 'fcm_c1':
        lc.fcm_block_c1_case(cdll, DEVICE, **kw)
    elif family == 'window':
        lc.conv1d_window_case(cdll, DEVICE, **kw)
    elif family == 'small':
        kw = dict(kw)
        what = kw.pop('what')
        {'tstp': lc.tstp_case, 'first': lc.conv2d_first_case, 'bn_relu': lc.bn_relu_rows_case, 'wave': lc.wave_prepare_case}[what](cdll, DEVICE, **kw)
    elif family == 'model':
        lc.model_case(cdll, DEVICE, kw['case'], frames=kw['frames'])
    else:
        raise SystemExit(f'unknown family {family}')


def worker(family, n, seed):
    r = random.Random(seed)
    ok = refused = 0
    failures = []
    for _ in range(n):
        kw = FAMILIES[family](r)
        if (ok + refused + len(failures)) % 10 == 0 and ok + refused + len(failures) > 0:
            print(f'  progress {family} seed {seed}: {ok} ok, {refused} refused, {len(failures)} failed so far', file=sys.stderr, flush=True)
        try:
            run_case(family, kw)
            ok += 1
        except RuntimeError as ex:
            msg = str(ex)
            if msg.startswith('libmvector_hip:') or 'mv_' in msg or family.split('_')[0] in msg or 'unsupported' in msg or 'must' in msg:
                refused += 1
                if os.environ.get('MV_FUZZ_VERBOSE'):
                    print(f'  refused: {msg[:160]}   {kw}', flush=True)
            else:
                failures.append((kw, f'RuntimeError: {msg[:200]}'))
        except AssertionError as ex:
            failures.append((kw, f'AssertionError: {str(ex)[:200]}'))
        except Exception as ex:  # noqa: BLE001  (whatever a case throws is a finding)
            failures.append((kw, f'{type(ex).__name__}: {str(ex)[:200]}'))
    print(f'RESULT {family} seed {seed}: {ok} ok, {refused} refused, {len(failures)} FAILED', flush=True)
    for kw, why in failures:
        print(f'  FAIL {family} "dict({", ".join(f"{k}={v!r}" for k, v in kw.items())})"  {why}', flush=True)
    return len(failures)


def mode_env(mode):
    env = dict(os.environ)
    for k in ('MV_EMU_SCHED', 'MV_EMU_SANITIZE', 'MV_EMU_POISON', 'MV_EMU_DMA', 'MV_EMU_LDS'):
        env.pop(k, None)
    llvm = '/opt/rocm/lib/llvm'
    if mode == 'asan':
        rt = glob.glob(llvm + '/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')[0]
        env.update(MV_EMU_SANITIZE='address', LD_PRELOAD=rt, ASAN_SYMBOLIZER_PATH=llvm + '/bin/llvm-symbolizer',
                   ASAN_OPTIONS='detect_leaks=0:verify_asan_link_order=0:halt_on_error=1')
    elif mode == 'ubsan':
        rt = glob.glob(llvm + '/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so')[0]
        env.update(MV_EMU_SANITIZE='undefined', LD_PRELOAD=rt, UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    elif mode == 'poison':
        env['MV_EMU_POISON'] = '1'
    elif mode.startswith('lazy-lds') or mode.startswith('lazy-all'):
        env.update(MV_EMU_POISON='1', MV_EMU_LDS='lazy')
        if mode.startswith('lazy-all'):
            env['MV_EMU_DMA'] = 'lazy'
        if '+' in mode:
            env['MV_EMU_SCHED'] = mode.split('+', 1)[1]
    elif mode.startswith('lazy-dma'):
        env.update(MV_EMU_POISON='1', MV_EMU_DMA='lazy')
        if '+' in mode:
            env['MV_EMU_SCHED'] = mode.split('+', 1)[1]
    elif mode != 'plain':
        env['MV_EMU_SCHED'] = mode
    return env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('family', help='one of ' + ', '.join(sorted(FAMILIES)) + ', a comma-separated list of them, or all')
    ap.add_argument('n', type=int, nargs='?', default=50)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--mode', default='plain')
    ap.add_argument('--jobs', type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument('--device', default='emu', choices=['emu', 'gpu'])
    ap.add_argument('--replay', default='')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    global DEVICE, BIG
    DEVICE = 'cuda' if args.device == 'gpu' else 'cpu'
    BIG = args.device == 'gpu'
    if args.replay:
        run_case(args.family, eval(args.replay))  # noqa: S307  (a developer's own command line)
        print('ok')
        return
    if args.worker:
        sys.exit(1 if worker(args.family, args.n, args.seed) else 0)
    env = mode_env(args.mode)
    if args.device == 'emu':
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'build_emu.py')], env={**env, 'LD_PRELOAD': ''}, stdout=subprocess.DEVNULL)
    fams = sorted(FAMILIES) if args.family == 'all' else args.family.split(',')
    assert all(f in FAMILIES for f in fams), fams
    t0 = time.time()
    bad = 0
    for fam in fams:
        per = -(-args.n // args.jobs)
        # (each worker's host-side references on a few threads: a dozen torch processes with one thread per core each thrash the box -- the first
        # GPU run of this tool, r13c, spent its whole time limit that way and printed nothing)
        if os.environ.get('MV_FUZZ_STREAM'):
            wenv = {**env, 'OMP_NUM_THREADS': str(max(1, min(8, (os.cpu_count() or 8) // max(1, args.jobs)))), 'PYTHONUNBUFFERED': '1'}
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), fam, str(per), '--seed', str(args.seed * 1000 + j), '--worker', '--device', args.device], env=wenv, cwd=ROOT)
                  for j in range(args.jobs)]
            bad += sum(1 for q in ps if q.wait() != 0)
            continue
        wenv = {**env, 'OMP_NUM_THREADS': str(max(1, min(8, (os.cpu_count() or 8) // max(1, args.jobs)))), 'PYTHONUNBUFFERED': '1'}
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), fam, str(per), '--seed', str(args.seed * 1000 + j), '--worker', '--device', args.device], env=wenv, cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for j in range(args.jobs)]
        if os.environ.get('MV_FUZZ_STREAM'):   # (a run under somebody's time limit: let the workers talk themselves -- a cut-off run keeps what they said)
            raise SystemExit('MV_FUZZ_STREAM is handled before the workers are started')
        ok = refused = failed = crashed = 0
        for p in procs:
            out, _ = p.communicate()
            res = [l for l in out.splitlines() if l.startswith('RESULT')]
            if not res:   # the worker died (sanitizer report, deadlock abort, crash): show the tail
                crashed += 1
                print(f'  CRASH {fam} worker rc={p.returncode}:\n    ' + '\n    '.join(out.splitlines()[-25:]), flush=True)
                continue
            w = res[-1].split(':')[1].split(',')
            ok += int(w[0].split()[0]); refused += int(w[1].split()[0]); failed += int(w[2].split()[0])
            for l in out.splitlines():
                if l.startswith('  FAIL'):
                    print(l, flush=True)
        print(f'{args.mode:16s} {fam:10s} {ok} ok, {refused} refused, {failed} failed, {crashed} workers crashed   ({time.time() - t0:.0f} s)', flush=True)
        bad += failed + crashed
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
