"""Host-tensor front-end: batched torch implementation of Kaldi Fbank / MelSpectrogram + CMN + mask.

Used only for CPU tensors (``use_gpu=False`` predictors and DataLoader worker processes, exactly where the
reference runs its featurizer, mvector/data_utils/featurizer.py:53-91).  CUDA tensors never come here: they go
to the HIP kernels.  The arithmetic is the one of torchaudio 2.4.0's ``kaldi.fbank`` / ``MelSpectrogram`` with
whole-batch tensor ops instead of the reference's per-utterance Python loop.
"""
import functools
import math

import torch

_FBANK_KEYS = {'sample_frequency': 16000.0, 'frame_length': 25.0, 'frame_shift': 10.0, 'num_mel_bins': 23,
               'low_freq': 20.0, 'high_freq': 0.0, 'preemphasis_coefficient': 0.97, 'remove_dc_offset': True,
               'use_power': True, 'use_log_fbank': True, 'window_type': 'povey', 'blackman_coeff': 0.42, 'snip_edges': True,
               'subtract_mean': False, 'min_duration': 0.0, 'vtln_warp': 1.0, 'vtln_low': 100.0, 'vtln_high': -500.0,
               'use_energy': False, 'raw_energy': True, 'energy_floor': 1.0, 'htk_compat': False}
# not implemented beyond these values: dither draws random numbers, FFT sizes that are not powers of two.  (use_energy adds a column
# AudioFeaturizer.feature_dim -- like the reference's, featurizer.py:110-111 -- does not count.)
_FBANK_FIXED = {'dither': (0.0,), 'round_to_power_of_two': (True,), 'channel': (-1, 0)}
_FBANK_IGNORED = ()
_WINDOWS = ('povey', 'hamming', 'hanning', 'rectangular', 'blackman')
_MEL_KEYS = {'sample_rate', 'n_fft', 'win_length', 'hop_length', 'f_min', 'f_max', 'pad', 'n_mels', 'power',
             'normalized', 'center', 'pad_mode', 'onesided', 'norm', 'mel_scale', 'window_fn', 'wkwargs'}


def validate_args(method, args):
    if method == 'Fbank':
        for k, v in args.items():
            if k == 'window_type' and v not in _WINDOWS:
                raise Exception('Invalid window type ' + str(v))
            if k in _FBANK_KEYS or k in _FBANK_IGNORED:
                continue
            if k in _FBANK_FIXED:
                if v not in _FBANK_FIXED[k]:
                    raise NotImplementedError(f'Fbank argument {k}={v!r} is not implemented')
            else:
                raise TypeError(f"fbank() got an unexpected keyword argument '{k}'")
    elif method == 'MelSpectrogram':
        for k in args:
            if k not in _MEL_KEYS:
                raise TypeError(f"MelSpectrogram got an unexpected keyword argument '{k}'")
        if args.get('onesided') not in (None, True) or args.get('power', 2.0) is None:
            raise NotImplementedError('MelSpectrogram option not implemented')
        if args.get('pad_mode', 'reflect') not in ('reflect', 'constant', 'replicate', 'circular'):
            raise NotImplementedError(f"Unrecognised padding mode {args.get('pad_mode')}")
        if args.get('norm') not in (None, 'slaney'):
            raise ValueError('norm must be one of None or "slaney"')
        if args.get('mel_scale', 'htk') not in ('htk', 'slaney'):
            raise ValueError('mel_scale should be one of "htk" or "slaney".')
        if args.get('normalized', False) not in (False, True, 'window', 'frame_length'):
            raise ValueError(f"Invalid normalized parameter: {args.get('normalized')}")


def _window(window_type, size, blackman_coeff):
    if window_type == 'povey':
        return torch.hann_window(size, periodic=False).pow(0.85)
    if window_type == 'hanning':
        return torch.hann_window(size, periodic=False)
    if window_type == 'hamming':
        return torch.hamming_window(size, periodic=False, alpha=0.54, beta=0.46)
    if window_type == 'rectangular':
        return torch.ones(size)
    a = 2 * math.pi / (size - 1)
    i = torch.arange(size, dtype=torch.float32)
    return blackman_coeff - 0.5 * torch.cos(a * i) + (0.5 - blackman_coeff) * torch.cos(2 * a * i)


def _vtln_warp_mel(mel, low, high, vtln_low, vtln_high, warp):
    """kaldi's vocal-tract-length warp of mel-scale filter edges (torchaudio.compliance.kaldi.vtln_warp_mel_freq): a 3-piece linear map of the
    frequency axis, the identity outside [low, high]"""
    f = 700.0 * ((mel / 1127.0).exp() - 1.0)
    lo_cut, hi_cut, scale = vtln_low * max(1.0, warp), vtln_high * min(1.0, warp), 1.0 / warp
    if not (lo_cut > low and hi_cut < high):
        raise AssertionError('VTLN cut-offs must lie inside (low_freq, high_freq)')
    left = low + (scale * lo_cut - low) / (lo_cut - low) * (f - low)
    right = high + (high - scale * hi_cut) / (high - hi_cut) * (f - high)
    r = torch.where(f < lo_cut, left, torch.where(f < hi_cut, scale * f, right))
    r = torch.where((f < low) | (f > high), f, r)
    return 1127.0 * (1.0 + r / 700.0).log()


@functools.lru_cache(maxsize=8)
def _fbank_tables(sf, frame_length, frame_shift, nbins, low, high, window_type, blackman_coeff, vtln_warp=1.0, vtln_low=100.0, vtln_high=-500.0):
    size = int(sf * frame_length * 0.001)
    shift = int(sf * frame_shift * 0.001)
    padded = max(2, 1 << (size - 1).bit_length())
    window = _window(window_type, size, blackman_coeff)
    nfft_bins = padded // 2
    high = high + 0.5 * sf if high <= 0.0 else high
    if nbins <= 3:   # (torchaudio.compliance.kaldi.get_mel_banks asserts on both)
        raise AssertionError('Must have at least 3 mel bins')
    if not (0.0 <= low < 0.5 * sf and 0.0 < high <= 0.5 * sf and low < high):
        raise AssertionError(f'Bad values in options: low-freq {low} and high-freq {high} vs. nyquist {0.5 * sf}')
    mel_lo = 1127.0 * math.log(1.0 + low / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + high / 700.0)
    delta = (mel_hi - mel_lo) / (nbins + 1)
    idx = torch.arange(nbins).unsqueeze(1)
    left, center, right = mel_lo + idx * delta, mel_lo + (idx + 1.0) * delta, mel_lo + (idx + 2.0) * delta
    mel = (1127.0 * (1.0 + (sf / padded) * torch.arange(nfft_bins) / 700.0).log()).unsqueeze(0)
    if vtln_warp == 1.0:
        banks = torch.clamp(torch.min((mel - left) / (center - left), (right - mel) / (right - center)), min=0.0)
    else:
        vtln_high = vtln_high + 0.5 * sf if vtln_high < 0.0 else vtln_high
        if not (low < vtln_low < high and 0.0 < vtln_high < high and vtln_low < vtln_high):
            raise AssertionError(f'Bad values in options: vtln-low {vtln_low} and vtln-high {vtln_high}, versus low-freq {low} and high-freq {high}')
        left, center, right = (_vtln_warp_mel(m, low, high, vtln_low, vtln_high, vtln_warp) for m in (left, center, right))
        up, down = (mel - left) / (center - left), (right - mel) / (right - center)
        zero = torch.zeros_like(up)
        banks = torch.where((mel > left) & (mel <= center), up, torch.where((mel > center) & (mel < right), down, zero))
    banks = torch.nn.functional.pad(banks, (0, 1))  # zero weight on the Nyquist bin
    return size, shift, padded, window, banks.t().contiguous()


def fbank_batch(wav, args):
    """[B, L] -> [B, m, num_mel_bins] log-mel energies (no CMN)."""
    a = dict(_FBANK_KEYS)
    a.update({k: v for k, v in args.items() if k in _FBANK_KEYS})
    size, shift, padded, window, banks_t = _fbank_tables(float(a['sample_frequency']), float(a['frame_length']),
                                                          float(a['frame_shift']), int(a['num_mel_bins']),
                                                          float(a['low_freq']), float(a['high_freq']), a['window_type'],
                                                          float(a['blackman_coeff']), float(a['vtln_warp']), float(a['vtln_low']), float(a['vtln_high']))
    B, L = wav.shape
    ncol = int(a['num_mel_bins']) + int(bool(a['use_energy']))
    eps = torch.finfo(torch.float32).eps

    def log_energy(fr):   # kaldi._get_log_energy
        le = fr.pow(2).sum(2).clamp(min=eps).log()
        return le if float(a['energy_floor']) == 0.0 else le.clamp(min=math.log(float(a['energy_floor'])))
    if L < float(a['min_duration']) * float(a['sample_frequency']) or (a['snip_edges'] and L < size):
        return wav.new_zeros((B, 0, ncol))
    if not a['snip_edges']:  # (L + shift // 2) // shift frames over the signal mirrored at both ends (kaldi._get_strided)
        m = (L + shift // 2) // shift
        pad = size // 2 - shift // 2
        rev = torch.flip(wav, [1])
        wav = torch.cat((rev[:, L - pad:], wav, rev), dim=1) if pad > 0 else torch.cat((wav[:, -pad:], rev), dim=1)
        if m == 0:
            return wav.new_zeros((B, 0, ncol))
        if (m - 1) * shift + size > wav.shape[1] or pad > L:
            raise RuntimeError('snip_edges=False: the signal is too short to be mirrored over its frames')
        wav = wav[:, :(m - 1) * shift + size]
    frames = wav.unfold(1, size, shift)  # [B, m, size]
    if a['remove_dc_offset']:
        frames = frames - frames.mean(dim=2, keepdim=True)
    if a['use_energy'] and a['raw_energy']:
        energy = log_energy(frames)
    pc = float(a['preemphasis_coefficient'])
    if not 0.0 <= pc <= 1.0:   # (kaldi._get_waveform_and_window_properties asserts)
        raise AssertionError('`preemphasis_coefficient` must be between [0,1]')
    if pc != 0.0:
        frames = frames - pc * torch.cat([frames[..., :1], frames[..., :-1]], dim=2)
    frames = frames * window
    if a['use_energy'] and not a['raw_energy']:
        energy = log_energy(frames)
    spec = torch.fft.rfft(frames, n=padded).abs()
    if a['use_power']:
        spec = spec.pow(2.0)
    mel = spec @ banks_t
    if a['use_log_fbank']:
        mel = torch.clamp(mel, min=eps).log()
    if a['use_energy']:
        mel = torch.cat((mel, energy.unsqueeze(2)), dim=2) if a['htk_compat'] else torch.cat((energy.unsqueeze(2), mel), dim=2)
    if a['subtract_mean']:
        mel = mel - mel.mean(dim=1, keepdim=True)
    return mel


def _hz_to_mel(f, slaney):
    if not slaney:
        return 2595.0 * math.log10(1.0 + f / 700.0)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return min_log_hz / f_sp + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp


def _mel_to_hz(m, slaney):
    if not slaney:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    f = f_sp * m
    log_t = m >= min_log_mel
    f[log_t] = min_log_hz * torch.exp(logstep * (m[log_t] - min_log_mel))
    return f


@functools.lru_cache(maxsize=8)
def _mel_fbank(sr, n_fft, f_min, f_max, n_mels, mel_scale, norm):
    """torchaudio.functional.melscale_fbanks: [n_fft // 2 + 1, n_mels]"""
    slaney = mel_scale == 'slaney'
    all_freqs = torch.linspace(0, sr // 2, n_fft // 2 + 1)
    m_pts = torch.linspace(_hz_to_mel(f_min, slaney), _hz_to_mel(f_max, slaney), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, slaney)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    if norm == 'slaney':
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb


def melspec_batch(wav, args):
    """[B, L] -> [B, frames, n_mels] mel spectrogram (no log, no CMN): torchaudio.transforms.MelSpectrogram(**args)"""
    sr = int(args.get('sample_rate', 16000))
    n_fft = int(args.get('n_fft', 400))
    win = args.get('win_length')
    win = int(win if win is not None else n_fft)
    hop = args.get('hop_length')
    hop = int(hop if hop is not None else win // 2)
    f_max = args.get('f_max')
    f_max = float(f_max if f_max is not None else sr // 2)
    if float(args.get('f_min', 0.0)) > f_max:   # torchaudio.transforms.MelScale.__init__
        raise ValueError(f"Require f_min: {float(args.get('f_min', 0.0))} <= f_max: {f_max}")
    power = float(args.get('power', 2.0))
    fb = _mel_fbank(sr, n_fft, float(args.get('f_min', 0.0)), f_max, int(args.get('n_mels', 128)), args.get('mel_scale', 'htk'), args.get('norm'))
    window = args['window_fn'](win, **(args.get('wkwargs') or {})).float() if args.get('window_fn') is not None else torch.hann_window(win)
    normalized = args.get('normalized', False)
    if int(args.get('pad', 0)) > 0:   # torchaudio.functional.spectrogram: zeros on both sides of the signal
        wav = torch.nn.functional.pad(wav, (int(args['pad']), int(args['pad'])), 'constant')
    spec = torch.stft(wav, n_fft, hop, win, window, center=bool(args.get('center', True)), pad_mode=args.get('pad_mode', 'reflect'),
                      normalized=normalized == 'frame_length', onesided=True, return_complex=True)
    if normalized in (True, 'window'):
        spec = spec / window.pow(2.0).sum().sqrt()
    spec = spec.abs()
    if power != 1.0:
        spec = spec.pow(power)
    return spec.transpose(1, 2) @ fb


def featurize(wav, lens_ratio, method, args):
    feats = fbank_batch(wav, args) if method == 'Fbank' else melspec_batch(wav, args)
    feats = feats - feats.mean(1, keepdim=True)  # time mean over ALL frames (featurizer.py:79)
    if lens_ratio is not None:
        T = feats.shape[1]
        mask_lens = torch.round(lens_ratio.to(torch.float32) * T).long().view(-1, 1, 1)  # half-to-even
        keep = torch.arange(T).view(1, T, 1) < mask_lens
        feats = torch.where(keep, feats, torch.zeros_like(feats))
    return feats
