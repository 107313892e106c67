"""Oracle front-end: Kaldi Fbank, MelSpectrogram and the AudioFeaturizer wrapper.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, step by step and in fp32 on the CPU:

* ``torchaudio.compliance.kaldi.fbank`` (torchaudio 2.4.0, not vendored) as
  called from mvector/data_utils/featurizer.py:128 with
  ``method_args = {sample_frequency: 16000, num_mel_bins: 80}``
  (configs/ecapa_tdnn.yml:49-51) -- algorithm in SURVEY.md section 8(c);
* ``torchaudio.transforms.MelSpectrogram(**method_args)`` as built at
  mvector/data_utils/featurizer.py:41-42;
* ``KaldiFbank.forward`` (featurizer.py:119-132: per-utterance loop, transpose,
  stack) and ``AudioFeaturizer.forward`` (featurizer.py:53-91: transpose to
  [B,T,F], time-mean subtraction over ALL frames, optional length mask with
  round-half-to-even).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FBANK_DEFAULTS = dict(
    blackman_coeff=0.42, channel=-1, dither=0.0, energy_floor=1.0, frame_length=25.0,
    frame_shift=10.0, high_freq=0.0, htk_compat=False, low_freq=20.0, min_duration=0.0,
    num_mel_bins=23, preemphasis_coefficient=0.97, raw_energy=True, remove_dc_offset=True,
    round_to_power_of_two=True, sample_frequency=16000.0, snip_edges=True, subtract_mean=False,
    use_energy=False, use_log_fbank=True, use_power=True, vtln_high=-500.0, vtln_low=100.0,
    vtln_warp=1.0, window_type='povey')

MELSPEC_DEFAULTS = dict(
    sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None,
    pad=0, n_mels=128, power=2.0, normalized=False, center=True, pad_mode='reflect',
    onesided=None, norm=None, mel_scale='htk', window_fn=None, wkwargs=None)

EPS_F32 = float(torch.finfo(torch.float32).eps)  # 1.1920929e-07, the Kaldi log floor


def _next_pow2(n):
    return 1 if n == 0 else 2 ** (n - 1).bit_length()


def _mel_scale(freq):
    return 1127.0 * (1.0 + freq / 700.0).log()


def _inverse_mel_scale(mel_freq):
    return 700.0 * ((mel_freq / 1127.0).exp() - 1.0)


def vtln_warp_freq(vtln_low_cutoff, vtln_high_cutoff, low_freq, high_freq, vtln_warp_factor, freq):
    """torchaudio.compliance.kaldi.vtln_warp_freq: the 3-piece linear warp of Kaldi's VTLN (identity outside [low_freq, high_freq])."""
    assert vtln_low_cutoff > low_freq, 'be sure to set the vtln_low option higher than low_freq'
    assert vtln_high_cutoff < high_freq, 'be sure to set the vtln_high option lower than high_freq [or negative]'
    l = vtln_low_cutoff * max(1.0, vtln_warp_factor)
    h = vtln_high_cutoff * min(1.0, vtln_warp_factor)
    scale = 1.0 / vtln_warp_factor
    Fl = scale * l
    Fh = scale * h
    assert l > low_freq and h < high_freq
    scale_left = (Fl - low_freq) / (l - low_freq)
    scale_right = (high_freq - Fh) / (high_freq - h)
    res = torch.empty_like(freq)
    outside_low_high_freq = torch.lt(freq, low_freq) | torch.gt(freq, high_freq)
    before_l = torch.lt(freq, l)
    before_h = torch.lt(freq, h)
    after_h = torch.ge(freq, h)
    res[after_h] = high_freq + scale_right * (freq[after_h] - high_freq)
    res[before_h] = scale * freq[before_h]
    res[before_l] = low_freq + scale_left * (freq[before_l] - low_freq)
    res[outside_low_high_freq] = freq[outside_low_high_freq]
    return res


def kaldi_mel_banks(num_bins, padded, sample_freq, low_freq, high_freq, vtln_low=100.0, vtln_high=-500.0, vtln_warp=1.0, dtype=torch.float32):
    """torchaudio.compliance.kaldi.get_mel_banks: Kaldi triangular filters, triangles in MEL space, no normalisation; vtln_warp != 1 moves the
    filter edges through the VTLN warp (and then the half-open comparisons of the warped branch decide the weights).

    Returns [num_bins, padded//2] (the caller appends the zero Nyquist column).
    """
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    assert num_bins > 3, 'Must have at least 3 mel bins'   # (get_mel_banks' own first line)
    if high_freq <= 0.0:
        high_freq += nyquist
    assert (0.0 <= low_freq < nyquist) and (0.0 < high_freq <= nyquist) and (low_freq < high_freq), \
        f'Bad values in options: low-freq {low_freq} and high-freq {high_freq} vs. nyquist {nyquist}'
    fft_bin_width = sample_freq / padded
    mel_lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    if vtln_high < 0.0:
        vtln_high += nyquist
    assert vtln_warp == 1.0 or ((low_freq < vtln_low < high_freq) and (0.0 < vtln_high < high_freq) and (vtln_low < vtln_high)), \
        f'Bad values in options: vtln-low {vtln_low} and vtln-high {vtln_high}, versus low-freq {low_freq} and high-freq {high_freq}'
    b = torch.arange(num_bins, dtype=dtype).unsqueeze(1)
    left = mel_lo + b * delta
    center = mel_lo + (b + 1.0) * delta
    right = mel_lo + (b + 2.0) * delta
    if vtln_warp != 1.0:
        warp = lambda m: _mel_scale(vtln_warp_freq(vtln_low, vtln_high, low_freq, high_freq, vtln_warp, _inverse_mel_scale(m)))
        left, center, right = warp(left), warp(center), warp(right)
    mel = _mel_scale(fft_bin_width * torch.arange(num_fft_bins, dtype=dtype)).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    if vtln_warp == 1.0:
        return torch.max(torch.zeros(1, dtype=dtype), torch.min(up, down))
    bins = torch.zeros_like(up)
    up_idx = torch.gt(mel, left) & torch.le(mel, center)
    down_idx = torch.gt(mel, center) & torch.lt(mel, right)
    bins[up_idx] = up[up_idx]
    bins[down_idx] = down[down_idx]
    return bins


def povey_window(n):
    return torch.hann_window(n, periodic=False).pow(0.85)


def kaldi_window(window_type, n, blackman_coeff=0.42, dtype=torch.float32):
    """torchaudio.compliance.kaldi._feature_window_function: symmetric (periodic=False) windows."""
    if window_type == 'hanning':
        return torch.hann_window(n, periodic=False, dtype=dtype)
    if window_type == 'hamming':
        return torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46, dtype=dtype)
    if window_type == 'povey':
        return torch.hann_window(n, periodic=False, dtype=dtype).pow(0.85)
    if window_type == 'rectangular':
        return torch.ones(n, dtype=dtype)
    if window_type == 'blackman':
        a = 2 * math.pi / (n - 1)
        i = torch.arange(n, dtype=dtype)
        return blackman_coeff - 0.5 * torch.cos(a * i) + (0.5 - blackman_coeff) * torch.cos(2 * a * i)
    raise Exception('Invalid window type ' + window_type)


def kaldi_strided_frames(w, size, shift, snip_edges):
    """torchaudio.compliance.kaldi._get_strided: [L] -> [m, size].  snip_edges=False: m = (L + shift // 2) // shift frames over the signal with
    its reversed copy attached to both ends (the last pad = size // 2 - shift // 2 samples of it in front; a negative pad trims the front)."""
    L = w.shape[0]
    if snip_edges:
        if L < size:
            return w.new_empty((0, size))
        m = 1 + (L - size) // shift
    else:
        rev = torch.flip(w, [0])
        m = (L + (shift // 2)) // shift
        pad = size // 2 - shift // 2
        if pad > 0:
            w = torch.cat((rev[-pad:], w, rev), dim=0)
        else:
            w = torch.cat((w[-pad:], rev), dim=0)
    return w.contiguous().as_strided((m, size), (shift, 1)).clone()   # (raises when the frames reach beyond the mirrored signal, like torchaudio)


def _kaldi_fbank_impl(waveform, kwargs, dtype):
    """torchaudio.compliance.kaldi.fbank (2.4.0) restated; dtype float32 = the oracle, float64 = the arbiter (same fp32 samples in)."""
    a = dict(FBANK_DEFAULTS)
    unknown = set(kwargs) - set(a)
    if unknown:
        raise TypeError(f'unexpected fbank arguments {sorted(unknown)}')
    a.update(kwargs)
    for k, v in (('dither', 0.0), ('round_to_power_of_two', True)):
        if a[k] != v:
            raise NotImplementedError(f'oracle restates only {k}={v!r}')
    w = torch.as_tensor(waveform, dtype=torch.float32)
    if w.dim() == 2:
        w = w[max(a['channel'], 0)]
    w = w.to(dtype)
    sf = a['sample_frequency']
    shift = int(sf * a['frame_shift'] * 0.001)
    size = int(sf * a['frame_length'] * 0.001)
    padded = _next_pow2(size)
    L = w.shape[0]
    nbins = a['num_mel_bins']
    eps = torch.tensor(EPS_F32, dtype=dtype)
    if L < a['min_duration'] * sf:
        return torch.empty(0, dtype=dtype)
    frames = kaldi_strided_frames(w, size, shift, a['snip_edges'])
    if frames.shape[0] == 0:
        return torch.empty(0, nbins + int(bool(a['use_energy'])), dtype=dtype)

    def log_energy(fr):
        le = torch.max(fr.pow(2).sum(1), eps).log()
        return le if a['energy_floor'] == 0.0 else torch.max(le, torch.tensor(math.log(a['energy_floor']), dtype=dtype))
    if a['remove_dc_offset']:
        frames = frames - frames.mean(dim=1, keepdim=True)
    if a['raw_energy']:
        energy = log_energy(frames)
    pc = a['preemphasis_coefficient']
    assert 0.0 <= pc <= 1.0, '`preemphasis_coefficient` must be between [0,1]'   # (_get_waveform_and_window_properties)
    if pc != 0.0:
        prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)  # replicate-pad on the left
        frames = frames - pc * prev
    frames = frames * kaldi_window(a['window_type'], size, a['blackman_coeff'], dtype).unsqueeze(0)
    if padded != size:
        frames = F.pad(frames, (0, padded - size))
    if not a['raw_energy']:
        energy = log_energy(frames)
    spec = torch.fft.rfft(frames).abs()
    if a['use_power']:
        spec = spec.pow(2.0)
    banks = kaldi_mel_banks(nbins, padded, sf, a['low_freq'], a['high_freq'], a['vtln_low'], a['vtln_high'], a['vtln_warp'], dtype)
    banks = F.pad(banks, (0, 1))   # zero weight on the Nyquist bin
    mel = torch.mm(spec, banks.T)
    if a['use_log_fbank']:
        mel = torch.max(mel, eps).log()
    if a['use_energy']:
        mel = torch.cat((mel, energy.unsqueeze(1)), dim=1) if a['htk_compat'] else torch.cat((energy.unsqueeze(1), mel), dim=1)
    if a['subtract_mean']:
        mel = mel - mel.mean(dim=0, keepdim=True)
    return mel


def kaldi_fbank(waveform, **kwargs):
    """waveform: fp32 tensor [1, L] or [L] -> [m, num_mel_bins] log-mel energies."""
    return _kaldi_fbank_impl(waveform, kwargs, torch.float32)


def kaldi_fbank_f64(waveform, **kwargs):
    """fp64 ARBITER of kaldi_fbank: the same steps on the same fp32 samples with every intermediate (window, filterbank, DC removal,
    pre-emphasis, FFT, power, mel sums, log) in float64.  Neither torchaudio nor the kernel computes this; it says which of two fp32
    evaluations that disagree on a near-floor log energy is the closer one (tests: |HIP - f64| against |oracle32 - f64|)."""
    return _kaldi_fbank_impl(waveform, kwargs, torch.float64)


def audio_featurizer_fbank_f64(waveforms, input_lens_ratio=None, method_args=None):
    """AudioFeaturizer.forward (Fbank) on the fp64 arbiter: [B, L] fp32 -> [B, T, F] fp64, time mean subtracted in fp64, the same mask."""
    feats = torch.stack([kaldi_fbank_f64(w, **dict(method_args or {})) for w in torch.as_tensor(waveforms, dtype=torch.float32)])
    feats = feats - feats.mean(1, keepdim=True)
    if input_lens_ratio is not None:
        ratio = torch.as_tensor(input_lens_ratio, dtype=torch.float32)
        mask_lens = torch.round(ratio * feats.shape[1]).long().unsqueeze(1)
        mask = (torch.arange(feats.shape[1]).repeat(feats.shape[0], 1) < mask_lens).unsqueeze(-1)
        feats = torch.where(mask, feats, torch.zeros_like(feats))
    return feats


def _hz_to_mel(freq, mel_scale='htk'):
    """torchaudio.functional.functional._hz_to_mel"""
    if mel_scale == 'htk':
        return 2595.0 * math.log10(1.0 + (freq / 700.0))
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (freq - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = math.log(6.4) / 27.0
    if freq >= min_log_hz:
        mels = min_log_mel + math.log(freq / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels, mel_scale='htk'):
    """torchaudio.functional.functional._mel_to_hz"""
    if mel_scale == 'htk':
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale='htk'):
    """torchaudio.functional.melscale_fbanks -> [n_freqs, n_mels] (triangles in Hz; norm='slaney': filter j times 2 / (f[j+2] - f[j]))."""
    if norm is not None and norm != 'slaney':
        raise ValueError('norm must be one of None or "slaney"')
    if mel_scale not in ('htk', 'slaney'):
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel(f_min, mel_scale)
    m_max = _hz_to_mel(f_max, mel_scale)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if norm == 'slaney':
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        fb = fb * enorm.unsqueeze(0)
    return fb


def htk_mel_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    return melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate)


def mel_spectrogram(waveforms, **kwargs):
    """waveforms fp32 [B, L] -> [B, n_mels, 1 + L // hop] mel spectrogram (torchaudio.transforms.MelSpectrogram = Spectrogram + MelScale;
    torchaudio.functional.spectrogram for the normalisation modes: "frame_length" is torch.stft(normalized=True), "window" / True divides the
    complex spectrum by sqrt(sum window^2))."""
    a = dict(MELSPEC_DEFAULTS)
    unknown = set(kwargs) - set(a)
    if unknown:
        raise TypeError(f'unexpected MelSpectrogram arguments {sorted(unknown)}')
    a.update(kwargs)
    if a['power'] is None or a['onesided'] not in (None, True):
        raise NotImplementedError('oracle restates MelSpectrogram with a real exponent, one-sided')
    if a['normalized'] not in (False, True, 'window', 'frame_length'):
        raise ValueError(f"Invalid normalized parameter: {a['normalized']}")
    n_fft = a['n_fft']
    win = a['win_length'] if a['win_length'] is not None else n_fft
    hop = a['hop_length'] if a['hop_length'] is not None else win // 2
    sr = a['sample_rate']
    f_max = a['f_max'] if a['f_max'] is not None else float(sr // 2)
    if a['f_min'] > f_max:   # torchaudio.transforms.MelScale.__init__
        raise ValueError(f"Require f_min: {a['f_min']} <= f_max: {f_max}")
    x = torch.as_tensor(waveforms, dtype=torch.float32)
    if a['pad'] > 0:   # torchaudio.functional.spectrogram: `waveform = torch.nn.functional.pad(waveform, (pad, pad), "constant")` in front of the stft
        x = F.pad(x, (a['pad'], a['pad']), 'constant')
    window = a['window_fn'](win, **(a['wkwargs'] or {})) if a['window_fn'] is not None else torch.hann_window(win)
    spec = torch.stft(x, n_fft, hop, win, window, center=a['center'], pad_mode=a['pad_mode'], normalized=a['normalized'] == 'frame_length',
                      onesided=True, return_complex=True)
    if a['normalized'] in (True, 'window'):
        spec = spec / window.pow(2.0).sum().sqrt()
    spec = spec.abs() if a['power'] == 1.0 else spec.abs().pow(a['power'])
    fb = melscale_fbanks(n_fft // 2 + 1, a['f_min'], f_max, a['n_mels'], sr, a['norm'], a['mel_scale'])
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def audio_featurizer(waveforms, input_lens_ratio=None, feature_method='Fbank', method_args=None):
    """AudioFeaturizer.forward: [B, L] (or [L]) fp32 -> [B, T, F] fp32."""
    method_args = dict(method_args or {})
    x = torch.as_tensor(waveforms, dtype=torch.float32)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if feature_method == 'Fbank':
        # per-utterance loop, [m, F] -> [F, m] -> stack (featurizer.py:125-131)
        feats = torch.stack([kaldi_fbank(row.unsqueeze(0), **method_args).transpose(0, 1) for row in x])
    elif feature_method == 'MelSpectrogram':
        feats = mel_spectrogram(x, **method_args)
    else:
        raise Exception(f'oracle has no restatement of feature_method {feature_method}')
    feats = feats.transpose(2, 1)
    feats = feats - feats.mean(1, keepdim=True)
    if input_lens_ratio is not None:
        ratio = torch.as_tensor(input_lens_ratio, dtype=torch.float32)
        mask_lens = torch.round(ratio * feats.shape[1]).long().unsqueeze(1)
        idxs = torch.arange(feats.shape[1]).repeat(feats.shape[0], 1)
        mask = (idxs < mask_lens).unsqueeze(-1)
        feats = torch.where(mask, feats, torch.zeros_like(feats))
    return feats


def feature_dim(feature_method, method_args=None):
    """AudioFeaturizer.feature_dim (featurizer.py:93-111)."""
    method_args = method_args or {}
    if feature_method == 'MelSpectrogram':
        return method_args.get('n_mels', 128)
    if feature_method == 'Fbank':
        return method_args.get('num_mel_bins', 23)
    raise Exception(f'no feature_dim for {feature_method}')


def synth_waveforms(batch, length, seed=1234, amplitude=0.1):
    """BASELINE.md section 3 input recipe: (0.1 * randn).clamp(-1, 1), Generator seed 1234."""
    g = torch.Generator().manual_seed(seed)
    return (amplitude * torch.randn([batch, length], generator=g)).clamp(-1, 1)


def as_numpy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def wave_prepare(pcm, num_samples=None, target_db=None, max_gain_db=300.0):
    """int16 PCM [B, L] -> float32 waveforms as the reference's host path builds them before featurisation
    (mvector/predict.py:185-212): samples / 32768 (AudioSegment int16 decode), optional ``normalize(target_db)`` over the
    true length with gain = 10^((target_db - rms_db) / 20), rms_db = 10 log10(mean x^2); zeros beyond the true length (the
    zero padding of predict_batch, predict.py:249-255).  yeaudio is not installed here: its ``normalize`` is restated from
    the call site and the package's documented behaviour -- parity unpinned against yeaudio itself.
    Returns (wav float32 [B, L], too_quiet bool [B])."""
    import numpy as np
    pcm = np.asarray(pcm)
    B, L = pcm.shape
    out = np.zeros((B, L), dtype=np.float32)
    quiet = np.zeros(B, dtype=bool)
    for b in range(B):
        n = L if num_samples is None else int(num_samples[b])
        x = pcm[b, :n].astype(np.float32) / 32768.0
        if target_db is not None:
            ms = float(np.mean(x.astype(np.float64) ** 2)) if n else 0.0
            gain = target_db - 10.0 * np.log10(ms) if ms > 0 else np.inf
            if gain > max_gain_db:
                quiet[b] = True
            else:
                x = x * np.float32(10.0 ** (gain / 20.0))
        out[b, :n] = x
    return out, quiet
