"""AudioFeaturizer with the reference's constructor / forward / feature_dim contract
(mvector/data_utils/featurizer.py:9-111), backed by the fused HIP front-end kernels.

Device rule
-----------
* CUDA (ROCm) tensors: ``mv_fbank_forward`` / ``mv_melspec_forward`` -- waveform batch in HBM ->
  STFT + mel + (log) + time-mean subtraction + length mask in one pass, output on the same device.
  There is no fallback on this path: a missing libmvector_hip.so raises.
* CPU tensors (``use_gpu=False`` predictors, DataLoader worker processes after fork -- they must not touch
  HIP): a batched torch implementation of the same arithmetic (``_cpu_frontend``), as the reference itself
  always runs its featurizer on the CPU.

Unlike the reference, nothing forces the features back to the CPU: the caller keeps the waveforms on the
device and the model consumes the features there.
"""
import torch
from torch import nn

from mvector.data_utils import _cpu_frontend
from mvector.utils.logger import logger


class AudioFeaturizer(nn.Module):
    """音频特征器

    :param feature_method: 所使用的预处理方法 (``Fbank`` / ``MelSpectrogram`` on the accelerated path)
    :param use_hf_model: HuggingFace feature models are outside the accelerated path
    :param method_args: 预处理方法的参数
    """

    def __init__(self, feature_method='MelSpectrogram', use_hf_model=False, method_args={}):
        super().__init__()
        self._method_args = dict(method_args or {})
        self._feature_method = feature_method
        self.use_hf_model = use_hf_model
        if use_hf_model:
            raise NotImplementedError('use_hf_model=True (Wav2Vec2-style HuggingFace front-ends) is outside the '
                                      'MI355X embedding path; use Fbank or MelSpectrogram')
        if feature_method not in ('Fbank', 'MelSpectrogram', 'Spectrogram', 'MFCC'):
            raise Exception(f'预处理方法 {self._feature_method} 不存在!')
        if feature_method in ('Spectrogram', 'MFCC'):
            raise NotImplementedError(f'{feature_method} is not part of the accelerated path (no BASELINE config uses '
                                      f'it); use Fbank or MelSpectrogram')
        _cpu_frontend.validate_args(feature_method, self._method_args)
        self._native = {}  # device index -> native handle (built lazily; never pickled)
        logger.info(f'使用【{feature_method}】提取特征')

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_native'] = {}
        return state

    def _handle(self, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        h = self._native.get(key)
        if h is None:
            from mvector import _hip
            with torch.cuda.device(key):
                if self._feature_method == 'Fbank':
                    h = _hip.Fbank(self._method_args)
                else:
                    h = _hip.MelSpec(self._method_args)
            self._native[key] = h
        return h

    def forward(self, waveforms, input_lens_ratio=None):
        """waveforms: [L] or [B, L] float32 -> [B, T, feature_dim] float32 on the same device."""
        if len(waveforms.shape) == 1:
            waveforms = waveforms.unsqueeze(0)
        if waveforms.dtype != torch.float32:
            waveforms = waveforms.float()
        if waveforms.is_cuda:
            with torch.cuda.device(waveforms.device):
                return self._handle(waveforms.device)(waveforms, input_lens_ratio)
        return _cpu_frontend.featurize(waveforms, input_lens_ratio, self._feature_method, self._method_args)

    def forward_varlen(self, waveforms, num_samples):
        """Zero-padded waveforms [B, L] + true lengths int64 [B] -> [B, T(L), feature_dim]: every row is featurised on its
        own length (own frame count, own time mean) and rows beyond it are zero -- what the reference's evaluation path
        gets from per-utterance featurisation + ``collate_fn`` padding, in one launch (Fbank only on the GPU)."""
        if waveforms.dtype != torch.float32:
            waveforms = waveforms.float()
        if waveforms.is_cuda and self._feature_method == 'Fbank':
            with torch.cuda.device(waveforms.device):
                return self._handle(waveforms.device)(waveforms, None, num_samples)
        T = self.forward(waveforms[:1]).size(1)
        out = torch.zeros((waveforms.size(0), T, self.feature_dim), dtype=torch.float32, device=waveforms.device)
        for i in range(waveforms.size(0)):
            n = int(num_samples[i])
            f = self.forward(waveforms[i, :n])
            out[i, :f.size(1)] = f[0]
        return out

    @property
    def feature_dim(self):
        if self._feature_method == 'MelSpectrogram':
            return self._method_args.get('n_mels', 128)
        elif self._feature_method == 'Spectrogram':
            return self._method_args.get('n_fft', 400) // 2 + 1
        elif self._feature_method == 'MFCC':
            return self._method_args.get('n_mfcc', 40)
        elif self._feature_method == 'Fbank':
            return self._method_args.get('num_mel_bins', 23)
        else:
            raise Exception('没有{}预处理方法'.format(self._feature_method))


class KaldiFbank(nn.Module):
    """The reference's Fbank module (featurizer.py:114-132: ``Kaldi.fbank(waveform, **kwargs)`` per utterance, stacked):
    ``[Batch, Length]`` -> ``[Batch, Feature, Length]`` log-mel energies, no mean subtraction.  ``AudioFeaturizer`` does not go through it
    here (its Fbank, time mean and mask are one launch); the class exists for code that uses it directly.  CUDA tensors run the same HIP
    kernel with the time-mean subtraction switched off, CPU tensors the batched torch restatement -- like ``AudioFeaturizer``."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        _cpu_frontend.validate_args('Fbank', self.kwargs)
        self._native = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_native'] = {}
        return state

    def forward(self, waveforms):
        if waveforms.dim() == 3 and waveforms.size(1) == 1:   # rows given as [1, Length] (the reference unsqueezes 1-D rows to that)
            waveforms = waveforms[:, 0]
        if waveforms.dim() != 2:
            raise ValueError(f'KaldiFbank expects [Batch, Length] waveforms, got {tuple(waveforms.shape)}')
        if waveforms.dtype != torch.float32:
            waveforms = waveforms.float()
        if waveforms.is_cuda:
            key = waveforms.device.index if waveforms.device.index is not None else torch.cuda.current_device()
            with torch.cuda.device(key):
                h = self._native.get(key)
                if h is None:
                    from mvector import _hip
                    h = self._native[key] = _hip.Fbank(self.kwargs, subtract_time_mean=False)
                feats = h(waveforms)
        else:
            feats = _cpu_frontend.fbank_batch(waveforms, self.kwargs)
        return feats.transpose(1, 2)
