"""Minimal ``AudioSegment`` for the predictor when ``yeaudio`` (the reference's audio I/O dependency,
requirements.txt:12) is not installed: WAV decode, channel down-mix, resampling, dB normalisation.
Host-side I/O only -- it is not part of the accelerated path."""
import io

import numpy as np

try:  # pragma: no cover
    from yeaudio.audio import AudioSegment  # noqa: F401
except ImportError:
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    class AudioSegment:
        def __init__(self, samples, sample_rate):
            samples = np.asarray(samples)
            if samples.dtype.kind in 'iu':
                info = np.iinfo(samples.dtype)
                if samples.dtype == np.uint8:
                    samples = (samples.astype(np.float32) - 128.0) / 128.0
                else:
                    samples = samples.astype(np.float32) / float(max(abs(info.min), info.max))
            samples = samples.astype(np.float32)
            if samples.ndim == 2:  # [frames, channels] -> mono
                samples = samples.mean(axis=1)
            self._samples = samples
            self._sample_rate = int(sample_rate)

        @classmethod
        def from_file(cls, file):
            sr, data = wavfile.read(file)
            return cls(data, sr)

        @classmethod
        def from_bytes(cls, data):
            return cls.from_file(io.BytesIO(data))

        @classmethod
        def from_ndarray(cls, data, sample_rate=16000):
            return cls(data, sample_rate)

        @property
        def samples(self):
            return self._samples

        @property
        def sample_rate(self):
            return self._sample_rate

        @property
        def num_samples(self):
            return self._samples.shape[0]

        @property
        def duration(self):
            return self._samples.shape[0] / float(self._sample_rate)

        @property
        def rms_db(self):
            mean_square = np.mean(self._samples.astype(np.float64) ** 2)
            return 10 * np.log10(max(mean_square, 1e-20))

        def resample(self, target_sample_rate, filter='kaiser_best'):
            if target_sample_rate == self._sample_rate:
                return
            g = np.gcd(int(target_sample_rate), self._sample_rate)
            self._samples = resample_poly(self._samples, target_sample_rate // g, self._sample_rate // g).astype(np.float32)
            self._sample_rate = int(target_sample_rate)

        def normalize(self, target_db=-20, max_gain_db=300.0):
            gain = target_db - self.rms_db
            if gain > max_gain_db:
                raise ValueError(f'无法将段规范化到{target_db}dB，音频增益{gain}增益已经超过max_gain_db ({max_gain_db}dB)')
            self._samples = (self._samples * 10.0 ** (min(max_gain_db, gain) / 20.0)).astype(np.float32)

        def crop(self, duration, mode='eval'):
            n = int(duration * self._sample_rate)
            if self.num_samples > n:
                start = 0 if mode == 'eval' else np.random.randint(0, self.num_samples - n)
                self._samples = self._samples[start:start + n]

        def to_wav_file(self, filepath, dtype='int16'):
            data = np.clip(self._samples, -1.0, 1.0)
            if dtype == 'int16':
                data = (data * 32767.0).astype(np.int16)
            wavfile.write(filepath, self._sample_rate, data)


def read_pcm16(file):
    """(int16 mono samples, sample rate) of a 16-bit PCM mono WAV given by path / file object / bytes, else None.
    The GPU predictor uploads such audio as int16 and scales / normalises it on the device (mv_wave_prepare_i16)."""
    from scipy.io import wavfile as _wavfile
    try:
        sr, data = _wavfile.read(io.BytesIO(file) if isinstance(file, (bytes, bytearray)) else file)
    except Exception:
        return None
    if data.dtype != np.int16 or data.ndim != 1:
        return None
    return data, int(sr)
