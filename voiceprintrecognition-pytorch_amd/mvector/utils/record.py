"""Microphone capture with the reference's surface (mvector/utils/record.py:8-36): ``RecordAudio(channels, sample_rate)``
and ``record(record_seconds, save_path) -> np.ndarray``.

Capturing sound is outside the accelerated path; this module exists so that ``infer_recognition.py`` (which imports and
instantiates ``RecordAudio`` at start-up, infer_recognition.py:5,26) runs unchanged.  ``soundcard`` / ``soundfile`` are looked
up when a recording is actually requested: machines without an audio stack can still enrol and identify from files or arrays
through ``MVectorPredictor``.  A different capture backend can be injected (``RecordAudio.backend = callable``), which is what
the tests do."""
import os
import time

import numpy as np


class RecordAudio:
    backend = None  # optional callable(sample_rate, num_frames, channels) -> float array [num_frames, channels]

    def __init__(self, channels=1, sample_rate=16000):
        self.channels = channels
        self.sample_rate = sample_rate
        self.default_mic = None  # opened on the first record() call

    def _capture(self, num_frames):
        if RecordAudio.backend is not None:
            return np.asarray(RecordAudio.backend(self.sample_rate, num_frames, self.channels))
        if self.default_mic is None:
            try:
                import soundcard
            except ImportError as e:
                raise Exception('录音需要安装soundcard：pip install soundcard') from e
            self.default_mic = soundcard.default_microphone()
        return self.default_mic.record(samplerate=self.sample_rate, numframes=num_frames, channels=self.channels)

    def record(self, record_seconds=3, save_path=None):
        """Blocks for ``record_seconds`` and returns the samples (squeezed to [num_frames] for one channel)."""
        print("开始录音......")
        num_frames = int(record_seconds * self.sample_rate)
        started = time.time()
        data = self._capture(num_frames)
        if RecordAudio.backend is None and int(time.time() - started) < record_seconds:
            raise Exception('录音错误，请检查录音设备，或者卸载soundfile，使用命令重新安装：'
                            'pip install git+https://github.com/bastibe/SoundCard.git')
        audio_data = np.squeeze(data)
        print("录音已结束!")
        if save_path is not None:
            os.makedirs(os.path.dirname(save_path) or '.', exist_ok=True)
            try:
                import soundfile
                soundfile.write(save_path, data=data, samplerate=self.sample_rate)
            except ImportError:
                import scipy.io.wavfile as wavfile
                pcm = np.clip(np.asarray(data, dtype=np.float64) * 32768.0, -32768, 32767).astype(np.int16)
                wavfile.write(save_path, self.sample_rate, pcm)
        return audio_data
