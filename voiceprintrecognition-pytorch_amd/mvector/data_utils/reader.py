"""Evaluation / feature-extraction dataset with the reference's constructor (mvector/data_utils/reader.py:15-139).

Training mode (augmentation chain, random crops, speed-perturbed labels) is outside the accelerated path and raises.
``return_waveform=True`` is the MI355X-side extension: items are (samples, label) and featurisation happens in one
batched HIP launch per batch (``collate_waveforms`` + ``AudioFeaturizer.forward_varlen``) instead of once per item on
the host.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from mvector.data_utils.audio import AudioSegment
from mvector.data_utils.featurizer import AudioFeaturizer
from mvector.utils.logger import logger


class MVectorDataset(Dataset):
    def __init__(self, data_list_path, audio_featurizer: AudioFeaturizer, max_duration=3, min_duration=0.5, mode='train',
                 sample_rate=16000, aug_conf=None, num_speakers=None, use_dB_normalization=True, target_dB=-20,
                 return_waveform=False):
        super().__init__()
        assert mode in ['train', 'eval', 'extract_feature']
        if mode == 'train':
            raise NotImplementedError('training data pipeline (augmentation, random crops) is outside the MI355X '
                                      'embedding path; use mode="eval" or "extract_feature"')
        self.data_list_path = data_list_path
        self.max_duration = max_duration
        self.min_duration = min_duration
        self.mode = mode
        self._target_sample_rate = sample_rate
        self._use_dB_normalization = use_dB_normalization
        self._target_dB = target_dB
        self.num_speakers = num_speakers
        self.audio_featurizer = audio_featurizer
        self.return_waveform = return_waveform
        self.max_feature_len = self.get_crop_feature_len()
        with open(self.data_list_path, 'r', encoding='utf-8') as f:
            self.lines = [line for line in f.readlines() if line.strip()]
        self.labels = [np.int64(line.strip().split('\t')[1]) for line in self.lines]
        if self.mode == 'eval':
            self.sort_list()

    def load_samples(self, data_path):
        """decode -> resample -> dB normalise -> crop from the start (reader.py:84-99, eval mode)."""
        seg = AudioSegment.from_file(data_path)
        if seg.sample_rate != self._target_sample_rate:
            seg.resample(self._target_sample_rate)
        if self._use_dB_normalization:
            seg.normalize(target_db=self._target_dB)
        if seg.duration > self.max_duration:
            seg.crop(duration=self.max_duration, mode='eval')
        return torch.tensor(seg.samples, dtype=torch.float32)

    def __getitem__(self, idx):
        data_path, spk_id = self.lines[idx].replace('\n', '').split('\t')
        spk_id = torch.tensor(int(spk_id), dtype=torch.int64)
        if data_path.endswith('.npy'):
            feature = np.load(data_path)
            if feature.shape[0] > self.max_feature_len:
                feature = feature[:self.max_feature_len, :]
            return torch.tensor(feature, dtype=torch.float32), spk_id
        if self.mode == 'extract_feature':
            seg = AudioSegment.from_file(data_path)
            if seg.duration < self.min_duration:
                return self.__getitem__(idx + 1 if idx < len(self.lines) - 1 else 0)
        samples = self.load_samples(data_path)
        if self.return_waveform:
            return samples, spk_id
        try:
            feature = self.audio_featurizer(samples)
        except Exception as e:  # same recovery as the reference: skip to the next item
            logger.error(f'[{data_path}]特征提取失败，错误信息：{e}')
            return self.__getitem__(idx + 1 if idx < len(self.lines) - 1 else 0)
        return feature.squeeze(0), spk_id

    def __len__(self):
        return len(self.lines)

    def get_crop_feature_len(self):
        """frames of a max_duration clip (the crop length for pre-extracted .npy features)."""
        samples = torch.zeros((1, int(self.max_duration * self._target_sample_rate)))
        return self.audio_featurizer(samples).size(1)

    def sort_list(self):
        """evaluation lists are walked shortest first so that batches are padded as little as possible"""
        lengths = []
        for line in self.lines:
            data_path = line.split('\t')[0]
            if data_path.endswith('.npy'):
                lengths.append(np.load(data_path).shape[0])
            else:
                lengths.append(AudioSegment.from_file(data_path).duration)
        order = np.argsort(lengths, kind='stable')
        self.lines = [self.lines[i] for i in order]
        self.labels = [self.labels[i] for i in order]
