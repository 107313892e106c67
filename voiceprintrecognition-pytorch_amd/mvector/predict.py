"""MVectorPredictor with the reference's public API (mvector/predict.py:22-396) on the MI355X-native path.

What changed underneath (not in the signatures): waveforms are padded on the host exactly as the reference
does (predict.py:244-255), moved to the GPU once, featurised there by the fused HIP front-end
(``AudioFeaturizer`` no longer runs on the CPU), embedded by the native backbone, and scored by the HIP cosine
kernel.  With ``use_gpu=False`` everything runs through the torch CPU graphs, as in the reference.
"""
import os
from io import BufferedReader

import numpy as np
import torch
import torch.nn as nn
import yaml

from mvector.data_utils.audio import AudioSegment
from mvector.data_utils.featurizer import AudioFeaturizer
from mvector.models import build_model
from mvector.utils.checkpoint import load_pretrained
from mvector.utils.logger import logger
from mvector.utils.utils import dict_to_object, print_arguments


class MVectorPredictor:
    def __init__(self, configs, threshold=0.6, audio_db_path=None, model_path='models/CAMPPlus_Fbank/best_model/',
                 use_gpu=True):
        """声纹识别预测工具

        :param configs: 配置参数 (YAML path or dict)
        :param threshold: 判断是否为同一个人的阈值
        :param audio_db_path: 声纹库路径
        :param model_path: 导出的预测模型文件夹路径
        :param use_gpu: 是否使用GPU预测
        """
        if use_gpu:
            assert (torch.cuda.is_available()), 'GPU不可用'
            self.device = torch.device("cuda")
        else:
            os.environ['CUDA_VISIBLE_DEVICES'] = '-1'
            self.device = torch.device("cpu")
        self.threshold = threshold
        if isinstance(configs, str):
            with open(configs, 'r', encoding='utf-8') as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
            print_arguments(configs=configs)
        self.configs = dict_to_object(configs)
        self._audio_featurizer = AudioFeaturizer(feature_method=self.configs.preprocess_conf.feature_method,
                                                 use_hf_model=self.configs.preprocess_conf.get('use_hf_model', False),
                                                 method_args=self.configs.preprocess_conf.get('method_args', {}))
        backbone = build_model(input_size=self._audio_featurizer.feature_dim, configs=self.configs)
        self.predictor = nn.Sequential(backbone)
        self.predictor.to(self.device)
        if os.path.isdir(model_path):
            model_path = os.path.join(model_path, 'model.pth')
        assert os.path.exists(model_path), f"{model_path} 模型不存在！"
        self.predictor = load_pretrained(self.predictor, model_path, use_gpu=use_gpu)
        logger.info(f"成功加载模型参数：{model_path}")
        self.predictor.eval()

        self.audio_db_path = audio_db_path
        self._gallery = None
        if audio_db_path is not None:
            self._open_gallery(audio_db_path)

    # ------------------------------------------------------------------ enrolment gallery (host side)
    def _open_gallery(self, audio_db_path):
        """Load ``audio_indexes.bin`` and embed (in eval-batch-sized batches) every audio file the index does not list yet."""
        from mvector.infer_utils.gallery import SpeakerGallery
        self._gallery = SpeakerGallery(audio_db_path)
        todo = self._gallery.unindexed_audio()
        if todo:
            logger.info('正在加载声纹库数据...')
            step = self.configs.dataset_conf.eval_conf.batch_size
            for lo in range(0, len(todo), step):
                chunk = todo[lo:lo + step]
                rows = self.predict_batch([self._load_audio(p).samples for p in chunk])
                for path, row in zip(chunk, rows):
                    self._gallery.add(os.path.basename(os.path.dirname(path)), path, row)
            self._gallery.save()
        users, _ = self._gallery.user_means()
        if users:
            logger.info(f'声纹库数据加载完成，一共有{len(users)}个用户，分别是：{users}')
        self._device_gallery = None
        if self.device.type == 'cuda':
            # GPU-resident enrolment matrix (per-user sums): recognition scores against it without re-uploading anything
            from mvector.infer_utils.gallery import DeviceGallery
            self._device_gallery = DeviceGallery.from_gallery(self._gallery, self.device, self.predictor[0].embd_dim)

    # ------------------------------------------------------------------ scoring
    @staticmethod
    def normalize_features(features):
        return features / np.linalg.norm(features, axis=1, keepdims=True)

    def _cosine(self, queries, gallery):
        """[N, D] x [M, D] -> [N, M]; HIP kernel on the GPU predictor, numpy otherwise."""
        queries = np.asarray(queries, dtype=np.float32)
        gallery = np.asarray(gallery, dtype=np.float32)
        if self.device.type == 'cuda':
            from mvector import _hip
            return _hip.cosine(torch.from_numpy(queries).to(self.device), torch.from_numpy(gallery).to(self.device)).cpu().numpy()
        return self.normalize_features(queries) @ self.normalize_features(gallery).T

    def _best_matches(self, embeddings):
        """[name, score] of the best enrolled user per query row, or [None, None] below the threshold.  ``embeddings``: array
        or (GPU predictor) a device tensor, which is scored against the device-resident gallery through mv_cosine_f32."""
        if getattr(self, '_device_gallery', None) is not None:
            from mvector import _hip
            users = self._device_gallery.users
            q = embeddings if torch.is_tensor(embeddings) else torch.as_tensor(np.atleast_2d(np.asarray(embeddings, dtype=np.float32)))
            scores = _hip.cosine(q.to(self.device).reshape(-1, q.shape[-1]), self._device_gallery.matrix()).cpu().numpy()
        else:
            users, means = self._gallery.user_means()
            scores = self._cosine(np.atleast_2d(np.asarray(embeddings, dtype=np.float32)), means)
        out = []
        for row in scores:
            best = int(np.argmax(row))
            out.append([users[best], round(float(row[best]), 5)] if row[best] >= self.threshold else [None, None])
        return out

    # ------------------------------------------------------------------ audio in
    def _load_audio(self, audio_data, sample_rate=16000):
        """支持文件路径，文件对象，字节，numpy，AudioSegment对象"""
        if isinstance(audio_data, (str, BufferedReader)):
            audio_segment = AudioSegment.from_file(audio_data)
        elif isinstance(audio_data, np.ndarray):
            audio_segment = AudioSegment.from_ndarray(audio_data, sample_rate)
        elif isinstance(audio_data, bytes):
            audio_segment = AudioSegment.from_bytes(audio_data)
        elif isinstance(audio_data, AudioSegment):
            audio_segment = audio_data
        else:
            raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')
        dataset = self.configs.dataset_conf.dataset
        assert audio_segment.duration >= dataset.min_duration, \
            f'音频太短，最小应该为{dataset.min_duration}s，当前音频为{audio_segment.duration}s'
        if audio_segment.sample_rate != dataset.sample_rate:
            audio_segment.resample(dataset.sample_rate)
        if dataset.use_dB_normalization:
            audio_segment.normalize(target_db=dataset.target_dB)
        return audio_segment

    # ------------------------------------------------------------------ embedding extraction (the hot path)
    @torch.no_grad()
    def _embed(self, audio_data, sample_rate=16000):
        """one utterance -> embedding tensor [1, embd_dim] on the predictor's device"""
        input_data = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        wav = torch.tensor(input_data.samples, dtype=torch.float32).unsqueeze(0).to(self.device)
        return self.predictor(self._audio_featurizer(wav)).data

    def predict(self, audio_data, sample_rate=16000):
        """预测一个音频的特征 -> np.ndarray [embd_dim]"""
        return self._embed(audio_data, sample_rate).cpu().numpy()[0]

    def _pcm16_batch(self, audios_data):
        """Raw int16 mono PCM at the configured rate for EVERY item (paths / bytes of 16-bit mono WAVs), else None.
        Such batches travel to the GPU as int16 (half the PCIe bytes) and are scaled / dB-normalised there."""
        from mvector.data_utils.audio import read_pcm16
        dataset = self.configs.dataset_conf.dataset
        out = []
        for a in audios_data:
            if not isinstance(a, (str, bytes)):
                return None
            got = read_pcm16(a)
            if got is None or got[1] != dataset.sample_rate:
                return None
            duration = got[0].shape[0] / float(got[1])
            assert duration >= dataset.min_duration, f'音频太短，最小应该为{dataset.min_duration}s，当前音频为{duration}s'
            out.append(got[0])
        return out

    @torch.no_grad()
    def _predict_batch_pcm16(self, pcms, batch_size):
        """GPU fast path of predict_batch: same padding / length-ratio semantics (predict.py:244-255), int16 upload from
        pinned memory, int16 -> float + dB normalisation + Fbank + CMN + mask + backbone on the device."""
        from mvector import _hip
        dataset = self.configs.dataset_conf.dataset
        lens = [p.shape[0] for p in pcms]
        max_len = max(lens)
        staging = torch.zeros((len(pcms), max_len), dtype=torch.int16)
        if torch.cuda.is_available():
            staging = staging.pin_memory()
        for i, p in enumerate(pcms):
            staging[i, :lens[i]] = torch.from_numpy(np.ascontiguousarray(p))
        pcm = staging.to(self.device, non_blocking=True)
        n = torch.tensor(lens, dtype=torch.int64, device=self.device)
        wav, too_quiet = _hip.wave_prepare(pcm, n, dataset.target_dB if dataset.use_dB_normalization else None)
        ratio = torch.tensor([l / max_len for l in lens], dtype=torch.float32, device=self.device)
        audio_feature = self._audio_featurizer(wav, ratio)
        step = max(int(batch_size), 256)  # 288 GB of HBM: the reference's 32-row chunks only multiply launches
        features = [self.predictor(audio_feature[i:i + step]).data for i in range(0, len(pcms), step)]
        out = torch.cat(features, dim=0).cpu().numpy()
        if bool(too_quiet.any().item()):
            raise ValueError(f'无法将段规范化到{dataset.target_dB}dB，音频增益已经超过max_gain_db (300.0dB)')
        return out

    @torch.no_grad()
    def predict_batch(self, audios_data, sample_rate=16000, batch_size=32):
        """预测一批音频的特征 -> np.ndarray [B, embd_dim] (row order = input order)"""
        self._last_batch_path = 'host'
        if self.device.type == 'cuda' and self.configs.preprocess_conf.feature_method == 'Fbank':
            pcms = self._pcm16_batch(audios_data)
            if pcms is not None:
                self._last_batch_path = 'pcm16'
                return self._predict_batch_pcm16(pcms, batch_size)
        samples = [self._load_audio(audio_data=a, sample_rate=sample_rate).samples for a in audios_data]
        max_len = max(s.shape[0] for s in samples)
        inputs = np.zeros((len(samples), max_len), dtype=np.float32)
        ratios = []
        for i, s in enumerate(samples):
            inputs[i, :s.shape[0]] = s
            ratios.append(s.shape[0] / max_len)
        wav = torch.from_numpy(inputs).to(self.device)
        ratio = torch.tensor(ratios, dtype=torch.float32, device=self.device)
        audio_feature = self._audio_featurizer(wav, ratio)
        features = []
        for i in range(0, len(samples), batch_size):
            features.append(self.predictor(audio_feature[i:i + batch_size]).data)
        return torch.cat(features, dim=0).cpu().numpy()

    def contrast(self, audio_data1, audio_data2):
        """声纹对比 -> 两个音频的相似度"""
        feature1 = self.predict(audio_data1)
        feature2 = self.predict(audio_data2)
        return np.dot(feature1, feature2) / (np.linalg.norm(feature1) * np.linalg.norm(feature2))

    def register(self, audio_data, user_name: str, sample_rate=16000):
        """声纹注册"""
        audio_segment = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        emb = self._embed(audio_segment)
        audio_path = self._gallery.next_audio_path(user_name)
        audio_segment.to_wav_file(audio_path)
        self._gallery.add(user_name, audio_path, emb.cpu().numpy()[0])
        self._gallery.save()
        if getattr(self, '_device_gallery', None) is not None:
            self._device_gallery.add(user_name, emb[0])  # the embedding never leaves the device for the resident matrix
        return True, "注册成功"

    def recognition(self, audio_data, threshold=None, sample_rate=16000):
        """声纹识别 -> [user name or None, score or None]"""
        if threshold:
            self.threshold = threshold
        return self._best_matches(self._embed(audio_data, sample_rate=sample_rate))[0]

    def get_users(self):
        """One entry per enrolled audio, as the reference returns its ``users_name`` list."""
        return list(self._gallery.names)

    def remove_user(self, user_name):
        ok = self._gallery.remove(user_name)
        if ok and getattr(self, '_device_gallery', None) is not None:
            self._device_gallery.remove(user_name)
        return ok

    def speaker_diarization(self, audio_data, sample_rate=16000, speaker_num=None, search_audio_db=False):
        """说话人日志: VAD segmentation and spectral clustering are host post-processing of the reference
        (mvector/infer_utils/speaker_diarization.py) and are not part of the accelerated embedding path."""
        raise NotImplementedError('speaker diarization (VAD + spectral clustering around predict_batch) is outside '
                                  'the MI355X embedding path of this package; use predict_batch for the embeddings')
