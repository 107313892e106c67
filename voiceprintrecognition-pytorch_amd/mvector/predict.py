"""MVectorPredictor with the reference's public API (mvector/predict.py:22-396) on the MI355X-native path.

What changed underneath (not in the signatures): waveforms are padded on the host exactly as the reference
does (predict.py:244-255), moved to the GPU once, featurised there by the fused HIP front-end
(``AudioFeaturizer`` no longer runs on the CPU), embedded by the native backbone, and scored by the HIP cosine
kernel.  With ``use_gpu=False`` everything runs through the torch CPU graphs, as in the reference.
"""
import os
import pickle
import shutil
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

        self.audio_feature = None        # enrolled embeddings [N, D]
        self.audio_feature_mean = None   # per-user mean embeddings [U, D]
        self.users_name = []
        self.users_audio_path = []
        self.users_name_mean = []
        self.audio_db_path = audio_db_path
        if self.audio_db_path is not None:
            self.audio_indexes_path = os.path.join(audio_db_path, "audio_indexes.bin")
            self.__load_audio_db(self.audio_db_path)
        self._speaker_diarize = None

    # ------------------------------------------------------------------ audio-db bookkeeping (host side)
    def __load_audio_indexes(self):
        if not os.path.exists(self.audio_indexes_path):
            return
        with open(self.audio_indexes_path, "rb") as f:
            indexes = pickle.load(f)
        for name, feature, path in zip(indexes["users_name"], indexes["faces_feature"], indexes["users_image_path"]):
            if not os.path.exists(path):
                continue
            self.users_name.append(name)
            self.users_audio_path.append(path)
            self.audio_feature = feature if self.audio_feature is None else np.vstack((self.audio_feature, feature))

    def __write_index(self):
        with open(self.audio_indexes_path, "wb") as f:
            pickle.dump({"users_name": self.users_name, "faces_feature": self.audio_feature,
                         "users_image_path": self.users_audio_path}, f)

    def __append_features(self, features):
        self.audio_feature = features if self.audio_feature is None else np.vstack((self.audio_feature, features))

    def __load_audio_db(self, audio_db_path):
        self.__load_audio_indexes()
        os.makedirs(audio_db_path, exist_ok=True)
        audios_path = []
        for name in os.listdir(audio_db_path):
            audio_dir = os.path.join(audio_db_path, name)
            if not os.path.isdir(audio_dir):
                continue
            for file in os.listdir(audio_dir):
                audios_path.append(os.path.join(audio_dir, file).replace('\\', '/'))
        if len(audios_path) == 0:
            return
        logger.info('正在加载声纹库数据...')
        batch_size = self.configs.dataset_conf.eval_conf.batch_size
        pending = []
        for audio_path in audios_path:
            if audio_path in self.users_audio_path:
                continue
            audio_segment = self._load_audio(audio_path)
            self.users_name.append(os.path.basename(os.path.dirname(audio_path)))
            self.users_audio_path.append(audio_path)
            pending.append(audio_segment.samples)
            if len(pending) == batch_size:
                self.__append_features(self.predict_batch(pending))
                pending = []
        if len(pending) != 0:
            self.__append_features(self.predict_batch(pending))
        assert len(self.audio_feature) == len(self.users_name) == len(self.users_audio_path), '加载的数量对不上！'
        self.__write_index()
        for name in set(self.users_name):
            rows = [i for i, v in enumerate(self.users_name) if v == name]
            feature = self.audio_feature[rows].mean(axis=0)
            self.audio_feature_mean = feature if self.audio_feature_mean is None else \
                np.vstack((self.audio_feature_mean, feature))
            self.users_name_mean.append(name)
        if len(self.audio_feature_mean.shape) == 1:
            self.audio_feature_mean = self.audio_feature_mean[np.newaxis, :]
        logger.info(f'声纹库数据加载完成，一共有{len(self.audio_feature_mean)}个用户，分别是：{self.users_name_mean}')

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
            q = torch.from_numpy(queries).to(self.device)
            g = torch.from_numpy(gallery).to(self.device)
            return _hip.cosine(q, g).cpu().numpy()
        qn = queries / np.linalg.norm(queries, axis=1, keepdims=True)
        gn = gallery / np.linalg.norm(gallery, axis=1, keepdims=True)
        return qn @ gn.T

    def __retrieval(self, np_feature):
        if isinstance(np_feature, list):
            np_feature = np.array(np_feature)
        labels = []
        np_feature = self.normalize_features(np_feature.astype(np.float32))
        similarities = self._cosine(np_feature, self.audio_feature_mean)
        for sim in similarities:
            idx = np.argmax(sim)
            sim = sim[idx]
            if sim >= self.threshold:
                labels.append([self.users_name_mean[idx], round(float(sim), 5)])
            else:
                labels.append([None, None])
        return labels

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
    def predict(self, audio_data, sample_rate=16000):
        """预测一个音频的特征 -> np.ndarray [embd_dim]"""
        input_data = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        wav = torch.tensor(input_data.samples, dtype=torch.float32).unsqueeze(0).to(self.device)
        audio_feature = self._audio_featurizer(wav)
        return self.predictor(audio_feature).data.cpu().numpy()[0]

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
        feature = self.predict(audio_data=audio_segment)
        self.__append_features(feature)
        user_dir = os.path.join(self.audio_db_path, user_name)
        index = len(os.listdir(user_dir)) if os.path.exists(user_dir) else 0
        audio_path = os.path.join(user_dir, f'{index}.wav')
        os.makedirs(user_dir, exist_ok=True)
        audio_segment.to_wav_file(audio_path)
        self.users_audio_path.append(audio_path.replace('\\', '/'))
        self.users_name.append(user_name)
        self.__write_index()
        if user_name in self.users_name_mean:
            index = self.users_name_mean.index(user_name)
            rows = [i for i, v in enumerate(self.users_name) if v == user_name]
            self.audio_feature_mean[index] = self.audio_feature[rows].mean(axis=0)
        else:
            self.users_name_mean.append(user_name)
            self.audio_feature_mean = feature[np.newaxis, :] if self.audio_feature_mean is None else \
                np.vstack((self.audio_feature_mean, feature))
        return True, "注册成功"

    def recognition(self, audio_data, threshold=None, sample_rate=16000):
        """声纹识别 -> [user name or None, score or None]"""
        if threshold:
            self.threshold = threshold
        feature = self.predict(audio_data, sample_rate=sample_rate)
        return self.__retrieval(np_feature=np.array([feature]))[0]

    def get_users(self):
        return self.users_name

    def remove_user(self, user_name):
        if user_name in self.users_name and user_name in self.users_name_mean:
            for index in sorted((i for i, v in enumerate(self.users_name) if v == user_name), reverse=True):
                del self.users_name[index]
                del self.users_audio_path[index]
                self.audio_feature = np.delete(self.audio_feature, index, axis=0)
            self.__write_index()
            shutil.rmtree(os.path.join(self.audio_db_path, user_name))
            index = self.users_name_mean.index(user_name)
            del self.users_name_mean[index]
            self.audio_feature_mean = np.delete(self.audio_feature_mean, index, axis=0)
            return True
        return False

    def speaker_diarization(self, audio_data, sample_rate=16000, speaker_num=None, search_audio_db=False):
        """说话人日志: VAD segmentation and spectral clustering are host post-processing of the reference
        (mvector/infer_utils/speaker_diarization.py) and are not part of the accelerated embedding path."""
        raise NotImplementedError('speaker diarization (VAD + spectral clustering around predict_batch) is outside '
                                  'the MI355X embedding path of this package; use predict_batch for the embeddings')
