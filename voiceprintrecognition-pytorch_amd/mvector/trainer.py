"""``MVectorTrainer.evaluate`` with the reference's constructor and return values (mvector/trainer.py:38-86,399-482).

Only the evaluation of a trained model is on the MI355X embedding path; ``train`` / ``export`` / ``extract_features``
raise.  What changed underneath ``evaluate`` (not in its contract):

* GPU: utterances travel as zero-padded waveform batches; one variable-length Fbank launch per batch featurises every
  row on its own length (``mv_fbank_forward_varlen`` = per-utterance featurisation + ``collate_fn`` zero padding of the
  reference), the native backbone embeds the batch, and the trials x enrolment score matrix is one HIP cosine launch
  instead of a Python loop over sklearn calls (trainer.py:452-461).
* CPU (``use_gpu=False``): the reference's flow -- per-item featurisation in the dataset, padded feature batches, torch
  graphs.
The metrics (EER, minDCF, threshold) are computed on the host from the same flattened score / label arrays.
"""
import os

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader

from mvector.data_utils.collate_fn import collate_fn, collate_waveforms
from mvector.data_utils.featurizer import AudioFeaturizer
from mvector.data_utils.reader import MVectorDataset
from mvector.metric.metrics import compute_fnr_fpr, compute_eer, compute_dcf
from mvector.models import build_model
from mvector.utils.checkpoint import load_pretrained
from mvector.utils.logger import logger
from mvector.utils.utils import dict_to_object, print_arguments


class MVectorTrainer(object):
    def __init__(self, configs, use_gpu=True, data_augment_configs=None):
        """:param configs: YAML path or dict (reference config files work unchanged)
        :param use_gpu: embed on the MI355X (HIP path) or on the CPU (torch graphs)
        :param data_augment_configs: training-only; ignored by ``evaluate``"""
        if use_gpu:
            assert (torch.cuda.is_available()), 'GPU不可用'
            self.device = torch.device('cuda')
        else:
            self.device = torch.device('cpu')
        self.use_gpu = use_gpu
        if isinstance(configs, str):
            with open(configs, 'r', encoding='utf-8') as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
            print_arguments(configs=configs)
        self.configs = dict_to_object(configs)
        self.model = None
        self.stop_eval = False
        self.audio_featurizer = AudioFeaturizer(feature_method=self.configs.preprocess_conf.feature_method,
                                                use_hf_model=self.configs.preprocess_conf.get('use_hf_model', False),
                                                method_args=self.configs.preprocess_conf.get('method_args', {}))
        # the GPU path batches raw waveforms; that is implemented for audio lists (not pre-extracted .npy features)
        self._waveform_batches = use_gpu and self.configs.preprocess_conf.feature_method == 'Fbank'

    # ---- out of scope -------------------------------------------------------------------------------------------
    def train(self, *args, **kwargs):
        raise NotImplementedError('training is outside the MI355X embedding path (SURVEY.md section 8: out of scope)')

    def export(self, *args, **kwargs):
        raise NotImplementedError('model export is outside the MI355X embedding path; model.pth files of the reference load as is')

    def extract_features(self, *args, **kwargs):
        raise NotImplementedError('offline feature extraction is outside the MI355X embedding path')

    # ---- evaluation ---------------------------------------------------------------------------------------------
    def _loaders(self):
        ds_conf = self.configs.dataset_conf
        args = dict(ds_conf.get('dataset', {}))
        args['max_duration'] = ds_conf.eval_conf.max_duration
        # DataLoader workers as configured (dataLoader.num_workers), also for the GPU path: there the workers only decode and
        # dB-normalise audio (waveform items; featurisation happens on the GPU in the main process), so forked workers never
        # touch the HIP runtime
        loader_args = dict(ds_conf.get('dataLoader', {}))
        lists = [ds_conf.enroll_list, ds_conf.trials_list]
        use_wave = self._waveform_batches and not any(self._has_npy(p) for p in lists)
        out = []
        for path in lists:
            ds = MVectorDataset(data_list_path=path, audio_featurizer=self.audio_featurizer, mode='eval',
                                return_waveform=use_wave, **args)
            out.append(DataLoader(dataset=ds, collate_fn=collate_waveforms if use_wave else collate_fn, shuffle=False,
                                  batch_size=ds_conf.eval_conf.batch_size, **loader_args))
        self.enroll_loader, self.trials_loader = out
        return use_wave

    @staticmethod
    def _has_npy(list_path):
        with open(list_path, 'r', encoding='utf-8') as f:
            return any(line.split('\t')[0].endswith('.npy') for line in f if line.strip())

    def _embed(self, loader, eval_model, use_wave):
        feats, labels = [], []
        with torch.no_grad():
            for data, label, lens in loader:
                if self.stop_eval:
                    break
                data = data.to(self.device)
                if use_wave:
                    data = self.audio_featurizer.forward_varlen(data, lens.to(self.device))
                feats.append(eval_model(data).float())
                labels.append(label)
        return torch.cat(feats), torch.cat(labels).numpy().astype(np.int32)

    def evaluate(self, resume_model=None, save_image_path=None):
        """:return: (eer, min_dcf, threshold) -- floats, as the reference"""
        use_wave = self._loaders()
        if self.model is None:
            backbone = build_model(input_size=self.audio_featurizer.feature_dim, configs=self.configs)
            self.model = torch.nn.Sequential(backbone)
        self.model.to(self.device)
        if resume_model is not None:
            if os.path.isdir(resume_model):
                resume_model = os.path.join(resume_model, 'model.pth')
            assert os.path.exists(resume_model), f'{resume_model} 模型不存在！'
            self.model = load_pretrained(self.model, resume_model, use_gpu=self.use_gpu)
        self.model.eval()
        eval_model = self.model if len(self.model) == 1 else self.model[0]

        enroll_features, enroll_labels = self._embed(self.enroll_loader, eval_model, use_wave)
        trials_features, trials_labels = self._embed(self.trials_loader, eval_model, use_wave)
        if self.stop_eval:
            return -1, -1, -1
        logger.info('开始对比音频特征...')
        if self.use_gpu:
            from mvector import _hip
            scores = _hip.cosine(trials_features, enroll_features).cpu().numpy()
        else:
            a = torch.nn.functional.normalize(trials_features, dim=1)
            b = torch.nn.functional.normalize(enroll_features, dim=1)
            scores = (a @ b.t()).numpy()
        # flattened trial-major, exactly the order of the reference's loop (trainer.py:452-461)
        all_score = scores.astype(np.float32).reshape(-1)
        all_labels = (trials_labels[:, None] == enroll_labels[None, :]).astype(np.int32).reshape(-1)
        fnr, fpr, thresholds = compute_fnr_fpr(all_score, all_labels)
        eer, threshold = compute_eer(fnr, fpr, all_score)
        min_dcf = compute_dcf(fnr, fpr)
        eer, min_dcf, threshold = float(eer), float(min_dcf), float(threshold)
        if save_image_path:
            self._save_curve(save_image_path, thresholds, fnr, fpr, eer, threshold)
        return eer, min_dcf, threshold

    @staticmethod
    def _save_curve(save_image_path, thresholds, fnr, fpr, eer, threshold):
        """the reference plots FNR / FPR over the threshold (trainer.py:470-481); matplotlib is optional here"""
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:
            logger.warning('matplotlib is not installed: evaluation curve not saved')
            return
        os.makedirs(save_image_path, exist_ok=True)
        plt.plot(thresholds, fnr, color='blue', linestyle='-', label='fnr')
        plt.plot(thresholds, fpr, color='red', linestyle='-', label='fpr')
        plt.plot(threshold, eer, 'ro-')
        plt.text(threshold, eer, f'({threshold:.3f}, {eer:.5f})', color='red')
        plt.xlabel('threshold')
        plt.title('fnr and fpr')
        plt.grid(True)
        plt.savefig(os.path.join(save_image_path, 'result.png'))
        plt.close()
