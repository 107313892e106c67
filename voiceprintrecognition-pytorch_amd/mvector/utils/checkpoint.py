"""Weights in: ``load_pretrained`` with the reference's contract (mvector/utils/checkpoint.py:11-51):
``model.pth`` is a plain state_dict of ``nn.Sequential(backbone[, classifier])`` (keys prefixed ``0.``),
loaded non-strictly after dropping shape-mismatched entries.  Training checkpoints (optimizer / scaler /
resume) are outside the embedding path and not provided."""
import os

import torch

from mvector.utils.logger import logger


def load_pretrained(model, pretrained_model, use_gpu=True):
    if pretrained_model is None:
        return model
    if os.path.isdir(pretrained_model):
        pretrained_model = os.path.join(pretrained_model, 'model.pth')
    assert os.path.exists(pretrained_model), f"{pretrained_model} 模型不存在！"
    on_gpu = torch.cuda.is_available() and use_gpu
    state = torch.load(pretrained_model, weights_only=False, map_location=None if on_gpu else 'cpu')
    target = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    current = target.state_dict()
    for name in list(state.keys()):
        if name in current and list(current[name].shape) != list(state[name].shape):
            logger.warning(f'{name} not used, shape {list(state[name].shape)} '
                           f'unmatched with {list(current[name].shape)} in model.')
            state.pop(name)
    missing_keys, unexpected_keys = target.load_state_dict(state, strict=False)
    if len(unexpected_keys) > 0:
        logger.warning('Unexpected key(s) in state_dict: {}. '.format(', '.join(f'"{k}"' for k in unexpected_keys)))
    if len(missing_keys) > 0:
        logger.warning('Missing key(s) in state_dict: {}. '.format(', '.join(f'"{k}"' for k in missing_keys)))
    logger.info('成功加载预训练模型：{}'.format(pretrained_model))
    return model
