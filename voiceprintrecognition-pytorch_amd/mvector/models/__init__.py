"""Backbone registry.  ``build_model(input_size, configs)`` keeps the reference's contract (mvector/models/__init__.py:15-21):
the class named by ``configs.model_conf.model`` (default CAMPPlus) is built with ``model_args``; an unknown name raises
AttributeError, as the reference's ``getattr`` on the module does."""
from mvector.utils.logger import logger
from .campplus import CAMPPlus
from .ecapa_tdnn import EcapaTdnn
from .eres2net import ERes2Net, ERes2NetV2
from .tdnn import TDNN

BACKBONES = {cls.__name__: cls for cls in (CAMPPlus, EcapaTdnn, ERes2Net, ERes2NetV2, TDNN)}
__all__ = ['build_model'] + sorted(BACKBONES)


def build_model(input_size, configs):
    conf = configs.model_conf
    name = conf.get('model', 'CAMPPlus')
    if name not in BACKBONES:
        raise AttributeError(f"module 'mvector.models' has no attribute '{name}' (available: {sorted(BACKBONES)})")
    kwargs = conf.get('model_args', {})
    backbone = BACKBONES[name](input_size=input_size, **kwargs)
    logger.info(f'成功创建模型：{name}，参数为：{kwargs}')
    return backbone
