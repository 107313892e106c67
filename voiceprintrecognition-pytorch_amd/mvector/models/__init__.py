"""Backbone registry.  ``build_model(input_size, configs)`` keeps the reference's contract (mvector/models/__init__.py:15-21):
the class named by ``configs.model_conf.model`` (default CAMPPlus) is built with ``model_args``; an unknown name raises
AttributeError, as the reference's ``getattr`` on the module does."""
from mvector.utils.logger import logger
from .campplus import CAMPPlus
from .ecapa_tdnn import EcapaTdnn
from .eres2net import ERes2Net, ERes2NetV2
from .tdnn import TDNN

BACKBONES = {cls.__name__: cls for cls in (CAMPPlus, EcapaTdnn, ERes2Net, ERes2NetV2, TDNN)}


def _outside_the_path(name, what):
    """The reference also exports ``Res2Net``, ``ResNetSE`` (backbones) and ``SpeakerIdentification`` (the training-time
    classifier head) from this module (mvector/models/__init__.py:4-10).  They are outside the embedding path this package
    accelerates (out of scope); the names exist so that configs naming them fail with a clear message
    instead of an AttributeError."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f'{name} ({what}) is outside the MI355X embedding path of this package; the accelerated '
                                  f'backbones are {sorted(BACKBONES)}')
    return type(name, (object,), {'__init__': __init__, '__doc__': _outside_the_path.__doc__})


Res2Net = _outside_the_path('Res2Net', 'configs/res2net.yml backbone')
ResNetSE = _outside_the_path('ResNetSE', 'configs/resnet_se.yml backbone')
SpeakerIdentification = _outside_the_path('SpeakerIdentification', 'training-time classifier head')
_NOT_ACCELERATED = {'Res2Net': Res2Net, 'ResNetSE': ResNetSE, 'SpeakerIdentification': SpeakerIdentification}
__all__ = ['build_model'] + sorted(BACKBONES) + sorted(_NOT_ACCELERATED)


def build_model(input_size, configs):
    conf = configs.model_conf
    name = conf.get('model', 'CAMPPlus')
    if name in _NOT_ACCELERATED:
        _NOT_ACCELERATED[name]()  # raises NotImplementedError with the reason
    if name not in BACKBONES:
        raise AttributeError(f"module 'mvector.models' has no attribute '{name}' (available: {sorted(BACKBONES)})")
    kwargs = conf.get('model_args', {})
    backbone = BACKBONES[name](input_size=input_size, **kwargs)
    logger.info(f'成功创建模型：{name}，参数为：{kwargs}')
    return backbone
