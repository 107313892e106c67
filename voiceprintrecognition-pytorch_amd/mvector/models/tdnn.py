"""x-vector TDNN with the reference's constructor and state_dict layout (mvector/models/tdnn.py:9-68)."""
import torch.nn as nn
import torch.nn.functional as F

from mvector.models._native import NativeBackbone
from mvector.models.pooling import (AttentiveStatisticsPooling, SelfAttentivePooling, TemporalAveragePooling,
                                    TemporalStatisticsPooling)


class TDNN(NativeBackbone, nn.Module):
    _native_kind = 'tdnn'
    _LAYERS = ((5, 1), (3, 2), (3, 3), (1, 1), (1, 1))  # (kernel, dilation) of td_layer1..5, all unpadded

    def __init__(self, input_size, channels=512, embd_dim=192, pooling_type="ASP"):
        super().__init__()
        self.embd_dim = embd_dim
        self._cfg = dict(input_size=input_size, channels=channels, pooling_type=pooling_type)
        for i, (k, d) in enumerate(self._LAYERS, start=1):
            setattr(self, f'td_layer{i}', nn.Conv1d(in_channels=input_size if i == 1 else channels, out_channels=channels,
                                                    dilation=d, kernel_size=k, stride=1))
            if i < 5:
                setattr(self, f'bn{i}', nn.BatchNorm1d(channels))
        if pooling_type == "ASP":
            self.pooling, width = AttentiveStatisticsPooling(channels, attention_channels=128), channels * 2
        elif pooling_type == "SAP":
            self.pooling, width = SelfAttentivePooling(channels, 128), channels
        elif pooling_type == "TAP":
            self.pooling, width = TemporalAveragePooling(), channels
        elif pooling_type == "TSP":
            self.pooling, width = TemporalStatisticsPooling(), channels * 2
        else:
            raise Exception(f'没有{pooling_type}池化层！')
        self.bn5 = nn.BatchNorm1d(width)
        self.linear = nn.Linear(width, embd_dim)
        self.bn6 = nn.BatchNorm1d(embd_dim)

    def _native_supported(self):
        if self._cfg['pooling_type'] != 'ASP':
            return False, f"pooling_type={self._cfg['pooling_type']!r}"
        return True, ''

    def _native_cfg(self):
        from mvector import _hip
        cfg = _hip.MvTdnnCfg()
        cfg.input_size, cfg.channels, cfg.embd_dim = self._cfg['input_size'], self._cfg['channels'], self.embd_dim
        return cfg

    def forward(self, x):
        """x: (N, time, freq) -> (N, embd_dim)."""
        if self._use_native(x):
            return self._native_forward(x)
        x = x.transpose(2, 1)
        for i in range(1, 5):
            x = getattr(self, f'bn{i}')(F.relu(getattr(self, f'td_layer{i}')(x)))
        x = F.relu(self.td_layer5(x))
        return self.bn6(self.linear(self.bn5(self.pooling(x))))
