"""EcapaTdnn with the reference's constructor, attributes and state_dict layout
(mvector/models/ecapa_tdnn.py:146-283).  Eval-mode CUDA forwards run the native MI355X pipeline
(csrc/model.hip); the torch graph below serves CPU tensors and training."""
import torch
import torch.nn as nn

from mvector.models._native import NativeBackbone
from mvector.models.pooling import (AttentiveStatisticsPooling, SelfAttentivePooling, TemporalAveragePooling,
                                    TemporalStatisticsPooling)
from mvector.models.utils import BatchNorm1d, Conv1d, TDNNBlock, length_to_mask


class Res2NetBlock(nn.Module):
    """Hierarchical residual over `scale` channel groups; group 0 passes through."""

    def __init__(self, in_channels, out_channels, scale=8, kernel_size=3, dilation=1):
        super().__init__()
        assert in_channels % scale == 0 and out_channels % scale == 0
        self.scale = scale
        self.blocks = nn.ModuleList(TDNNBlock(in_channels // scale, out_channels // scale, kernel_size=kernel_size,
                                              dilation=dilation) for _ in range(scale - 1))

    def forward(self, x):
        groups = torch.chunk(x, self.scale, dim=1)
        outs = [groups[0]]
        for i in range(1, self.scale):
            inp = groups[i] if i == 1 else groups[i] + outs[-1]
            outs.append(self.blocks[i - 1](inp))
        return torch.cat(outs, dim=1)


class SEBlock(nn.Module):
    def __init__(self, in_channels, se_channels, out_channels):
        super().__init__()
        self.conv1 = Conv1d(in_channels=in_channels, out_channels=se_channels, kernel_size=1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv1d(in_channels=se_channels, out_channels=out_channels, kernel_size=1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, lengths=None):
        if lengths is not None:
            L = x.shape[-1]
            mask = length_to_mask(lengths * L, max_len=L, device=x.device).unsqueeze(1)
            s = (x * mask).sum(dim=2, keepdim=True) / mask.sum(dim=2, keepdim=True)
        else:
            s = x.mean(dim=2, keepdim=True)
        return self.sigmoid(self.conv2(self.relu(self.conv1(s)))) * x


class SERes2NetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, res2net_scale=8, se_channels=128, kernel_size=1, dilation=1,
                 activation=nn.ReLU, groups=1):
        super().__init__()
        self.out_channels = out_channels
        self.tdnn1 = TDNNBlock(in_channels, out_channels, kernel_size=1, dilation=1, activation=activation, groups=groups)
        self.res2net_block = Res2NetBlock(out_channels, out_channels, res2net_scale, kernel_size, dilation)
        self.tdnn2 = TDNNBlock(out_channels, out_channels, kernel_size=1, dilation=1, activation=activation, groups=groups)
        self.se_block = SEBlock(out_channels, se_channels, out_channels)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=1)

    def forward(self, x, lengths=None):
        residual = self.shortcut(x) if self.shortcut is not None else x
        y = self.se_block(self.tdnn2(self.res2net_block(self.tdnn1(x))), lengths)
        return y + residual


class EcapaTdnn(NativeBackbone, nn.Module):
    _native_kind = 'ecapa'

    def __init__(self, input_size, embd_dim=192, pooling_type="ASP", activation=nn.ReLU,
                 channels=[512, 512, 512, 512, 1536], kernel_sizes=[5, 3, 3, 3, 1], dilations=[1, 2, 3, 4, 1],
                 attention_channels=128, res2net_scale=8, se_channels=128, global_context=True,
                 groups=[1, 1, 1, 1, 1]):
        super().__init__()
        assert len(channels) == len(kernel_sizes) == len(dilations)
        self.channels = channels
        self.embd_dim = embd_dim
        self._cfg = dict(input_size=input_size, channels=list(channels), kernel_sizes=list(kernel_sizes),
                         dilations=list(dilations), attention_channels=attention_channels,
                         res2net_scale=res2net_scale, se_channels=se_channels, global_context=global_context,
                         groups=list(groups), pooling_type=pooling_type, relu=activation is nn.ReLU)
        self.blocks = nn.ModuleList([TDNNBlock(input_size, channels[0], kernel_sizes[0], dilations[0], activation,
                                               groups[0])])
        for i in range(1, len(channels) - 1):
            self.blocks.append(SERes2NetBlock(channels[i - 1], channels[i], res2net_scale=res2net_scale,
                                              se_channels=se_channels, kernel_size=kernel_sizes[i],
                                              dilation=dilations[i], activation=activation, groups=groups[i]))
        self.mfa = TDNNBlock(channels[-1], channels[-1], kernel_sizes[-1], dilations[-1], activation, groups=groups[-1])
        cat_channels = channels[-1]
        if pooling_type == "ASP":
            self.asp = AttentiveStatisticsPooling(cat_channels, attention_channels=attention_channels,
                                                  global_context=global_context)
            self.asp_bn = BatchNorm1d(input_size=cat_channels * 2)
            pooled = cat_channels * 2
        elif pooling_type == "SAP":
            self.asp = SelfAttentivePooling(cat_channels, 128)
            self.asp_bn = nn.BatchNorm1d(cat_channels)
            pooled = cat_channels
        elif pooling_type == "TAP":
            self.asp = TemporalAveragePooling()
            self.asp_bn = nn.BatchNorm1d(cat_channels)
            pooled = cat_channels
        elif pooling_type == "TSP":
            self.asp = TemporalStatisticsPooling()
            self.asp_bn = nn.BatchNorm1d(cat_channels * 2)
            pooled = cat_channels * 2
        else:
            raise Exception(f'没有{pooling_type}池化层！')
        self.fc = Conv1d(in_channels=pooled, out_channels=self.embd_dim, kernel_size=1)

    # ---- native (HIP) dispatch -------------------------------------------------------------------------
    def _native_supported(self):
        c = self._cfg
        if c['pooling_type'] != 'ASP':
            return False, f"pooling_type={c['pooling_type']!r}"
        if any(g != 1 for g in c['groups']):
            return False, 'grouped convolution'
        if not c['relu']:
            return False, 'a non-ReLU activation'
        if len(c['channels']) != 5:
            return False, 'a block count other than 3 SE-Res2Net blocks'
        return True, ''

    def _native_cfg(self):
        from mvector import _hip
        c = self._cfg
        cfg = _hip.MvEcapaCfg()
        cfg.input_size, cfg.embd_dim = c['input_size'], self.embd_dim
        for i in range(5):
            cfg.channels[i], cfg.kernel_sizes[i], cfg.dilations[i] = c['channels'][i], c['kernel_sizes'][i], c['dilations'][i]
        cfg.attention_channels, cfg.res2net_scale = c['attention_channels'], c['res2net_scale']
        cfg.se_channels, cfg.global_context = c['se_channels'], int(bool(c['global_context']))
        return cfg

    def forward(self, x, lengths=None):
        """x: (batch, time, feature) -> (batch, embd_dim)."""
        if lengths is None and self._use_native(x):
            return self._native_forward(x)
        x = x.transpose(1, 2)
        outs = []
        for layer in self.blocks:
            x = layer(x, lengths=lengths) if isinstance(layer, SERes2NetBlock) else layer(x)
            outs.append(x)
        x = self.mfa(torch.cat(outs[1:], dim=1))
        x = self.asp_bn(self.asp(x)).unsqueeze(2)
        return self.fc(x).squeeze(-1)
