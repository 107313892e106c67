"""CAM++ with the reference's constructor and state_dict layout (mvector/models/campplus.py:295-357):
``head`` (FCM 2-D residual front-end) + ``xvector`` (strided TDNN, three CAM dense-TDNN blocks with transit
layers, statistics pooling, dense embedding layer).  Eval-mode CUDA forwards run the native MI355X pipeline
(csrc/campplus.hip); the torch graph below serves CPU tensors and training."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
import torch.utils.checkpoint as cp
from torch import nn

from mvector.models._native import NativeBackbone


def get_nonlinear(config_str, channels):
    """'batchnorm-relu' style spec -> nn.Sequential whose child names are the spec tokens."""
    seq = nn.Sequential()
    for name in config_str.split('-'):
        if name == 'relu':
            seq.add_module('relu', nn.ReLU(inplace=True))
        elif name == 'prelu':
            seq.add_module('prelu', nn.PReLU(channels))
        elif name == 'batchnorm':
            seq.add_module('batchnorm', nn.BatchNorm1d(channels))
        elif name == 'batchnorm_':
            seq.add_module('batchnorm', nn.BatchNorm1d(channels, affine=False))
        else:
            raise ValueError('Unexpected module ({}).'.format(name))
    return seq


def statistics_pooling(x, dim=-1, keepdim=False, unbiased=True, eps=1e-2):
    stats = torch.cat([x.mean(dim=dim), x.std(dim=dim, unbiased=unbiased)], dim=-1)
    return stats.unsqueeze(dim=dim) if keepdim else stats


class StatsPool(nn.Module):
    def forward(self, x):
        return statistics_pooling(x)


class TDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False,
                 config_str='batchnorm-relu'):
        super().__init__()
        if padding < 0:
            assert kernel_size % 2 == 1, f'Expect equal paddings, but got even kernel size ({kernel_size})'
            padding = (kernel_size - 1) // 2 * dilation
        self.linear = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                dilation=dilation, bias=bias)
        self.nonlinear = get_nonlinear(config_str, out_channels)

    def forward(self, x):
        return self.nonlinear(self.linear(x))


class CAMLayer(nn.Module):
    """Local conv masked by a context gate computed from global + 100-frame segment means."""

    def __init__(self, bn_channels, out_channels, kernel_size, stride, padding, dilation, bias, reduction=2):
        super().__init__()
        self.linear_local = nn.Conv1d(bn_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                      dilation=dilation, bias=bias)
        self.linear1 = nn.Conv1d(bn_channels, bn_channels // reduction, 1)
        self.relu = nn.ReLU(inplace=True)
        self.linear2 = nn.Conv1d(bn_channels // reduction, out_channels, 1)
        self.sigmoid = nn.Sigmoid()

    def seg_pooling(self, x, seg_len=100, stype='avg'):
        pool = {'avg': F.avg_pool1d, 'max': F.max_pool1d}.get(stype)
        if pool is None:
            raise ValueError('Wrong segment pooling type.')
        seg = pool(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
        seg = seg.unsqueeze(-1).expand(*seg.shape, seg_len).reshape(*seg.shape[:-1], -1)
        return seg[..., :x.shape[-1]]

    def forward(self, x):
        context = x.mean(-1, keepdim=True) + self.seg_pooling(x)
        gate = self.sigmoid(self.linear2(self.relu(self.linear1(context))))
        return self.linear_local(x) * gate


class CAMDenseTDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1, bias=False,
                 config_str='batchnorm-relu', memory_efficient=False):
        super().__init__()
        assert kernel_size % 2 == 1, f'Expect equal paddings, but got even kernel size ({kernel_size})'
        self.memory_efficient = memory_efficient
        self.nonlinear1 = get_nonlinear(config_str, in_channels)
        self.linear1 = nn.Conv1d(in_channels, bn_channels, 1, bias=False)
        self.nonlinear2 = get_nonlinear(config_str, bn_channels)
        self.cam_layer = CAMLayer(bn_channels, out_channels, kernel_size, stride=stride,
                                  padding=(kernel_size - 1) // 2 * dilation, dilation=dilation, bias=bias)

    def bn_function(self, x):
        return self.linear1(self.nonlinear1(x))

    def forward(self, x):
        if self.training and self.memory_efficient:
            x = cp.checkpoint(self.bn_function, x, use_reentrant=False)
        else:
            x = self.bn_function(x)
        return self.cam_layer(self.nonlinear2(x))


class CAMDenseTDNNBlock(nn.ModuleList):
    def __init__(self, num_layers, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1,
                 bias=False, config_str='batchnorm-relu', memory_efficient=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module('tdnnd%d' % (i + 1),
                            CAMDenseTDNNLayer(in_channels=in_channels + i * out_channels, out_channels=out_channels,
                                              bn_channels=bn_channels, kernel_size=kernel_size, stride=stride,
                                              dilation=dilation, bias=bias, config_str=config_str,
                                              memory_efficient=memory_efficient))

    def forward(self, x):
        for layer in self:
            x = torch.cat([x, layer(x)], dim=1)
        return x


class TransitLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, config_str='batchnorm-relu'):
        super().__init__()
        self.nonlinear = get_nonlinear(config_str, in_channels)
        self.linear = nn.Conv1d(in_channels, out_channels, 1, bias=bias)

    def forward(self, x):
        return self.linear(self.nonlinear(x))


class DenseLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=False, config_str='batchnorm-relu'):
        super().__init__()
        self.linear = nn.Conv1d(in_channels, out_channels, 1, bias=bias)
        self.nonlinear = get_nonlinear(config_str, out_channels)

    def forward(self, x):
        if len(x.shape) == 2:
            x = self.linear(x.unsqueeze(dim=-1)).squeeze(dim=-1)
        else:
            x = self.linear(x)
        return self.nonlinear(x)


class BasicResBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=(stride, 1), padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=(stride, 1), bias=False),
                nn.BatchNorm2d(self.expansion * planes))

    def forward(self, x):
        out = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(out + self.shortcut(x))


class FCM(nn.Module):
    """2-D convolutional front-end: frequency axis 80 -> 10 with 32 maps, flattened to 320 channels."""

    def __init__(self, block=BasicResBlock, num_blocks=[2, 2], m_channels=32, feat_dim=80):
        super().__init__()
        self.in_planes = m_channels
        self.conv1 = nn.Conv2d(1, m_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], stride=2)
        self.layer2 = self._make_layer(block, m_channels, num_blocks[0], stride=2)
        self.conv2 = nn.Conv2d(m_channels, m_channels, kernel_size=3, stride=(2, 1), padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(m_channels)
        self.out_channels = m_channels * (math.ceil(feat_dim / 8))

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x.unsqueeze(1))))
        out = self.layer2(self.layer1(out))
        out = F.relu(self.bn2(self.conv2(out)))
        return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])


class CAMPPlus(NativeBackbone, nn.Module):
    _native_kind = 'campp'

    def __init__(self, input_size, embd_dim=512, growth_rate=32, bn_size=4, init_channels=128,
                 config_str='batchnorm-relu', memory_efficient=True):
        super().__init__()
        self.head = FCM(feat_dim=input_size)
        channels = self.head.out_channels
        self.embd_dim = embd_dim
        self._cfg = dict(input_size=input_size, growth_rate=growth_rate, bn_size=bn_size, init_channels=init_channels,
                         config_str=config_str)
        self.xvector = nn.Sequential(OrderedDict([
            ('tdnn', TDNNLayer(channels, init_channels, 5, stride=2, dilation=1, padding=-1, config_str=config_str))]))
        channels = init_channels
        for i, (num_layers, kernel_size, dilation) in enumerate(zip((12, 24, 16), (3, 3, 3), (1, 2, 2))):
            self.xvector.add_module('block%d' % (i + 1),
                                    CAMDenseTDNNBlock(num_layers=num_layers, in_channels=channels,
                                                      out_channels=growth_rate, bn_channels=bn_size * growth_rate,
                                                      kernel_size=kernel_size, dilation=dilation,
                                                      config_str=config_str, memory_efficient=memory_efficient))
            channels = channels + num_layers * growth_rate
            self.xvector.add_module('transit%d' % (i + 1),
                                    TransitLayer(channels, channels // 2, bias=False, config_str=config_str))
            channels //= 2
        self.xvector.add_module('out_nonlinear', get_nonlinear(config_str, channels))
        self.xvector.add_module('stats', StatsPool())
        self.xvector.add_module('dense', DenseLayer(channels * 2, embd_dim, config_str='batchnorm_'))
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _native_supported(self):
        if self._cfg['config_str'] != 'batchnorm-relu':
            return False, f"config_str={self._cfg['config_str']!r}"
        if self._cfg['growth_rate'] != 32 or self._cfg['bn_size'] != 4 or self._cfg['init_channels'] % 64 != 0:
            return False, 'growth_rate/bn_size/init_channels other than 32/4/multiple of 64'
        return True, ''

    def _native_cfg(self):
        from mvector import _hip
        cfg = _hip.MvCamppCfg()
        cfg.input_size, cfg.embd_dim = self._cfg['input_size'], self.embd_dim
        cfg.growth_rate, cfg.bn_size, cfg.init_channels = self._cfg['growth_rate'], self._cfg['bn_size'], self._cfg['init_channels']
        return cfg

    def forward(self, x):
        """x: (B, T, F) -> (B, embd_dim)."""
        if self._use_native(x):
            return self._native_forward(x)
        return self.xvector(self.head(x.permute(0, 2, 1)))
