"""ERes2Net / ERes2NetV2 with the reference's constructors and state_dict layout (mvector/models/eres2net.py:173-456).

The parameter tree (``conv1, bn1, layer{1..4}.{j}.{conv1,bn1,convs.i,bns.i,fuse_models.i.local_att.{0,1,3,4},conv3,bn3,
shortcut.{0,1}}``, ``layer{1,2,3}_downsample`` / ``layer3_ds``, ``fuse_mode{12,123,1234}`` / ``fuse34``, ``seg_1``,
``seg_bn_1``, ``seg_2``) is the reference's, so its ``model.pth`` loads unchanged.  Eval-mode CUDA forwards run on the
native handle (csrc/eres2net.hip: every conv + BatchNorm + activation / residual / fusion is one conv2d_kernel launch on
channel-last fp32 maps); the torch forward below serves CPU tensors and training-mode calls.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from mvector.models._native import NativeBackbone
from mvector.models.pooling import TemporalStatsPool

__all__ = ['ERes2Net', 'ERes2NetV2']


def _relu20(x):
    """The family's ReLU is Hardtanh(0, 20) (eres2net.py:12-15)."""
    return torch.clamp(x, 0.0, 20.0)


class AFF(nn.Module):
    """Attentional feature fusion of two equally shaped maps (eres2net.py:32-52)."""

    def __init__(self, channels=64, r=4):
        super().__init__()
        inter = int(channels // r)
        self.local_att = nn.Sequential(
            nn.Conv2d(channels * 2, inter, kernel_size=1, stride=1, padding=0),
            nn.BatchNorm2d(inter),
            nn.SiLU(inplace=True),
            nn.Conv2d(inter, channels, kernel_size=1, stride=1, padding=0),
            nn.BatchNorm2d(channels),
        )

    def forward(self, x, ds_y):
        gate = torch.tanh(self.local_att(torch.cat((x, ds_y), dim=1)))
        return x * (1.0 + gate) + ds_y * (1.0 - gate)


class _Res2Block(nn.Module):
    """One residual block of the family: 1x1 (strided) -> `scale` chained 3x3 convs over channel groups -> 1x1 + shortcut.
    ``fuse=True`` joins a group with the previous result through an AFF instead of a sum (the *_AFF blocks)."""

    def __init__(self, expansion, in_planes, planes, stride=1, base_width=32, scale=2, fuse=False):
        super().__init__()
        self.expansion, self.stride, self.scale, self.nums = expansion, stride, scale, scale
        self.width = width = int(math.floor(planes * (base_width / 64.0)))
        self.conv1 = nn.Conv2d(in_planes, width * scale, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(width * scale)
        self.convs = nn.ModuleList(nn.Conv2d(width, width, kernel_size=3, padding=1, bias=False) for _ in range(scale))
        self.bns = nn.ModuleList(nn.BatchNorm2d(width) for _ in range(scale))
        if fuse:
            self.fuse_models = nn.ModuleList(AFF(channels=width, r=4) for _ in range(scale - 1))
        self.conv3 = nn.Conv2d(width * scale, planes * expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * expansion)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, expansion * planes, kernel_size=1, stride=stride, bias=False),
                                          nn.BatchNorm2d(expansion * planes))

    def forward(self, x):
        groups = torch.split(_relu20(self.bn1(self.conv1(x))), self.width, 1)
        outs, sp = [], None
        for i, g in enumerate(groups):
            if i == 0:
                sp = g
            elif hasattr(self, 'fuse_models'):
                sp = self.fuse_models[i - 1](sp, g)
            else:
                sp = sp + g
            sp = _relu20(self.bns[i](self.convs[i](sp)))
            outs.append(sp)
        out = self.bn3(self.conv3(torch.cat(outs, 1)))
        return _relu20(out + self.shortcut(x))


# names kept for callers that pass the block classes explicitly, as the reference signature allows
class BasicBlockERes2Net(_Res2Block):
    def __init__(self, expansion, in_planes, planes, stride=1, base_width=32, scale=2):
        super().__init__(expansion, in_planes, planes, stride, base_width, scale, fuse=False)


class BasicBlockERes2Net_diff_AFF(_Res2Block):
    def __init__(self, expansion, in_planes, planes, stride=1, base_width=32, scale=2):
        super().__init__(expansion, in_planes, planes, stride, base_width, scale, fuse=True)


class BasicBlockERes2NetV2(_Res2Block):
    def __init__(self, expansion, in_planes, planes, stride=1, base_width=26, scale=2):
        super().__init__(expansion, in_planes, planes, stride, base_width, scale, fuse=False)


class BasicBlockERes2NetV2_AFF(_Res2Block):
    def __init__(self, expansion, in_planes, planes, stride=1, base_width=26, scale=2):
        super().__init__(expansion, in_planes, planes, stride, base_width, scale, fuse=True)


class _ERes2Base(NativeBackbone, nn.Module):
    _native_kind = 'eres2net'
    _version = 0

    def _build_trunk(self, input_size, block, block_fuse, num_blocks, m_channels, expansion, base_width, scale, embd_dim,
                     two_emb_layer):
        self.in_planes = m_channels
        self.expansion = expansion
        self.embd_dim = embd_dim
        self.stats_dim = int(input_size / 8) * m_channels * 8
        self.two_emb_layer = two_emb_layer
        self._cfg = dict(input_size=input_size, num_blocks=list(num_blocks), m_channels=m_channels, expansion=expansion,
                         base_width=base_width, scale=scale, custom_blocks=not (issubclass(block, _Res2Block) and
                                                                                issubclass(block_fuse, _Res2Block)))
        self.conv1 = nn.Conv2d(1, m_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(m_channels)
        for i, (blk, stride) in enumerate(((block, 1), (block, 2), (block_fuse, 2), (block_fuse, 2))):
            setattr(self, f'layer{i + 1}', self._make_layer(blk, m_channels << i, num_blocks[i], stride, base_width, scale))

    def _build_head(self, embd_dim):
        self.n_stats = 2
        self.pooling = TemporalStatsPool()
        self.seg_1 = nn.Linear(self.stats_dim * self.expansion * self.n_stats, embd_dim)
        if self.two_emb_layer:
            self.seg_bn_1 = nn.BatchNorm1d(embd_dim, affine=False)
            self.seg_2 = nn.Linear(embd_dim, embd_dim)
        else:
            self.seg_bn_1 = nn.Identity()
            self.seg_2 = nn.Identity()

    def _make_layer(self, block, planes, num_blocks, stride, base_width, scale):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.expansion, self.in_planes, planes, s, base_width, scale))
            self.in_planes = planes * self.expansion
        return nn.Sequential(*layers)

    def _native_supported(self):
        c = self._cfg
        if c['custom_blocks']:
            return False, 'custom block classes'
        if c['expansion'] != 2 or c['input_size'] % 8 or c['m_channels'] % 16 or not 2 <= c['scale'] <= 8:
            return False, f'this configuration ({c})'
        if getattr(self, '_mul_channel', 1) != 1:
            return False, 'mul_channel != 1'
        return True, ''

    def _native_cfg(self):
        from mvector import _hip
        c = self._cfg
        cfg = _hip.MvEres2Cfg()
        cfg.version, cfg.input_size, cfg.embd_dim = self._version, c['input_size'], self.embd_dim
        for i in range(4):
            cfg.num_blocks[i] = c['num_blocks'][i]
        cfg.m_channels, cfg.mul_channel, cfg.expansion = c['m_channels'], getattr(self, '_mul_channel', 1), c['expansion']
        cfg.base_width, cfg.scale, cfg.two_emb_layer = c['base_width'], c['scale'], int(bool(self.two_emb_layer))
        return cfg

    def _stages(self, x):
        out = F.relu(self.bn1(self.conv1(x.permute(0, 2, 1).unsqueeze(1))))  # (B,T,F) => (B,1,F,T)
        outs = []
        for i in range(1, 5):
            out = getattr(self, f'layer{i}')(out)
            outs.append(out)
        return outs

    def _head(self, fused):
        embed_a = self.seg_1(self.pooling(fused))
        if self.two_emb_layer:
            return self.seg_2(self.seg_bn_1(F.relu(embed_a)))
        return embed_a


class ERes2Net(_ERes2Base):
    _version = 1

    def __init__(self, input_size, block=BasicBlockERes2Net, block_fuse=BasicBlockERes2Net_diff_AFF, num_blocks=[3, 4, 6, 3],
                 m_channels=32, mul_channel=1, expansion=2, base_width=32, scale=2, embd_dim=192, two_emb_layer=False):
        super().__init__()
        self.feat_dim = input_size
        self._mul_channel = mul_channel
        self._build_trunk(input_size, block, block_fuse, num_blocks, m_channels, expansion, base_width, scale, embd_dim,
                          two_emb_layer)
        m = m_channels * mul_channel
        for i in (1, 2, 3):  # downsampling of the running fusion, then the fusion with the next stage's output
            setattr(self, f'layer{i}_downsample', nn.Conv2d(m << i, m << (i + 1), kernel_size=3, padding=1, stride=2, bias=False))
        self.fuse_mode12 = AFF(channels=m * 4)
        self.fuse_mode123 = AFF(channels=m * 8)
        self.fuse_mode1234 = AFF(channels=m * 16)
        self._build_head(embd_dim)

    def forward(self, x):
        """x: (B, T, F) -> (B, embd_dim)."""
        if self._use_native(x):
            return self._native_forward(x)
        out1, out2, out3, out4 = self._stages(x)
        fused = self.fuse_mode12(out2, self.layer1_downsample(out1))
        fused = self.fuse_mode123(out3, self.layer2_downsample(fused))
        fused = self.fuse_mode1234(out4, self.layer3_downsample(fused))
        return self._head(fused)


class ERes2NetV2(_ERes2Base):
    _version = 2

    def __init__(self, input_size, block=BasicBlockERes2NetV2, block_fuse=BasicBlockERes2NetV2_AFF, num_blocks=[3, 4, 6, 3],
                 m_channels=32, expansion=2, base_width=26, scale=2, embd_dim=192, two_emb_layer=False):
        super().__init__()
        self._build_trunk(input_size, block, block_fuse, num_blocks, m_channels, expansion, base_width, scale, embd_dim,
                          two_emb_layer)
        self.layer3_ds = nn.Conv2d(m_channels * 8, m_channels * 16, kernel_size=3, padding=1, stride=2, bias=False)
        self.fuse34 = AFF(channels=m_channels * 16, r=4)
        self._build_head(embd_dim)

    def forward(self, x):
        """x: (B, T, F) -> (B, embd_dim)."""
        if self._use_native(x):
            return self._native_forward(x)
        _, _, out3, out4 = self._stages(x)
        return self._head(self.fuse34(out4, self.layer3_ds(out3)))
