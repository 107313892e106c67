"""CAM++ with the reference's constructor and state_dict layout (mvector/models/campplus.py:295-357).

Only the parameter TREE is the reference's (so its ``model.pth`` loads unchanged):

    head.{conv1,bn1, layer{1,2}.{0,1}.{conv1,bn1,conv2,bn2[,shortcut.{0,1}]}, conv2,bn2}
    xvector.tdnn.{linear, nonlinear.batchnorm}
    xvector.block{1,2,3}.tdnnd{n}.{nonlinear1.batchnorm, linear1, nonlinear2.batchnorm, cam_layer.{linear_local,linear1,linear2}}
    xvector.transit{1,2,3}.{nonlinear.batchnorm, linear}      xvector.out_nonlinear.batchnorm
    xvector.dense.{linear, nonlinear.batchnorm}

It is built from a nested description by ``_node`` (plain containers, no per-layer classes); the torch forward below walks
that tree functionally and serves CPU tensors / training-mode calls.  Eval-mode CUDA forwards run the native MI355X
pipeline (csrc/campplus.hip), which needs none of this code -- only the state_dict.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from mvector.models._native import NativeBackbone

_DENSE_BLOCKS = ((12, 1), (24, 2), (16, 2))  # (layers, dilation) of block1..3; kernel 3 everywhere (campplus.py:315-317)
_SEG_LEN = 100                                # frames per context segment (campplus.py:101)


class _Node(nn.Module):
    """Container with named children; carries no behaviour (the forward lives in the functions below)."""

    def __init__(self, children):
        super().__init__()
        for name, child in children:
            self.add_module(name, child)


def _node(*children):
    return _Node(children)


def _norm_act(spec, channels):
    """'batchnorm-relu' style description -> container whose child names are the description's tokens
    ('batchnorm_' = BatchNorm without affine parameters, still called 'batchnorm')."""
    parts = []
    for token in spec.split('-'):
        if token == 'batchnorm':
            parts.append(('batchnorm', nn.BatchNorm1d(channels)))
        elif token == 'batchnorm_':
            parts.append(('batchnorm', nn.BatchNorm1d(channels, affine=False)))
        elif token == 'relu':
            parts.append(('relu', nn.ReLU(inplace=True)))
        elif token == 'prelu':
            parts.append(('prelu', nn.PReLU(channels)))
        else:
            raise ValueError('Unexpected module ({}).'.format(token))
    seq = nn.Sequential()
    for name, module in parts:
        seq.add_module(name, module)
    return seq


def _conv1d(cin, cout, k=1, stride=1, dilation=1, bias=False):
    return nn.Conv1d(cin, cout, k, stride=stride, padding=(k - 1) // 2 * dilation, dilation=dilation, bias=bias)


def _res_block(cin, cout, stride):
    kids = [('conv1', nn.Conv2d(cin, cout, 3, stride=(stride, 1), padding=1, bias=False)), ('bn1', nn.BatchNorm2d(cout)),
            ('conv2', nn.Conv2d(cout, cout, 3, stride=1, padding=1, bias=False)), ('bn2', nn.BatchNorm2d(cout))]
    short = nn.Sequential()
    if stride != 1 or cin != cout:
        short = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=(stride, 1), bias=False), nn.BatchNorm2d(cout))
    return _node(*kids, ('shortcut', short))


def _head(maps, feat_dim):
    """FCM front-end (campplus.py:257-292): frequency axis F -> ceil(F/8) with `maps` feature maps."""
    stage = lambda: nn.Sequential(_res_block(maps, maps, 2), _res_block(maps, maps, 1))
    h = _node(('conv1', nn.Conv2d(1, maps, 3, stride=1, padding=1, bias=False)), ('bn1', nn.BatchNorm2d(maps)),
              ('layer1', stage()), ('layer2', stage()),
              ('conv2', nn.Conv2d(maps, maps, 3, stride=(2, 1), padding=1, bias=False)), ('bn2', nn.BatchNorm2d(maps)))
    h.out_channels = maps * math.ceil(feat_dim / 8)
    return h


def _dense_layer(cin, growth, bottleneck, dilation, spec):
    cam = _node(('linear_local', _conv1d(bottleneck, growth, 3, dilation=dilation)),
                ('linear1', nn.Conv1d(bottleneck, bottleneck // 2, 1)), ('linear2', nn.Conv1d(bottleneck // 2, growth, 1)))
    return _node(('nonlinear1', _norm_act(spec, cin)), ('linear1', nn.Conv1d(cin, bottleneck, 1, bias=False)),
                 ('nonlinear2', _norm_act(spec, bottleneck)), ('cam_layer', cam))


# ---- functional forward over the tree ------------------------------------------------------------------------------

def _res_forward(blk, x):
    y = blk.bn2(blk.conv2(F.relu(blk.bn1(blk.conv1(x)))))
    return F.relu(y + blk.shortcut(x))


def _head_forward(h, feats):
    y = F.relu(h.bn1(h.conv1(feats.unsqueeze(1))))
    for stage in (h.layer1, h.layer2):
        for blk in stage:
            y = _res_forward(blk, y)
    y = F.relu(h.bn2(h.conv2(y)))
    return y.flatten(1, 2)  # [B, maps * F/8, T]


def _segment_means(x, seg_len=_SEG_LEN):
    """Mean of every seg_len-frame segment (the last one over its true length), repeated back to T frames."""
    T = x.shape[-1]
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    return seg.repeat_interleave(seg_len, dim=-1)[..., :T]


def _dense_layer_forward(layer, x):
    h = layer.nonlinear2(layer.linear1(layer.nonlinear1(x)))
    cam = layer.cam_layer
    context = h.mean(-1, keepdim=True) + _segment_means(h)
    gate = torch.sigmoid(cam.linear2(F.relu(cam.linear1(context))))
    return cam.linear_local(h) * gate


class CAMPPlus(NativeBackbone, nn.Module):
    _native_kind = 'campp'

    def __init__(self, input_size, embd_dim=512, growth_rate=32, bn_size=4, init_channels=128,
                 config_str='batchnorm-relu', memory_efficient=True):
        super().__init__()
        self.embd_dim = embd_dim
        self._cfg = dict(input_size=input_size, growth_rate=growth_rate, bn_size=bn_size, init_channels=init_channels,
                         config_str=config_str)
        self.memory_efficient = memory_efficient  # activation checkpointing of the reference's training path: not used here
        # MI355X path: 'auto' lets the native handle pick the FCM head's precision from three probe utterances when it is built (fp16 maps, or
        # fp32 maps for a checkpoint whose head amplifies the fp16 rounding); 'f16' / 'f32' pin it.  Not a constructor argument: the reference's
        # signature stays.  In a torch.distributed job every rank calls sync_native_head() once (mvector.parallel.sync_native_choices): rank 0's
        # choice is then pinned on all of them.  forward() itself never issues a collective.
        self.head_precision = 'auto'
        self.head = _head(32, input_size)
        bottleneck = bn_size * growth_rate
        kids = [('tdnn', _node(('linear', _conv1d(self.head.out_channels, init_channels, 5, stride=2)),
                               ('nonlinear', _norm_act(config_str, init_channels))))]
        width = init_channels
        for b, (layers, dilation) in enumerate(_DENSE_BLOCKS, start=1):
            kids.append((f'block{b}', _node(*[(f'tdnnd{i + 1}', _dense_layer(width + i * growth_rate, growth_rate, bottleneck,
                                                                             dilation, config_str)) for i in range(layers)])))
            width += layers * growth_rate
            kids.append((f'transit{b}', _node(('nonlinear', _norm_act(config_str, width)),
                                              ('linear', nn.Conv1d(width, width // 2, 1, bias=False)))))
            width //= 2
        kids.append(('out_nonlinear', _norm_act(config_str, width)))
        kids.append(('dense', _node(('linear', nn.Conv1d(2 * width, embd_dim, 1, bias=False)),
                                    ('nonlinear', _norm_act('batchnorm_', embd_dim)))))
        self.xvector = _node(*kids)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _native_supported(self):
        c = self._cfg
        if c['config_str'] != 'batchnorm-relu':
            return False, f"config_str={c['config_str']!r}"
        if c['growth_rate'] != 32 or c['bn_size'] != 4 or c['init_channels'] % 64 != 0:
            return False, 'growth_rate/bn_size/init_channels other than 32/4/multiple of 64'
        return True, ''

    def _native_cfg(self):
        from mvector import _hip
        cfg = _hip.MvCamppCfg()
        cfg.input_size, cfg.embd_dim = self._cfg['input_size'], self.embd_dim
        cfg.growth_rate, cfg.bn_size, cfg.init_channels = self._cfg['growth_rate'], self._cfg['bn_size'], self._cfg['init_channels']
        cfg.head_precision = {'auto': 0, 'f16': 1, 'f32': 2}[self.head_precision]
        return cfg

    def _native_created(self, handle, build):
        """a fresh handle: say so ONCE per build when the checkpoint is sensitive to the fp16 operands of the x-vector part -- the part that has no
        exact form to fall back to (csrc/campplus.hip::probe_xvector)"""
        rep = handle.campp_head()
        if rep.get('xvector_warning'):
            import warnings
            warnings.warn(f"CAMPPlus: this checkpoint's embedding moves by 1 - cos = {rep['xvector_sensitivity']:.2e} on the handle's probe utterances when "
                          f"the x-vector part runs on fp16 operands (threshold {handle.XVEC_WARN:.1e}); the 1e-4 agreement with the fp32 reference is "
                          f"not guaranteed for it on the MI355X path (mvector.models.CAMPPlus.native_head() reports the figures)", RuntimeWarning, stacklevel=3)
        return handle

    def sync_native_head(self, device=None, group=None, src=0):
        """COLLECTIVE (every rank of ``group`` calls it, once, after the weights are loaded and the module is in eval mode on its device): rank
        ``src``'s FCM head choice becomes every rank's pinned ``head_precision``, so enrol and verify embeddings of a job come out of one numerics.
        The automatic choice is a deterministic function of the checkpoint (fixed probe utterances, deterministic kernels), so ranks on identical
        devices agree anyway; this call makes that a guarantee.  It is NOT issued from ``forward`` (round 4 did that: a rank that skips a forward --
        an empty shard in ``parallel.embed_bucketed``, rank-0-only evaluation as in the reference's trainer.py:376 -- left the others blocked in a
        broadcast).  Returns the pinned value ('f16' / 'f32'), or 'auto' when ``src`` has nothing to decide with (CPU module, no native handle)."""
        import torch.distributed as dist
        codes = {'auto': 0, 'f16': 1, 'f32': 2}
        mine = codes[self.head_precision]
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        if mine == 0 and dev.type == 'cuda' and not self.training and self._native_supported()[0]:
            mine = codes[self._native_handle(dev).campp_head()['head']]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            on_device = dist.get_backend(group) == 'nccl'
            flag = torch.tensor([mine], dtype=torch.int32, device=dev if on_device else 'cpu')
            dist.broadcast(flag, src, group=group)
            mine = int(flag.item())
        want = {v: k for k, v in codes.items()}[mine]
        if want != self.head_precision:
            self.head_precision = want
            self.invalidate_native()   # (a handle built under 'auto' that chose the same head is rebuilt pinned: same kernels, same bits)
        return want

    def native_head(self, range=False):
        """{'head', 'calibration', 'probes'} of the native handle on the current device (None before the first CUDA forward); range=True adds the
        exact head's peak / saturation on the inputs seen so far and waits for the device to read them (a diagnostic)"""
        hs = self.__dict__.get('_native_handles', {})
        for h, _, _ in hs.values():
            return h.campp_head(range=range)
        return None

    def forward(self, x):
        """x: (B, T, F) -> (B, embd_dim)."""
        if self._use_native(x):
            return self._native_forward(x)
        xv = self.xvector
        y = _head_forward(self.head, x.transpose(1, 2))
        y = xv.tdnn.nonlinear(xv.tdnn.linear(y))
        for b in range(1, len(_DENSE_BLOCKS) + 1):
            for layer in getattr(xv, f'block{b}').children():
                y = torch.cat((y, _dense_layer_forward(layer, y)), dim=1)
            transit = getattr(xv, f'transit{b}')
            y = transit.linear(transit.nonlinear(y))
        y = xv.out_nonlinear(y)
        stats = torch.cat((y.mean(-1), y.std(-1, unbiased=True)), dim=-1)   # StatsPool (campplus.py:27-38)
        return xv.dense.nonlinear(xv.dense.linear(stats.unsqueeze(-1)).squeeze(-1))
