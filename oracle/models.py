"""Oracle backbones: functional fp32 CPU forwards over a plain ``state_dict``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function restates the eval-mode forward of a reference module as a pure
function ``f(state_dict, x)``; parameter names are the reference's state_dict
keys, so one seeded weight set drives the reference (in make_golden.py), this
oracle and the HIP path alike.  Reference lines are cited per function.
"""
import torch
import torch.nn.functional as F


class _P:
    """state_dict view under a dotted prefix."""

    def __init__(self, sd, prefix=''):
        self.sd, self.prefix = sd, prefix

    def sub(self, name):
        return _P(self.sd, f'{self.prefix}{name}.')

    def __getitem__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd


def _bn(p, x, eps=1e-5):
    """nn.BatchNorm{1,2}d in eval mode; affine optional (campplus.py:19-21 'batchnorm_')."""
    w = p['weight'] if p.has('weight') else None
    b = p['bias'] if p.has('bias') else None
    return F.batch_norm(x, p['running_mean'], p['running_var'], w, b, False, 0.0, eps)


# --------------------------------------------------------------------------- ECAPA-TDNN

def _same_reflect_conv(p, x, dilation=1):
    """models/utils.py:39-103: 'same' padding done with F.pad(mode='reflect') then a pad-free conv."""
    w = p['conv.weight']
    k = w.shape[-1]
    pad = dilation * (k - 1) // 2
    if pad > 0:
        x = F.pad(x, (pad, pad), mode='reflect')
    return F.conv1d(x, w, p['conv.bias'] if p.has('conv.bias') else None, dilation=dilation)


def _tdnn_block(p, x, dilation=1):
    """models/utils.py:115-138: BN(ReLU(conv(x))) -- note the order."""
    return _bn(p.sub('norm.norm'), torch.relu(_same_reflect_conv(p.sub('conv'), x, dilation)))


def _res2net(p, x, scale, dilation):
    """ecapa_tdnn.py:39-51."""
    out, prev = [], None
    for i, xi in enumerate(torch.chunk(x, scale, dim=1)):
        if i == 0:
            prev = xi
        elif i == 1:
            prev = _tdnn_block(p.sub(f'blocks.{i - 1}'), xi, dilation)
        else:
            prev = _tdnn_block(p.sub(f'blocks.{i - 1}'), xi + prev, dilation)
        out.append(prev)
    return torch.cat(out, dim=1)


def _se(p, x):
    """ecapa_tdnn.py:71-84 with lengths=None (never passed at inference)."""
    s = x.mean(dim=2, keepdim=True)
    s = torch.relu(_same_reflect_conv(p.sub('conv1'), s))
    s = torch.sigmoid(_same_reflect_conv(p.sub('conv2'), s))
    return s * x


def _se_res2net_block(p, x, scale, dilation):
    """ecapa_tdnn.py:133-143."""
    res = _same_reflect_conv(p.sub('shortcut'), x) if p.has('shortcut.conv.weight') else x
    y = _tdnn_block(p.sub('tdnn1'), x)
    y = _res2net(p.sub('res2net_block'), y, scale, dilation)
    y = _tdnn_block(p.sub('tdnn2'), y)
    return _se(p.sub('se_block'), y) + res


def attentive_stats_pool(p, x, global_context=True, eps=1e-12):
    """pooling.py:86-127 with lengths=None (mask all ones, weights 1/L)."""
    L = x.shape[-1]

    def stats(x, m):
        mean = (m * x).sum(2)
        std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
        return mean, std

    if global_context:
        uni = torch.full((x.shape[0], 1, L), 1.0 / L)
        mean, std = stats(x, uni)
        attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
    else:
        attn = x
    attn = _same_reflect_conv(p.sub('conv'), torch.tanh(_tdnn_block(p.sub('tdnn'), attn)))
    attn = F.softmax(attn, dim=2)
    mean, std = stats(x, attn)
    return torch.cat((mean, std), dim=1)


def ecapa_tdnn(sd, x, dilations=(1, 2, 3, 4, 1), res2net_scale=8, global_context=True, prefix='',
               return_layers=False):
    """EcapaTdnn.forward (ecapa_tdnn.py:253-283), pooling_type='ASP'.  x: [B, T, F] -> [B, embd]."""
    p = _P(sd, prefix)
    x = x.transpose(1, 2)
    outs = []
    x = _tdnn_block(p.sub('blocks.0'), x, dilations[0])
    outs.append(x)
    n_blocks = 1
    while p.has(f'blocks.{n_blocks}.tdnn1.conv.conv.weight'):
        x = _se_res2net_block(p.sub(f'blocks.{n_blocks}'), x, res2net_scale, dilations[n_blocks])
        outs.append(x)
        n_blocks += 1
    x = torch.cat(outs[1:], dim=1)
    mfa = _tdnn_block(p.sub('mfa'), x, dilations[-1])
    pooled = attentive_stats_pool(p.sub('asp'), mfa, global_context)
    y = _bn(p.sub('asp_bn.norm'), pooled)
    emb = _same_reflect_conv(p.sub('fc'), y.unsqueeze(2)).squeeze(-1)
    if return_layers:
        return emb, dict(blocks=outs, mfa=mfa, pooled=pooled)
    return emb


# --------------------------------------------------------------------------- TDNN (x-vector)

def tdnn(sd, x, prefix=''):
    """TDNN.forward (tdnn.py:46-68), pooling_type='ASP'.  Unpadded convs: T shrinks."""
    p = _P(sd, prefix)
    x = x.transpose(2, 1)
    for i, dil in enumerate((1, 2, 3, 1), start=1):
        x = torch.relu(F.conv1d(x, p[f'td_layer{i}.weight'], p[f'td_layer{i}.bias'], dilation=dil))
        x = _bn(p.sub(f'bn{i}'), x)
    x = torch.relu(F.conv1d(x, p['td_layer5.weight'], p['td_layer5.bias']))
    out = _bn(p.sub('bn5'), attentive_stats_pool(p.sub('pooling'), x))
    out = F.linear(out, p['linear.weight'], p['linear.bias'])
    return _bn(p.sub('bn6'), out)


# --------------------------------------------------------------------------- CAM++

def _basic_res_block(p, x, stride):
    """campplus.py:221-254 (BasicResBlock); shortcut conv present iff stride != 1."""
    out = torch.relu(_bn(p.sub('bn1'), F.conv2d(x, p['conv1.weight'], stride=(stride, 1), padding=1)))
    out = _bn(p.sub('bn2'), F.conv2d(out, p['conv2.weight'], padding=1))
    if p.has('shortcut.0.weight'):
        x = _bn(p.sub('shortcut.1'), F.conv2d(x, p['shortcut.0.weight'], stride=(stride, 1)))
    return torch.relu(out + x)


def _fcm(p, x):
    """campplus.py:257-292 (FCM head): [B, F, T] -> [B, 32 * F/8, T]."""
    x = x.unsqueeze(1)
    out = torch.relu(_bn(p.sub('bn1'), F.conv2d(x, p['conv1.weight'], padding=1)))
    for layer in ('layer1', 'layer2'):
        out = _basic_res_block(p.sub(f'{layer}.0'), out, 2)
        out = _basic_res_block(p.sub(f'{layer}.1'), out, 1)
    out = torch.relu(_bn(p.sub('bn2'), F.conv2d(out, p['conv2.weight'], stride=(2, 1), padding=1)))
    return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])


def _seg_pool(x, seg_len=100):
    """campplus.py:101-111: ceil-mode average pooling expanded back to T."""
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    seg = seg.unsqueeze(-1).expand(*seg.shape, seg_len).reshape(*seg.shape[:-1], -1)
    return seg[..., :x.shape[-1]]


def _cam_layer(p, x, dilation):
    """campplus.py:94-99."""
    k = p['linear_local.weight'].shape[-1]
    y = F.conv1d(x, p['linear_local.weight'], None, padding=(k - 1) // 2 * dilation, dilation=dilation)
    ctx = x.mean(-1, keepdim=True) + _seg_pool(x)
    ctx = torch.relu(F.conv1d(ctx, p['linear1.weight'], p['linear1.bias']))
    m = torch.sigmoid(F.conv1d(ctx, p['linear2.weight'], p['linear2.bias']))
    return y * m


def _cam_dense_layer(p, x, dilation):
    """campplus.py:139-150 ('batchnorm-relu' nonlinears: BN first, then ReLU)."""
    h = F.conv1d(torch.relu(_bn(p.sub('nonlinear1.batchnorm'), x)), p['linear1.weight'])
    return _cam_layer(p.sub('cam_layer'), torch.relu(_bn(p.sub('nonlinear2.batchnorm'), h)), dilation)


def campplus(sd, x, prefix='', blocks=((12, 1), (24, 2), (16, 2)), return_layers=False):
    """CAMPPlus.forward (campplus.py:353-357).  x: [B, T, F] -> [B, embd]."""
    p = _P(sd, prefix)
    x = _fcm(p.sub('head'), x.permute(0, 2, 1))
    head = x
    xv = p.sub('xvector')
    x = F.conv1d(x, xv['tdnn.linear.weight'], None, stride=2, padding=2)
    x = torch.relu(_bn(xv.sub('tdnn.nonlinear.batchnorm'), x))
    for bi, (n_layers, dil) in enumerate(blocks, start=1):
        for li in range(1, n_layers + 1):
            x = torch.cat([x, _cam_dense_layer(xv.sub(f'block{bi}.tdnnd{li}'), x, dil)], dim=1)
        tr = xv.sub(f'transit{bi}')
        x = F.conv1d(torch.relu(_bn(tr.sub('nonlinear.batchnorm'), x)), tr['linear.weight'],
                     tr['linear.bias'] if tr.has('linear.bias') else None)
    x = torch.relu(_bn(xv.sub('out_nonlinear.batchnorm'), x))
    stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)  # campplus.py:27-33
    emb = F.conv1d(stats.unsqueeze(-1), xv['dense.linear.weight']).squeeze(-1)
    emb = _bn(xv.sub('dense.nonlinear.batchnorm'), emb)
    if return_layers:
        return emb, dict(head=head, stats=stats)
    return emb


# --------------------------------------------------------------------------- ERes2Net / ERes2NetV2

def _relu20(x):
    """eres2net.py:12-15: the family's "ReLU" is Hardtanh(0, 20)."""
    return x.clamp(0.0, 20.0)


def _aff(p, x, y):
    """eres2net.py:32-52 (AFF): attention from cat(x, y) through conv1x1 -> BN -> SiLU -> conv1x1 -> BN."""
    a = p.sub('local_att')
    h = F.conv2d(torch.cat((x, y), dim=1), a['0.weight'], a['0.bias'])
    h = F.silu(_bn(a.sub('1'), h))
    h = _bn(a.sub('4'), F.conv2d(h, a['3.weight'], a['3.bias']))
    att = 1.0 + torch.tanh(h)
    return x * att + y * (2.0 - att)


def _eres_block(p, x, stride):
    """eres2net.py:55-108 / 111-170 (and the V2 twins :290-335 / :338-380): 1x1 (strided) -> scale x [3x3 on
    sp (+ spx[i] | AFF(sp, spx[i]))] -> 1x1, shortcut conv+BN when the shape changes, ReLU20 everywhere."""
    out = _relu20(_bn(p.sub('bn1'), F.conv2d(x, p['conv1.weight'], stride=stride)))
    width = p['convs.0.weight'].shape[0]
    groups = torch.split(out, width, 1)
    outs, sp = [], None
    for i, g in enumerate(groups):
        if i == 0:
            sp = g
        elif p.has(f'fuse_models.{i - 1}.local_att.0.weight'):
            sp = _aff(p.sub(f'fuse_models.{i - 1}'), sp, g)
        else:
            sp = sp + g
        sp = _relu20(_bn(p.sub(f'bns.{i}'), F.conv2d(sp, p[f'convs.{i}.weight'], padding=1)))
        outs.append(sp)
    out = _bn(p.sub('bn3'), F.conv2d(torch.cat(outs, 1), p['conv3.weight']))
    if p.has('shortcut.0.weight'):
        x = _bn(p.sub('shortcut.1'), F.conv2d(x, p['shortcut.0.weight'], stride=stride))
    return _relu20(out + x)


def _tstp(x):
    """pooling.py:130-148 (TemporalStatsPool): mean and sqrt(unbiased var + 1e-8) over the last axis, flattened [B, C*F]."""
    return torch.cat((x.mean(-1).flatten(1), torch.sqrt(x.var(-1) + 1e-8).flatten(1)), 1)


def _eres_layers(p, x):
    """conv1/bn1/relu stem (plain ReLU, eres2net.py:270) and the four stages; block counts read from the keys."""
    out = torch.relu(_bn(p.sub('bn1'), F.conv2d(x.permute(0, 2, 1).unsqueeze(1), p['conv1.weight'], padding=1)))
    outs = []
    for l in range(1, 5):
        j = 0
        while p.has(f'layer{l}.{j}.conv1.weight'):
            out = _eres_block(p.sub(f'layer{l}.{j}'), out, 2 if (j == 0 and l > 1) else 1)
            j += 1
        outs.append(out)
    return outs


def _eres_head(p, stats):
    emb = F.linear(stats, p['seg_1.weight'], p['seg_1.bias'])
    if p.has('seg_2.weight'):  # two_emb_layer (eres2net.py:283-288)
        emb = F.linear(_bn(p.sub('seg_bn_1'), torch.relu(emb)), p['seg_2.weight'], p['seg_2.bias'])
    return emb


def eres2net(sd, x, prefix=''):
    """ERes2Net.forward, eres2net.py:266-289: bottom-up fusion of all four stage outputs."""
    p = _P(sd, prefix)
    o1, o2, o3, o4 = _eres_layers(p, x)
    f12 = _aff(p.sub('fuse_mode12'), o2, F.conv2d(o1, p['layer1_downsample.weight'], stride=2, padding=1))
    f123 = _aff(p.sub('fuse_mode123'), o3, F.conv2d(f12, p['layer2_downsample.weight'], stride=2, padding=1))
    f1234 = _aff(p.sub('fuse_mode1234'), o4, F.conv2d(f123, p['layer3_downsample.weight'], stride=2, padding=1))
    return _eres_head(p, _tstp(f1234))


def eres2netv2(sd, x, prefix=''):
    """ERes2NetV2.forward, eres2net.py:441-456: only stages 3 and 4 are fused."""
    p = _P(sd, prefix)
    _, _, o3, o4 = _eres_layers(p, x)
    f34 = _aff(p.sub('fuse34'), o4, F.conv2d(o3, p['layer3_ds.weight'], stride=2, padding=1))
    return _eres_head(p, _tstp(f34))


FORWARDS = {'EcapaTdnn': ecapa_tdnn, 'TDNN': tdnn, 'CAMPPlus': campplus, 'ERes2Net': eres2net, 'ERes2NetV2': eres2netv2}
