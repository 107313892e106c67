"""Error budget of the fp16 CAM++ path on the stress golden (TEST INFRASTRUCTURE: imports the oracle).

The product stores activations as fp16 and feeds fp16 operands to the matrix pipe (fp32 accumulate).  This script restates
CAMPPlus.forward (oracle/models.py:campplus) with a rounding hook at every site where the HIP path rounds, switches groups of
sites on and off, and prints 1 - cos against the reference embedding of tests/golden/campp_stress.npz -- which site owns how
much of the miss, without a GPU.  usage: python tests/budget_campp.py [case]  (writes nothing; the log is committed by hand
under profiles/).
"""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import cos_dist, load_case  # noqa: E402
from oracle.models import _P, _seg_pool  # noqa: E402


def r16(x):
    return x.to(torch.float16).to(torch.float32)


class Sim:
    """on: set of site-group names that round to fp16.  Groups:
    w_fcm, w_xv          folded weights of the FCM convs / of the x-vector convs
    fcm_c1               conv1 output map
    fcm_mid              BasicResBlock intermediate map (conv1 -> BN -> ReLU)
    fcm_out              BasicResBlock output map
    fcm_rows             head.conv2 output (the TDNN input rows)
    xv_store             stored dense-block tensors (tdnn output, every layer's 32 new channels, transit outputs)
    xv_pre               BN+ReLU pre-activation rounded to the fp16 MFMA operand
    xv_h                 bottleneck h (fp16 in LDS)
    split_mid            (modifier) the intermediate map as hi + lo fp16 pair: no rounding beyond 2^-22
    """

    def __init__(self, on):
        self.on = set(on)

    def q(self, group, x):
        return r16(x) if group in self.on else x

    def fold(self, p, conv, bn, group, eps=1e-5):
        w = p[conv + '.weight']
        b = p.sub(bn)
        s = b['weight'] / torch.sqrt(b['running_var'] + eps) if b.has('weight') else 1.0 / torch.sqrt(b['running_var'] + eps)
        t = (b['bias'] if b.has('bias') else 0.0) - b['running_mean'] * s
        return self.q(group, w * s.view(-1, *([1] * (w.dim() - 1)))), t

    def block(self, p, x, stride):
        w1, t1 = self.fold(p, 'conv1', 'bn1', 'w_fcm')
        mid = torch.relu(F.conv2d(x, w1, t1, stride=(stride, 1), padding=1))
        mid = self.q('fcm_mid', mid)
        w2, t2 = self.fold(p, 'conv2', 'bn2', 'w_fcm')
        out = F.conv2d(mid, w2, t2, padding=1)
        if p.has('shortcut.0.weight'):
            ws, ts = self.fold(p, 'shortcut.0', 'shortcut.1', 'w_fcm')
            x = F.conv2d(x, ws, ts, stride=(stride, 1))
        return self.q('fcm_out', torch.relu(out + x))

    def fcm(self, p, x):
        x = x.unsqueeze(1)
        w, t = self.fold(p, 'conv1', 'bn1', 'none')  # first conv: fp32 weights on the vector pipe
        out = self.q('fcm_c1', torch.relu(F.conv2d(x, w, t, padding=1)))
        for layer in ('layer1', 'layer2'):
            out = self.block(p.sub(f'{layer}.0'), out, 2)
            out = self.block(p.sub(f'{layer}.1'), out, 1)
        w, t = self.fold(p, 'conv2', 'bn2', 'w_fcm')
        out = self.q('fcm_rows', torch.relu(F.conv2d(out, w, t, stride=(2, 1), padding=1)))
        return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])

    @staticmethod
    def bn(p, x, eps=1e-5):
        w = p['weight'] if p.has('weight') else None
        b = p['bias'] if p.has('bias') else None
        return F.batch_norm(x, p['running_mean'], p['running_var'], w, b, False, 0.0, eps)

    def dense_layer(self, p, x, dil):
        pre = self.q('xv_pre', torch.relu(self.bn(p.sub('nonlinear1.batchnorm'), x)))
        h = F.conv1d(pre, self.q('w_xv', p['linear1.weight']))
        h = self.q('xv_h', torch.relu(self.bn(p.sub('nonlinear2.batchnorm'), h)))
        c = p.sub('cam_layer')
        k = c['linear_local.weight'].shape[-1]
        y = F.conv1d(h, self.q('w_xv', c['linear_local.weight']), None, padding=(k - 1) // 2 * dil, dilation=dil)
        ctx = h.mean(-1, keepdim=True) + _seg_pool(h)
        ctx = torch.relu(F.conv1d(ctx, c['linear1.weight'], c['linear1.bias']))
        m = torch.sigmoid(F.conv1d(ctx, c['linear2.weight'], c['linear2.bias']))
        return self.q('xv_store', y * m)

    def forward(self, sd, x, blocks=((12, 1), (24, 2), (16, 2))):
        p = _P(sd, '')
        x = self.fcm(p.sub('head'), x.permute(0, 2, 1))
        xv = p.sub('xvector')
        x = F.conv1d(x, self.q('w_xv', xv['tdnn.linear.weight']), None, stride=2, padding=2)
        x = self.q('xv_store', torch.relu(self.bn(xv.sub('tdnn.nonlinear.batchnorm'), x)))
        for bi, (n_layers, dil) in enumerate(blocks, start=1):
            for li in range(1, n_layers + 1):
                x = torch.cat([x, self.dense_layer(xv.sub(f'block{bi}.tdnnd{li}'), x, dil)], dim=1)
            tr = xv.sub(f'transit{bi}')
            pre = self.q('xv_pre', torch.relu(self.bn(tr.sub('nonlinear.batchnorm'), x)))
            x = F.conv1d(pre, self.q('w_xv', tr['linear.weight']), tr['linear.bias'] if tr.has('linear.bias') else None)
            x = self.q('xv_store', x)
        x = torch.relu(self.bn(xv.sub('out_nonlinear.batchnorm'), x))
        stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
        emb = F.conv1d(stats.unsqueeze(-1), xv['dense.linear.weight']).squeeze(-1)
        return self.bn(xv.sub('dense.nonlinear.batchnorm'), emb)


class Sim2(Sim):
    """per-layer switches: names like 'w:layer1.0.conv1', 'm:layer1.0.mid', 'm:layer1.0.out', 'm:c1', 'm:rows'"""
    def __init__(self, on, fine):
        super().__init__(on); self.fine = set(fine); self.cur = ''
    def fold(self, p, conv, bn, group, eps=1e-5):
        w, t = super().fold(p, conv, bn, 'none')
        name = 'w:' + p.prefix[len('head.'):] + conv
        if name in self.fine: w = r16(w)
        return w, t
    def q(self, group, x):
        if group.startswith('fcm_'):
            return x  # handled in block override
        return super().q(group, x)
    def block(self, p, x, stride):
        n = p.prefix[len('head.'):-1]
        w1, t1 = self.fold(p, 'conv1', 'bn1', '')
        mid = torch.relu(F.conv2d(x, w1, t1, stride=(stride, 1), padding=1))
        if 'm:' + n + '.mid' in self.fine: mid = r16(mid)
        w2, t2 = self.fold(p, 'conv2', 'bn2', '')
        out = F.conv2d(mid, w2, t2, padding=1)
        if p.has('shortcut.0.weight'):
            ws, ts = self.fold(p, 'shortcut.0', 'shortcut.1', '')
            x = F.conv2d(x, ws, ts, stride=(stride, 1))
        out = torch.relu(out + x)
        if 'm:' + n + '.out' in self.fine: out = r16(out)
        return out
    def fcm(self, p, x):
        x = x.unsqueeze(1)
        w, t = Sim.fold(self, p, 'conv1', 'bn1', 'none')
        out = torch.relu(F.conv2d(x, w, t, padding=1))
        if 'm:c1' in self.fine: out = r16(out)
        for layer in ('layer1', 'layer2'):
            out = self.block(p.sub(f'{layer}.0'), out, 2)
            out = self.block(p.sub(f'{layer}.1'), out, 1)
        w, t = self.fold(p, 'conv2', 'bn2', '')
        out = torch.relu(F.conv2d(out, w, t, stride=(2, 1), padding=1))
        if 'm:rows' in self.fine: out = r16(out)
        return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])


ALL = ['w_fcm', 'w_xv', 'fcm_c1', 'fcm_mid', 'fcm_out', 'fcm_rows', 'xv_store', 'xv_pre', 'xv_h']


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'campp_stress'
    man, sd, x, emb, _ = load_case(case)
    torch.set_num_threads(16)
    with torch.no_grad():
        def run(on):
            return cos_dist(Sim(on).forward(sd, x), emb).max().item()
        print(f'case {case}: x {tuple(x.shape)}')
        print(f'{"no rounding (restatement check)":48s} {run([]):.3e}')
        for g in ALL:
            print(f'{"only " + g:48s} {run([g]):.3e}')
        print(f'{"all sites (= the round-2 product path)":48s} {run(ALL):.3e}')
        print(f'{"all but fcm_mid (fused BasicResBlock, fp32 mid)":48s} {run([g for g in ALL if g != "fcm_mid"]):.3e}')
        print(f'{"all but fcm_mid, fcm_c1":48s} {run([g for g in ALL if g not in ("fcm_mid", "fcm_c1")]):.3e}')
        print(f'{"all but fcm_* maps":48s} {run([g for g in ALL if not g.startswith("fcm_")]):.3e}')
        print(f'{"all but fcm_* maps and w_fcm":48s} {run([g for g in ALL if not g.startswith("fcm_") and g != "w_fcm"]):.3e}')
        print(f'{"only fcm_* maps":48s} {run([g for g in ALL if g.startswith("fcm_")]):.3e}')


def per_layer():
    """every FCM rounding site on its own (weights of each conv, each stored map): no single owner of the miss"""
    man, sd, x, emb, _ = load_case(sys.argv[1] if len(sys.argv) > 1 else 'campp_stress')
    blocks = ['layer1.0', 'layer1.1', 'layer2.0', 'layer2.1']
    fine_all = (['m:c1', 'm:rows', 'w:conv2'] + [f'w:{b}.{c}' for b in blocks for c in ('conv1', 'conv2')] +
                ['w:layer1.0.shortcut.0', 'w:layer2.0.shortcut.0'] + [f'm:{b}.{s}' for b in blocks for s in ('mid', 'out')])
    xv = ['w_xv', 'xv_store', 'xv_pre', 'xv_h']
    with torch.no_grad():
        run = lambda on, fine: cos_dist(Sim2(on, fine).forward(sd, x), emb).max().item()
        print(f'{"x-vector sites only (FCM exact)":48s} {run(xv, []):.3e}')
        print(f'{"every site":48s} {run(xv, fine_all):.3e}')
        for f in fine_all:
            print(f'{"only " + f:48s} {run([], [f]):.3e}')
        l1 = [f for f in fine_all if 'layer1' in f or f == 'm:c1']
        print(f'{"x-vector + layer2 + head (layer1, c1 exact)":48s} {run(xv, [f for f in fine_all if f not in l1]):.3e}')
        print(f'{"x-vector + layer1 + c1 (layer2, head exact)":48s} {run(xv, l1):.3e}')
        print(f'{"x-vector + FCM weights only":48s} {run(xv, [f for f in fine_all if f.startswith("w:")]):.3e}')
        print(f'{"x-vector + FCM maps only":48s} {run(xv, [f for f in fine_all if f.startswith("m:")]):.3e}')


if __name__ == '__main__':
    main()
    per_layer()
