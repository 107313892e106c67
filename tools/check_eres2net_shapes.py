import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import mvector.models as M
from mvector import _hip
from oracle import weights, models as om
from helpers import cos_dist
import layer_checks as lc
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from yardstick import binding as ybinding, check_conv2d
lib = _hip.lib()
for cfg in [dict(cin=48, cout=48, ks=3, H=5, W=40, B=1), dict(cin=80, cout=80, ks=3, H=5, W=40, B=1), dict(cin=160, cout=160, ks=3, H=5, W=40, B=1),
            dict(cin=320, cout=320, ks=3, H=5, W=40, B=1), dict(cin=192, cout=192, ks=1, H=5, W=40, B=1), dict(cin=1280, cout=1536, ks=1, H=3, W=20, B=1, with_res=True),
            dict(cin=320, cout=48, ks=1, x2_mode=2, epi=1, H=5, W=40, B=1), dict(cin=48, cout=160, ks=1, epi=2, H=5, W=40, B=1),
            dict(cin=768, cout=1536, ks=3, stride=2, H=6, W=20, B=1, lo=-3e38, hi=3e38)]:
    try:
        print('layer', cfg, check_conv2d.conv2d_case(ybinding.load(), 'cuda', **cfg), flush=True)
    except AssertionError as e:
        print('layer', cfg, 'FAIL', e, flush=True)
torch.set_num_threads(32)
for kw in [dict(m_channels=96, scale=2, num_blocks=[1, 1, 1, 1]), dict(m_channels=32, scale=4, num_blocks=[1, 1, 1, 1]),
           dict(m_channels=96, scale=4, num_blocks=[1, 1, 1, 1]), dict(m_channels=64, scale=2, num_blocks=[1, 1, 1, 1]),
           dict(m_channels=96, scale=4, num_blocks=[3, 4, 6, 3])]:
    m = M.ERes2NetV2(input_size=80, base_width=26, embd_dim=192, **kw)
    sd = weights.make_state_dict(weights.shapes_of(m.state_dict()), 2)
    m.load_state_dict(sd); m.eval()
    x = torch.randn(1, 61, 80, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = om.eres2netv2(sd, x)
    m.to('cuda')
    emb = m(x.cuda()).cpu()
    print('model', kw, '1-cos', cos_dist(emb, ref).max().item(), flush=True)
