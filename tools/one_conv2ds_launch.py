"""Three launches of ONE split-fp16 conv layer (for rocprofv3 --pmc passes: counters over a command with hundreds of launches take minutes).
usage: python tools/one_conv2ds_launch.py <name>    name = a layer of tools/bench_conv2d.py's SHAPES (substring), B from MV_BENCH_B (default 16)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
cdll = _hip.lib()
B = int(os.environ.get('MV_BENCH_B', '16'))
SHAPES = {'s1 conv1': (80, 298, 96, 192, 1, 1, False), 's1 3x3': (80, 298, 48, 48, 3, 1, False), 's1 conv3': (80, 298, 192, 192, 1, 1, True),
          's3 3x3': (20, 75, 160, 160, 3, 1, False), 's4 conv3': (10, 38, 1280, 1536, 1, 1, True)}
name = sys.argv[1]
H, W, cin, cout, ks, stride, with_res = SHAPES[name]
st = lambda: _hip.current_stream(torch.empty(1, device='cuda'))
g = torch.Generator().manual_seed(1)
p = ks // 2
Ho, Wo = (H + 2 * p - ks) // stride + 1, (W + 2 * p - ks) // stride + 1
x = torch.randn(B, H, W, cin, generator=g).clamp(0, 20).cuda()
w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
bias = torch.zeros(cout).cuda()
res = torch.randn(B, Ho, Wo, cout, generator=g).cuda() if with_res else None
xs = torch.empty_like(x)
_hip.check(cdll.mv_map_split_f32(x.data_ptr(), xs.data_ptr(), x.numel(), st()), cdll)
rs = None
if with_res:
    rs = torch.empty_like(res)
    _hip.check(cdll.mv_map_split_f32(res.data_ptr(), rs.data_ptr(), res.numel(), st()), cdll)
ys = torch.empty(B, Ho, Wo, cout, device='cuda')
pks = torch.zeros(cdll.mv_conv2ds_packed_elems(cout, cin, ks), device='cuda')
osc = ctypes.c_float(0)
_hip.check(cdll.mv_conv2ds_pack_weight(w.data_ptr(), None, cout, cin, ks, pks.data_ptr(), ctypes.byref(osc), st()), cdll)
e = _hip.MvConv2dsDesc()
e.x, e.ldx, e.w, e.bias, e.oscale, e.y, e.ldy = xs.data_ptr(), cin, pks.data_ptr(), bias.data_ptr(), osc.value, ys.data_ptr(), cout
e.res, e.ldres = (rs.data_ptr() if with_res else None), cout
e.B, e.H, e.W, e.cin16, e.cout16, e.ks, e.stride, e.epi, e.lo, e.hi = B, H, W, cin, cout, ks, stride, 0, 0.0, 20.0
for _ in range(3):
    _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)
torch.cuda.synchronize()
mb = 4.0 * (x.numel() + ys.numel() + (res.numel() if with_res else 0)) / 1e6
print('layer %s B=%d: algorithmic traffic %.1f MB per launch (input + output%s, 4 bytes per channel), weights %.2f MB' % (name, B, mb, ' + residual' if with_res else '', pks.numel() * 4 / 1e6))
