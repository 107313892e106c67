"""A/B of the head's first conv: its own launch (fcm_conv1_kernel, map in HBM) + the first block's kernel on that map, against the first
block's kernel making its input rows from the features (mv_fcm_block_c1_f16), at B = 256, T = 298, 80 mel bins.  usage: python tools/bench_fcm_c1.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
lib = _hip.lib()
B, T, F = 256, 298, 80
Fout = 40
g = torch.Generator().manual_seed(0)
nrot = 3
feats = [(torch.randn(B, T, F, generator=g) * 4).cuda() for _ in range(nrot)]
xs = [torch.randn(B, F, T, 32, generator=g).half().cuda() for _ in range(nrot)]
ys = [torch.empty(B, Fout, T, 32, dtype=torch.float16, device='cuda') for _ in range(nrot)]
c1w = torch.randn(32, 3, 3, generator=g) * 0.3
packed = torch.zeros(2 * 64 * 8, dtype=torch.float16)
_hip.check(lib.mv_fcm_c1_pack(c1w.data_ptr(), packed.data_ptr()), lib)
pd, c1b = packed.cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
w1 = (torch.randn(9, 32, 32, generator=g) * 0.08).half().cuda()
w2 = (torch.randn(10, 32, 32, generator=g) * 0.08).half().cuda()
b1 = (torch.randn(32, generator=g) * 0.1).cuda()
b2 = (torch.randn(32, generator=g) * 0.1).cuda()
st = _hip.current_stream(xs[0])
plain = lambda i: lib.mv_fcm_block_f16(xs[i].data_ptr(), F, 2, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1, ys[i].data_ptr(), Fout * T * 32, T * 32, 32, B, T, st)
fused = lambda i: lib.mv_fcm_block_c1_f16(feats[i].data_ptr(), F, pd.data_ptr(), c1b.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), ys[i].data_ptr(),
                                          Fout * T * 32, T * 32, 32, B, T, st)
for rep in range(2):
    for name, call in (('block on the stored map', plain), ('block with the first conv inside', fused)):
        for i in range(3):
            _hip.check(call(i % nrot), lib)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for i in range(n):
            call(i % nrot)
        e1.record()
        torch.cuda.synchronize()
        print(f'{name:34s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us per launch', flush=True)
