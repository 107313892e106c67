"""Micro-benchmark of mv_asp_pool_f16 at the bench shape (HIP events): python tools/bench_asp.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
B, T, C, A = 256, 298, 3072, 128
h = torch.tanh(torch.randn(B, T, A, device='cuda')).half()
x = (torch.randn(B, T, C, device='cuda') * 1.5 + 0.3).half()
w2 = (torch.rand(C, A, device='cuda') * 2 - 1) * 0.08
packed = lc.pack_weight(lib, (w2 * 1.4426950408889634).reshape(C, A, 1))
gmean = x.float().mean(1)
out = torch.empty(B, 2 * C, device='cuda')
st = _hip.current_stream(x)
for name, bound in (('nomax', float(w2.abs().sum(1).max()) * 1.4427 * 1.001), ('online', -1.0)):
    call = lambda: lib.mv_asp_pool_f16(h.data_ptr(), packed.data_ptr(), x.data_ptr(), C, gmean.data_ptr(), C, out.data_ptr(), B, T, C, A,
                                       bound, st)
    for _ in range(3):
        _hip.check(call(), lib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(json.dumps(dict(lib=os.path.basename(os.environ.get('MV_PROBE_LIB', 'product')), form=name, us=round(us, 1),
                          x_GBps=round(B * T * C * 2 / us / 1e3, 1))), flush=True)
