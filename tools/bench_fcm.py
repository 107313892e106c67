"""Micro-benchmark of the fused BasicResBlock kernel (mv_fcm_block_f16) on the four block shapes of the CAM++ head at B=256, T=298
(layer1.0: 80 -> 40 rows strided, layer1.1: 40 rows, layer2.0: 40 -> 20 strided, layer2.1: 20 rows).  Every call works on its own
input / output pair out of a rotation larger than L2 + MALL (the model's blocks read what the previous launch wrote: both numbers are
printed, `warm` = the same buffers every call).  MV_PROBE_LIB selects an alternative library (tools/probe) for A/B runs in one box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
tag = os.path.basename(os.environ.get('MV_PROBE_LIB', 'product'))
B, T = 256, 298
g = torch.Generator().manual_seed(0)
for (Fin, sf) in [(80, 2), (40, 1), (40, 2), (20, 1)]:
    Fout = (Fin - 1) // sf + 1
    nrot = max(2, int(1.2e9 / (B * Fin * T * 64)))
    xs = [torch.randn(B, Fin, T, 32, generator=g).half().cuda() for _ in range(nrot)]
    ys = [torch.empty(B, Fout, T, 32, dtype=torch.float16, device='cuda') for _ in range(nrot)]
    w1 = (torch.randn(9, 32, 32, generator=g) * 0.08).half().cuda()
    w2 = (torch.randn(10, 32, 32, generator=g) * 0.08).half().cuda()
    b1 = (torch.randn(32, generator=g) * 0.1).cuda()
    b2 = (torch.randn(32, generator=g) * 0.1).cuda()
    st = _hip.current_stream(xs[0])
    call = lambda i: lib.mv_fcm_block_f16(xs[i].data_ptr(), Fin, sf, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 1 if sf == 2 else 0,
                                          ys[i].data_ptr(), Fout * T * 32, T * 32, 32, B, T, st)
    res = {}
    for mode in ('cold', 'warm'):
        for i in range(3):
            _hip.check(call(i % nrot), lib)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for i in range(n):
            call(i % nrot if mode == 'cold' else 0)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / n * 1e3
    traffic = B * T * 64 * (Fin + Fout)
    print(f'{tag}: fcm block Fin={Fin} sf={sf}: cold {res["cold"]:.1f} us = {traffic / res["cold"] * 1e-6:.2f} TB/s   warm {res["warm"]:.1f} us   '
          f'({(Fout + 1) and res["cold"] / (Fout + 1):.2f} us per step)  checksum {ys[0].float().abs().mean().item():.5f}', flush=True)
    del xs, ys

if hasattr(lib, 'mv_fcm_trace_read'):
    # timeline of the last launch (Fin=20, sf=1): ticks of s_memtime (100 MHz reference) relative to the producers' first barrier
    import numpy as np
    buf = np.zeros(8 * 48 * 4, dtype=np.uint64)
    assert lib.mv_fcm_trace_read(ctypes.c_void_p(buf.ctypes.data)) == 0
    tr = buf.reshape(8, 48, 4).astype(np.int64)
    t00 = tr[0, 0, 0]
    print('timeline (workgroup 100, ticks since the first barrier): wave | step: E0 after barrier, E1 requests / stores done, E2 matrix phase done, E3 end of step')
    for w in (0, 3, 4, 7):
        for i in range(2, 9):
            e = tr[w, i] - t00
            print(f'  wave {w} step {i}: ' + ' '.join(f'{int(v):7d}' for v in e) + '   phases ' + ' '.join(f'{int(e[k + 1] - e[k]):5d}' for k in range(3)) +
                  f'   step {int(tr[w, i + 1, 0] - tr[w, i, 0]):5d}')
