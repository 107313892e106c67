"""Micro-benchmark of the MelSpectrogram front-end alone (HIP events): python tools/bench_melspec.py [B] [L]
MV_MELSPEC_IMPL=dft runs the dense-DFT kernels instead of melspec_tile_kernel."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
import ctypes
cdll = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else None
ms = _hip.MelSpec({}, cdll=cdll)
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([B, L], generator=g)).clamp(-1, 1).cuda()
for _ in range(5):
    out = ms(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    out = ms(wav)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
nbytes = B * (L * 4 + out.shape[1] * out.shape[2] * 4)
print(json.dumps(dict(lib=os.path.basename(os.environ.get('MV_PROBE_LIB', 'product')), impl=os.environ.get('MV_MELSPEC_IMPL', 'fft'), tile_kernel=ms.info()['tile_kernel'], B=B, L=L, frames=out.shape[1], us=round(us, 2),
                      GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 4))))
