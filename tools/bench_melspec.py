"""Micro-benchmark of the MelSpectrogram front-end alone (HIP events): python tools/bench_melspec.py [B] [L] [readme | n_fft]
MV_MELSPEC_IMPL=dft runs the dense-DFT kernels instead of the FFT kernels; `readme` = the reference README's arguments (n_fft 1024, hop 320,
64 mels), a number = that n_fft with the other defaults."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
import ctypes
cdll = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else None
which = sys.argv[3] if len(sys.argv) > 3 else ''
margs = dict(sample_rate=16000, n_fft=1024, win_length=1024, hop_length=320, f_min=50, f_max=14000, n_mels=64) if which == 'readme' else (dict(n_fft=int(which)) if which else {})
ms = _hip.MelSpec(margs, cdll=cdll)
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
print(json.dumps(dict(lib=os.path.basename(os.environ.get('MV_PROBE_LIB', 'product')), impl=os.environ.get('MV_MELSPEC_IMPL', 'fft'), kernel=ms.info()['kernel'], args=which or 'default', B=B, L=L, frames=out.shape[1], us=round(us, 2),
                      GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 4))))
