"""Micro-benchmark of the Fbank kernel alone (HIP events on the launch stream): python tools/bench_fbank.py [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
import ctypes
cdll = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else None
KERNEL = os.environ.get('MV_BENCH_KERNEL', 'auto')                    # auto | generic | tile (MvFbankCfg.kernel)
FL = float(os.environ.get('MV_BENCH_FRAME_LENGTH', '25'))             # ms: 25 = fbank_tile_kernel<13>, 20 / 30 = <16>
fb_h = _hip.Fbank(dict(sample_frequency=16000, num_mel_bins=80, frame_length=FL), cdll=cdll, kernel=KERNEL)
T = fb_h.num_frames(L)
_out = torch.empty((B, T, 80), dtype=torch.float32, device='cuda')


def fb(wav):
    """mv_fbank_forward (one workgroup per utterance; the entry point every library revision has) -- or the product call with MV_BENCH_WS=1"""
    if os.environ.get('MV_BENCH_WS') == '1':
        return fb_h(wav)
    _hip.check(fb_h._cdll.mv_fbank_forward(fb_h._h, wav.data_ptr(), B, L, wav.stride(0), None, _out.data_ptr(), _hip.current_stream(wav)), fb_h._cdll)
    return _out


fb.info = fb_h.info
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn([B, L], generator=g)).clamp(-1, 1).cuda()
for _ in range(5):
    out = fb(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    out = fb(wav)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(json.dumps(dict(info=fb.info(), lib=os.path.basename(os.environ.get('MV_PROBE_LIB', 'product')), kernel=KERNEL, frame_length_ms=FL, B=B, L=L, us=round(us, 2),
                      GBps=round(B * (L * 4 + T * 320) / us / 1e3, 1), frac_of_8TBps=round(B * (L * 4 + T * 320) / us / 1e3 / 8000, 4))))
