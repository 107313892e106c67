"""In-kernel timeline of the fused Res2Net chain (probe build: bash tools/build_probe.sh res2.hip 1): python tools/trace_res2.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
raw = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libres2_probe1.so'))
lib = _hip.bind_partial(raw)
B, T, width, groups, k, dil = 256, 298, 128, 8, 3, 3
C = width * groups
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).half().cuda()
y = torch.empty_like(x)
ws = [lc.pack_weight(lib, (torch.randn(width, width, k, generator=g) * (2.0 / (width * k)) ** 0.5).cuda()) for _ in range(groups - 1)]
par = [[(torch.rand(width, generator=g) + 0.5).cuda() for _ in range(groups - 1)] for _ in range(3)]
arr = lambda lst: (ctypes.c_void_p * len(lst))(*[t.data_ptr() for t in lst])
st = _hip.current_stream(x)
for _ in range(3):
    _hip.check(lib.mv_res2net_chain_f16(x.data_ptr(), y.data_ptr(), arr(ws), arr(par[0]), arr(par[1]), arr(par[2]), B, T, C, groups, k, dil, st), lib)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
raw.mv_debug_trace_read.restype = ctypes.c_int
n = raw.mv_debug_trace_read(buf, 4096)
ev = [(buf[i] >> 4, buf[i] & 15) for i in range(n)]
t0 = ev[0][0]
names = {0: 'prologue done', 1: 'step start', 2: 'weights landed', 3: 'barrier passed', 4: 'MFMAs issued', 5: 'final barrier passed', 6: 'epilogue issued'}
print(f'{n} events; s_memtime ticks relative to the first event (wave 0 of workgroup 0)')
prev = t0
for t, tag in ev:
    print(f'{t - t0:9d}  +{t - prev:7d}  {names.get(tag, tag)}')
    prev = t
