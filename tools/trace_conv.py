"""In-kernel timeline of the persistent conv kernel (probe build -DMV_PROBE=3): python tools/trace_conv.py [cin] [cout]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
raw = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libconv1d_probe3.so'))
lib = _hip.bind_partial(raw)
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B, T = 256, 298
x = (torch.randn(B, T, cin, device='cuda') * 0.5).half()
w = torch.randn(cout, cin, 1, device='cuda') * (2.0 / cin) ** 0.5
packed = lc.pack_weight(lib, w)
bias = torch.randn(cout, device='cuda') * 0.1
scale = torch.rand(cout, device='cuda') + 0.5
shift = torch.randn(cout, device='cuda') * 0.1
y = torch.empty(B, T, cout, dtype=torch.float16, device='cuda')
d = _hip.MvConv1dDesc()
d.x, d.x_dtype, d.ldx = x.data_ptr(), _hip.MV_DT_F16, cin
d.w_packed, d.bias, d.scale, d.shift = packed.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr()
d.pre_act, d.post_act = 1, 0
d.y, d.y_dtype, d.ldy = y.data_ptr(), _hip.MV_DT_F16, cout
d.B, d.T_in, d.T_out, d.cin, d.cout, d.k, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
d.pad, d.pad_mode, d.tile = 0, _hip.MV_PAD_REFLECT, 256
st = _hip.current_stream(x)
for _ in range(3):
    _hip.check(lib.mv_conv1d_forward(ctypes.byref(d), st), lib)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    lib.mv_conv1d_forward(ctypes.byref(d), st)
e1.record()
torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) / 5 * 1e3
buf = (ctypes.c_ulonglong * 8192)()
raw.mv_debug_trace_read.restype = ctypes.c_int
n = raw.mv_debug_trace_read(buf, 8192)
ev = [(buf[i] >> 2, buf[i] & 3) for i in range(n)]
t0 = ev[0][0]
nst = cin // 64
print(f'{n} events, {nst} stages per tile; cycles (s_memtime) relative to the first event')
# per stage: tag0 (loads landed) tag1 (barrier passed) tag2 (next stage requested) tag3 (MFMAs issued)
rows, cur = [], {}
for t, tag in ev:
    if tag == 0 and cur:
        rows.append(cur); cur = {}
    cur[tag] = t - t0
rows.append(cur)
for i, r in enumerate(rows[: 3 * nst + 2]):
    s = i % nst
    prev3 = rows[i - 1].get(3) if i else None
    print(f'tile {i // nst} stage {s:2d}: wait_loads_done={r.get(0)} barrier+{(r.get(1, 0) - r.get(0, 0))} issue+{(r.get(2, 0) - r.get(1, 0))} '
          f'epi+mma+{(r.get(3, 0) - r.get(2, 0))}  | stage start since prev stage end: {None if prev3 is None else r.get(0) - prev3}')
tot = rows[-1].get(0, 0)
print('kernel span (cycles):', tot, ' tiles:', (len(rows) - 1) / nst)
print(f'launch duration (HIP events, probe build): {wall_us:.1f} us -> {tot / wall_us / 1e3:.3f} GHz of s_memtime ticks if the traced workgroup spans the launch')
