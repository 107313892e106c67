"""Micro-benchmark of the ASP hidden 1x1 conv (3072 -> 128 at the bench shape): fp16 / fp32 output, with / without the fused input
statistics (HIP events): python tools/bench_asp_hidden.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
B, T, cin, cout = 256, 298, 3072, 128
x = (torch.randn(B, T, cin, device='cuda') * 0.5).half()
w = torch.randn(cout, cin, 1, device='cuda') * (2.0 / cin) ** 0.5
packed = lc.pack_weight(lib, w)
nin = lib.mv_conv1d_in_stats_elems(B, T, cin)
isum, isq = torch.empty(nin, device='cuda'), torch.empty(nin, device='cuda')
for y_f32 in (False, True):
    for stats in (False, True):
        y = torch.empty(B, T, cout, dtype=torch.float32 if y_f32 else torch.float16, device='cuda')
        d = _hip.MvConv1dDesc()
        d.x, d.x_dtype, d.ldx = x.data_ptr(), _hip.MV_DT_F16, cin
        d.w_packed = packed.data_ptr()
        d.y, d.y_dtype, d.ldy = y.data_ptr(), (_hip.MV_DT_F32 if y_f32 else _hip.MV_DT_F16), cout
        d.B, d.T_in, d.T_out, d.cin, d.cout, d.k, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
        d.pad_mode, d.tile = _hip.MV_PAD_REFLECT, 160
        if stats:
            d.in_stat_sum, d.in_stat_sq = isum.data_ptr(), isq.data_ptr()
        st = _hip.current_stream(x)
        for _ in range(3):
            _hip.check(lib.mv_conv1d_forward(ctypes.byref(d), st), lib)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.mv_conv1d_forward(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(json.dumps(dict(y='f32' if y_f32 else 'f16', in_stats=stats, us=round(us, 1), x_GBps=round(B * T * cin * 2 / us / 1e3, 1))), flush=True)
