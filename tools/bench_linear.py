"""Micro-benchmark of mv_linear_f32 on the long-K layers of the Ecapa ASP head (context bias [256, 6144] x [128, 6144], final linear
[256, 6144] x [192, 6144]) and the SE excitation shapes.  MV_PROBE_LIB selects an alternative library (tools/probe) for A/B runs in one box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
big = torch.empty(64 * 1024 * 1024, device='cuda')   # 256 MB: flushes L2 / Infinity Cache between timed calls
for (B, K, O) in [(256, 6144, 128), (256, 6144, 192), (256, 1024, 128), (256, 128, 1024), (2048, 6144, 192)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, K, generator=g).cuda()
    w = (torch.randn(O, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(O, generator=g).cuda()
    y = torch.empty(B, O, device='cuda')
    st = _hip.current_stream(x)
    call = lambda: lib.mv_linear_f32(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), 0, y.data_ptr(), O, B, K, O, st)
    _hip.check(call(), lib)
    err = (y - (x.double() @ w.double().T + bias.double()).float()).abs().max().item()
    ts = []
    for _ in range(10):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    line = f'linear B={B} K={K} O={O}: direct cold {sorted(ts)[len(ts) // 2]:.1f} us  warm {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  max err {err:.2e}'
    need = int(lib.mv_linear_f32_workspace_floats(B, K, O)) if hasattr(lib, 'mv_linear_f32_workspace_floats') else 0
    if need:   # the split-K workspace form (round 5)
        ws = torch.empty(need, device='cuda')
        y2 = torch.empty(B, O, device='cuda')
        call2 = lambda: lib.mv_linear_f32_ws(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), 0, y2.data_ptr(), O, B, K, O, ws.data_ptr(), need, st)
        _hip.check(call2(), lib)
        err2 = (y2 - (x.double() @ w.double().T + bias.double()).float()).abs().max().item()
        ts2 = []
        for _ in range(10):
            big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call2(); e1.record()
            torch.cuda.synchronize()
            ts2.append(e0.elapsed_time(e1) * 1e3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call2()
        e1.record()
        torch.cuda.synchronize()
        line += f'  |  split-K cold {sorted(ts2)[len(ts2) // 2]:.1f} us  warm {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  max err {err2:.2e}'
    print(line, flush=True)
