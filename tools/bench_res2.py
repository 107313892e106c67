"""Micro-benchmark of the fused Res2Net chain (mv_res2net_chain_f16) on the Ecapa-1024 shape: B=256, T=298, C=1024, 8 groups.
MV_PROBE_LIB selects an alternative library (tools/probe) for A/B runs inside one box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
for (B, T, width, dil) in [(256, 298, 128, 2), (256, 298, 128, 4), (256, 298, 64, 3)]:
    groups, k = 8, 3
    C = width * groups
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g).half().cuda()
    y = torch.empty_like(x)
    ws = [lc.pack_weight(lib, (torch.randn(width, width, k, generator=g) * (2.0 / (width * k)) ** 0.5).cuda()) for _ in range(groups - 1)]
    par = [[(torch.rand(width, generator=g) + 0.5).cuda() for _ in range(groups - 1)] for _ in range(3)]
    arr = lambda lst: (ctypes.c_void_p * len(lst))(*[t.data_ptr() for t in lst])
    st = _hip.current_stream(x)
    call = lambda: lib.mv_res2net_chain_f16(x.data_ptr(), y.data_ptr(), arr(ws), arr(par[0]), arr(par[1]), arr(par[2]), B, T, C, groups, k, dil, st)
    for _ in range(3):
        _hip.check(call(), lib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f'res2 chain B={B} T={T} width={width} dil={dil}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {y.float().abs().mean().item():.6f}', flush=True)
