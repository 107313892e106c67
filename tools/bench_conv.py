"""Micro-benchmark of mv_conv1d_forward tile variants on the backbone's GEMM shapes (HIP events)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
B, T = int(os.environ.get('MV_BENCH_B', '256')), int(os.environ.get('MV_BENCH_T', '298'))
TILES = [int(t) for t in os.environ.get('MV_BENCH_TILES', '128,256').split(',')]
shapes = [('c2c 1024->1024 k1', 1024, 1024, 1, 1), ('mfa 3072->3072 k1', 3072, 3072, 1, 1), ('asp 3072->128 k1', 3072, 128, 1, 1),
          ('res2 128->128 k3d3', 128, 128, 3, 3), ('c2c 512->512 k1', 512, 512, 1, 1), ('mfa 1536->1536', 1536, 1536, 1, 1)]
only = os.environ.get('MV_BENCH_SHAPES')
for name, cin, cout, k, dil in shapes:
    if only and not any(o in name for o in only.split(',')):
        continue
    PADX = int(os.environ.get('MV_PADX', '0'))
    xfull = (torch.randn(B, T, cin + PADX, device='cuda') * 0.5).half()
    x = xfull
    w = torch.randn(cout, cin, k, device='cuda') * (2.0 / (cin * k)) ** 0.5
    packed = lc.pack_weight(lib, w)
    bias = torch.randn(cout, device='cuda') * 0.1
    scale = torch.rand(cout, device='cuda') + 0.5
    shift = torch.randn(cout, device='cuda') * 0.1
    y = torch.empty(B, T, cout + PADX, dtype=torch.float16, device='cuda')
    for tile in TILES:
        if tile == 256 and cout % 256:
            continue
        d = _hip.MvConv1dDesc()
        d.x, d.x_dtype, d.ldx = x.data_ptr(), _hip.MV_DT_F16, cin + PADX
        d.w_packed, d.bias, d.scale, d.shift = packed.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr()
        d.pre_act, d.post_act = 1, 0
        d.y, d.y_dtype, d.ldy = y.data_ptr(), _hip.MV_DT_F16, cout + PADX
        d.B, d.T_in, d.T_out, d.cin, d.cout, d.k, d.dilation, d.stride = B, T, T, cin, cout, k, dil, 1
        d.pad, d.pad_mode, d.tile = dil * (k - 1) // 2, _hip.MV_PAD_REFLECT, tile
        d.persist_blocks_hint = int(os.environ.get('MV_BENCH_BLOCKS', '0'))   # resident workgroups of the persistent kernels (0: one per CU)
        probe = None
        if os.environ.get('MV_BENCH_CLOCK') == '1' and hasattr(d, 'clock_probe'):   # the launch's sustained shader clock (ring kernel only)
            probe = torch.zeros(4 * 264, dtype=torch.int64, device='cuda')
            d.clock_probe = probe.data_ptr()
        st = _hip.current_stream(x)
        for _ in range(int(os.environ.get('MV_BENCH_WARM', '3'))):   # (the chip needs ~30 ms under load to reach its clock: MV_BENCH_WARM=30 for hot figures)
            _hip.check(lib.mv_conv1d_forward(ctypes.byref(d), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        prof = os.environ.get('MV_BENCH_PROF') == '1'   # per-launch HIP events of the library: the ring kernel (class 3) and the other conv1d launches of the call
        def prof_read(k, reset=0):
            c, ms, w = ctypes.c_int32(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
            lib.mv_profile_read(k, ctypes.byref(c), ctypes.byref(ms), ctypes.byref(w), reset)
            return c.value, ms.value
        if prof:
            prof_read(0, 1)
            lib.mv_profile_enable(1)
        e0.record()
        for _ in range(n):
            lib.mv_conv1d_forward(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        extra = {}
        if prof:
            lib.mv_profile_enable(0)
            (c3, ms3), (c0, ms0) = prof_read(3), prof_read(0, 1)
            extra = dict(ring_us=round(ms3 / max(c3, 1) * 1e3, 1), other_launches=(c0 - c3) // n, other_us=round((ms0 - ms3) / n * 1e3, 1))
        us = e0.elapsed_time(e1) / n * 1e3
        tf = 2.0 * B * T * cin * cout * k / us / 1e6
        rec = dict(B=B, shape=name, tile=tile, us=round(us, 1), TFLOPs=round(tf, 1), **extra)
        if probe is not None:
            t = probe.cpu().reshape(-1, 4).double()
            t = t[t[:, 3] > t[:, 2]]
            if t.shape[0]:
                rec['clock_ghz'] = round(((t[:, 1] - t[:, 0]) / (t[:, 3] - t[:, 2]) * 0.1).median().item(), 3)
        print(json.dumps(rec), flush=True)
