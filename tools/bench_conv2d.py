"""ERes2Net conv layers alone, fp32 form (mv_conv2d_forward of the yard-stick library, tools/yardstick/) beside the split-fp16 form (mv_conv2ds_forward, conv2ds.hip), HIP events.
usage: python tools/bench_conv2d.py [B]     MV_BENCH_SWEEP=1 also runs the tile-shape hints of the split form
Shapes: the 54.9 M ERes2NetV2 (m_channels 96, base_width 26, scale 4; widths 39 / 78 / 156 / 312 padded to 48 / 80 / 160 / 320) at 3 s."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cdll = _hip.bind(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()   # (a probe / baseline build of the library for an A/B inside one call)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from yardstick import binding as ybinding
ycdll = ybinding.load()
st = lambda: _hip.current_stream(torch.empty(1, device='cuda'))
r16 = lambda n: -(-n // 16) * 16
# (name, H, W, cin, cout, ks, stride, with_res)
SHAPES = [
    ('s1 conv1 96->192', 80, 298, 96, 192, 1, 1, False),
    ('s1 3x3 48->48', 80, 298, 48, 48, 3, 1, False),
    ('s1 conv3 192->192+res', 80, 298, 192, 192, 1, 1, True),
    ('s2 conv1 192->320 /2', 80, 298, 192, 320, 1, 2, False),
    ('s2 3x3 80->80', 40, 149, 80, 80, 3, 1, False),
    ('s2 conv3 320->384+res', 40, 149, 320, 384, 1, 1, True),
    ('s3 conv1 768->640', 20, 75, 768, 640, 1, 1, False),
    ('s3 3x3 160->160', 20, 75, 160, 160, 3, 1, False),
    ('s3 conv3 640->768+res', 20, 75, 640, 768, 1, 1, True),
    ('s4 conv1 1536->1280', 10, 38, 1536, 1280, 1, 1, False),
    ('s4 3x3 320->320', 10, 38, 320, 320, 3, 1, False),
    ('s4 conv3 1280->1536+res', 10, 38, 1280, 1536, 1, 1, True),
    ('ds 3x3 768->1536 /2', 20, 75, 768, 1536, 3, 2, False),
    ('m32 s1 conv1 64->32', 80, 298, 64, 32, 1, 1, False),
    ('m32 s1 3x3 16->16', 80, 298, 16, 16, 3, 1, False),
    ('m32 s1 conv3 32->64+res', 80, 298, 32, 64, 1, 1, True),
    ('m32 s2 3x3 32->32', 40, 149, 32, 32, 3, 1, False),
]
only = os.environ.get('MV_BENCH_SHAPES')
if only:
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in only.split(','))]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, H, W, cin, cout, ks, stride, with_res in SHAPES:
    g = torch.Generator().manual_seed(1)
    p = ks // 2
    Ho, Wo = (H + 2 * p - ks) // stride + 1, (W + 2 * p - ks) // stride + 1
    x = (torch.randn(B, H, W, cin, generator=g).clamp(0, 20)).cuda()
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    res = torch.randn(B, Ho, Wo, cout, generator=g).cuda() if with_res else None
    y = torch.empty(B, Ho, Wo, cout, device='cuda')
    gflop = 2.0 * B * Ho * Wo * cin * cout * ks * ks / 1e9
    mbytes = 4.0 * (x.numel() + y.numel() + (res.numel() if with_res else 0)) / 1e6
    # fp32 form (the yard-stick: not in the product library)
    n = ycdll.mv_conv2d_packed_elems(cout, cin, ks)
    pk = torch.zeros(n, device='cuda')
    _hip.check(ycdll.mv_conv2d_pack_weight(w.data_ptr(), None, cout, cin, ks, pk.data_ptr(), st()), ycdll)
    d = ybinding.MvConv2dDesc()
    d.x, d.ldx, d.w, d.bias, d.y, d.ldy = x.data_ptr(), cin, pk.data_ptr(), bias.data_ptr(), y.data_ptr(), cout
    d.res, d.ldres = (res.data_ptr() if with_res else None), cout
    d.B, d.H, d.W, d.cin16, d.cout16, d.ks, d.stride, d.epi, d.lo, d.hi = B, H, W, cin, cout, ks, stride, 0, 0.0, 20.0
    t32 = timed(lambda: _hip.check(ycdll.mv_conv2d_forward(ctypes.byref(d), st()), ycdll))
    y32 = y.clone()
    # split form
    sp = lambda t: None if t is None else (lambda o: (_hip.check(cdll.mv_map_split_f32(t.data_ptr(), o.data_ptr(), t.numel(), st()), cdll), o)[1])(torch.empty_like(t))
    xs, rs = sp(x), sp(res)
    ys = torch.empty_like(y)
    n = cdll.mv_conv2ds_packed_elems(cout, cin, ks)
    pks = torch.zeros(n, device='cuda')
    osc = ctypes.c_float(0)
    _hip.check(cdll.mv_conv2ds_pack_weight(w.data_ptr(), None, cout, cin, ks, pks.data_ptr(), ctypes.byref(osc), st()), cdll)
    e = _hip.MvConv2dsDesc()
    e.x, e.ldx, e.w, e.bias, e.oscale, e.y, e.ldy = xs.data_ptr(), cin, pks.data_ptr(), bias.data_ptr(), osc.value, ys.data_ptr(), cout
    e.res, e.ldres = (rs.data_ptr() if with_res else None), cout
    e.B, e.H, e.W, e.cin16, e.cout16, e.ks, e.stride, e.epi, e.lo, e.hi = B, H, W, cin, cout, ks, stride, 0, 0.0, 20.0
    hints = [(0, 0, 0, 0, 0, 0, 0)]   # nbw, ct, rows, ring, wgs, spw, nprod
    if os.environ.get('MV_BENCH_SWEEP') == '1':
        if cout >= 128:
            hints += [(2, 0, 0, 0, 0, 0, 2), (2, 0, 0, 0, 0, 0, 4), (3, 0, 0, 0, 0, 0, 4), (3, 0, 0, 0, 0, 0, 2)]
        if cout <= 64:
            hints += [(1, 0, 0, 0, 0, 8, 0), (1, 0, 0, 0, 0, 4, 0)]
        if ks == 3 and stride == 1 and cout >= 128:
            hints += [(0, 0, 4, 0, 0, 0, 0)]
    out = dict(layer=name, B=B, gflop=round(gflop, 1), mbytes=round(mbytes, 1), f32_us=round(t32, 1), f32_tflops=round(gflop / t32 * 1e3, 1))
    for nbw, ct, rows, ring, wgs, spw, nprod in hints:
        e.nbw_hint, e.ct_hint, e.rows_hint, e.ring_hint, e.wgs_hint, e.spw_hint, e.nprod_hint = nbw, ct, rows, ring, wgs, spw, nprod
        key = 'split' if (nbw, ct, rows, ring, wgs, spw, nprod) == (0, 0, 0, 0, 0, 0, 0) else 'nbw%d_rows%d_spw%d_prod%d' % (nbw, rows, spw, nprod)
        try:
            ts = timed(lambda: _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll))
        except RuntimeError as ex:
            out[key] = 'n/a'
            continue
        out[key + '_us'] = round(ts, 1)
        if (nbw, ct, rows, ring, wgs, spw, nprod) == (0, 0, 0, 0, 0, 0, 0):
            ym = torch.empty_like(ys)
            _hip.check(cdll.mv_map_merge_f32(ys.data_ptr(), ym.data_ptr(), ys.numel(), st()), cdll)
            torch.cuda.synchronize()
            out['max_abs_diff_vs_f32'] = float((ym - y32).abs().max())
            out['split_GBps'] = round(mbytes / ts * 1e3, 0)
            out['split_mfma_tflops_x3'] = round(3 * gflop / ts * 1e3, 0)
            out['speedup'] = round(t32 / ts, 2)
    print(json.dumps(out), flush=True)
