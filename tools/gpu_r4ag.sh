#!/bin/bash
# Round-4 session AG: several blocks per wave on pixel groups for layers with few output channels (launch-shape hints) against the defaults, one call
TAG=${1:-r12ag}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 400 python - > $OUT/ab.log 2>&1 <<PY
import ctypes, json, os, sys
sys.path[:0] = ['$REPO', '$REPO/voiceprintrecognition-pytorch_amd']
import torch
from mvector import _hip
cdll = _hip.lib()
st = lambda: _hip.current_stream(torch.empty(1, device='cuda'))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (B, H, W, cin, cout, ks, res)
CASES = [(16, 80, 298, 48, 48, 3, False), (16, 40, 149, 80, 80, 3, False), (16, 80, 298, 16, 16, 3, False), (16, 40, 149, 32, 32, 3, False), (16, 20, 75, 64, 64, 3, False), (16, 10, 38, 112, 112, 3, False),
         (16, 80, 298, 64, 32, 1, False), (16, 80, 298, 32, 64, 1, True), (16, 40, 149, 64, 128, 1, True)]
for B, H, W, cin, cout, ks, with_res in CASES:
    g = torch.Generator().manual_seed(1)
    r16 = lambda n: -(-n // 16) * 16
    ci, co = r16(cin), r16(cout)
    x = torch.zeros(B, H, W, ci); x[..., :cin] = torch.randn(B, H, W, cin, generator=g).clamp(0, 20)
    x = x.cuda()
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
    xq = torch.empty_like(x); _hip.check(cdll.mv_map_split_f32(x.data_ptr(), xq.data_ptr(), x.numel(), st()), cdll)
    res = torch.randn(B, H, W, co, generator=g).cuda() if with_res else None
    rq = None
    if with_res:
        rq = torch.empty_like(res); _hip.check(cdll.mv_map_split_f32(res.data_ptr(), rq.data_ptr(), res.numel(), st()), cdll)
    ys = torch.empty(B, H, W, co, device='cuda')
    pk = torch.zeros(cdll.mv_conv2ds_packed_elems(cout, cin, ks), device='cuda'); osc = ctypes.c_float(0)
    _hip.check(cdll.mv_conv2ds_pack_weight(w.data_ptr(), None, cout, cin, ks, pk.data_ptr(), ctypes.byref(osc), st()), cdll)
    bias = torch.zeros(co, device='cuda')
    e = _hip.MvConv2dsDesc()
    e.x, e.ldx, e.w, e.bias, e.oscale, e.y, e.ldy = xq.data_ptr(), ci, pk.data_ptr(), bias.data_ptr(), osc.value, ys.data_ptr(), co
    e.res, e.ldres = (rq.data_ptr() if with_res else None), co
    e.B, e.H, e.W, e.cin16, e.cout16, e.ks, e.stride, e.epi, e.lo, e.hi = B, H, W, ci, co, ks, 1, 0, 0.0, 20.0
    out = {'layer': '%dx%d %d->%d%s @%dx%dx%d' % (ks, ks, cin, cout, '+res' if with_res else '', B, H, W)}
    nblk = co // 16
    combos = [(0, 0, 0)]
    for nbw in (2, 3):
        for spw in (8, 4, 2):
            for wgs in (0, 2):
                combos.append((nbw, spw, wgs))
    for nbw, spw, wgs in combos:
        if nbw > nblk:
            continue
        e.nbw_hint, e.spw_hint, e.wgs_hint = nbw, spw, wgs
        key = 'default' if (nbw, spw, wgs) == (0, 0, 0) else 'nbw%d_spw%d_wgs%d' % (nbw, spw, wgs)
        try:
            out[key] = round(timed(lambda: _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)), 1)
        except RuntimeError as ex:
            pass
    e.nbw_hint, e.spw_hint, e.wgs_hint = 0, 0, 0
    out['default_again'] = round(timed(lambda: _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)), 1)
    print(json.dumps(out), flush=True)
PY
grep "^{" $OUT/ab.log | cut -c1-700; grep -v "^{" $OUT/ab.log | tail -3
