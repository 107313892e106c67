#!/bin/bash
# Round-4 session AF: ten consumer waves (two pixel groups x five blocks) for the 80-channel 3x3 layers against the default five
TAG=${1:-r12af}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 300 python - > $OUT/ab.log 2>&1 <<PY
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
for B, H, W, c in [(16, 40, 149, 80), (64, 40, 149, 80), (16, 40, 149, 112), (16, 80, 298, 48)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, c, generator=g).clamp(0, 20).cuda()
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).cuda()
    c16 = -(-c // 16) * 16
    xs = torch.zeros(B, H, W, c16, device='cuda'); xs[..., :c] = x
    xq = torch.empty_like(xs); _hip.check(cdll.mv_map_split_f32(xs.data_ptr(), xq.data_ptr(), xs.numel(), st()), cdll)
    ys = torch.empty(B, H, W, c16, device='cuda')
    pk = torch.zeros(cdll.mv_conv2ds_packed_elems(c, c, 3), device='cuda'); osc = ctypes.c_float(0)
    _hip.check(cdll.mv_conv2ds_pack_weight(w.data_ptr(), None, c, c, 3, pk.data_ptr(), ctypes.byref(osc), st()), cdll)
    bias = torch.zeros(c16, device='cuda')
    e = _hip.MvConv2dsDesc()
    e.x, e.ldx, e.w, e.bias, e.oscale, e.y, e.ldy = xq.data_ptr(), c16, pk.data_ptr(), bias.data_ptr(), osc.value, ys.data_ptr(), c16
    e.B, e.H, e.W, e.cin16, e.cout16, e.ks, e.stride, e.epi, e.lo, e.hi = B, H, W, c16, c16, 3, 1, 0, 0.0, 20.0
    out = {'B': B, 'HxW': '%dx%d' % (H, W), 'channels': c}
    for rep in range(2):
        for spw in (0, 8, 4):
            e.spw_hint = spw; e.nbw_hint = 1
            try:
                out['spw%d_%d' % (spw, rep)] = round(timed(lambda: _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)), 1)
            except RuntimeError as ex:
                out['spw%d_%d' % (spw, rep)] = 'n/a'
    print(json.dumps(out), flush=True)
PY
cat $OUT/ab.log | grep "^{" | cut -c1-300
