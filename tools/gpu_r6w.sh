#!/bin/bash
# Round-6 session W: the vendor library (torch.matmul fp16 = hipBLASLt) and the ring GEMM on the same box, both hot (warm-up of ~40 ms each), alternating
TAG=${1:-r15w}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python - <<'PY' 2>&1 | grep -v Warning | tee $OUT/ring_vs_hipblaslt_hot.log
import sys, ctypes, json, torch
sys.path[:0] = ['.', 'tests', 'voiceprintrecognition-pytorch_amd']
from mvector import _hip
import layer_checks as lc
lib = _hip.lib()
B, T = 256, 298
def ours(cin, cout, warm, iters):
    x = (torch.randn(B, T, cin, device='cuda') * 0.5).half()
    w = torch.randn(cout, cin, 1, device='cuda') * (2.0 / cin) ** 0.5
    packed = lc.pack_weight(lib, w)
    bias = torch.randn(cout, device='cuda') * 0.1; scale = torch.rand(cout, device='cuda') + 0.5; shift = torch.randn(cout, device='cuda') * 0.1
    y = torch.empty(B, T, cout, dtype=torch.float16, device='cuda')
    probe = torch.zeros(4 * 264, dtype=torch.int64, device='cuda')
    d = _hip.MvConv1dDesc()
    d.x, d.x_dtype, d.ldx = x.data_ptr(), _hip.MV_DT_F16, cin
    d.w_packed, d.bias, d.scale, d.shift = packed.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr()
    d.pre_act, d.post_act = 1, 0
    d.y, d.y_dtype, d.ldy = y.data_ptr(), _hip.MV_DT_F16, cout
    d.B, d.T_in, d.T_out, d.cin, d.cout, d.k, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
    d.pad, d.pad_mode, d.tile = 0, _hip.MV_PAD_REFLECT, 256
    d.clock_probe = probe.data_ptr()
    st = _hip.current_stream(x)
    for _ in range(warm): lib.mv_conv1d_forward(ctypes.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.mv_conv1d_forward(ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    t = probe.cpu().reshape(-1, 4).double(); t = t[t[:, 3] > t[:, 2]]
    ghz = ((t[:, 1] - t[:, 0]) / (t[:, 3] - t[:, 2]) * 0.1).median().item()
    return us, ghz
def vendor(M, N, K, warm, iters):
    a = torch.randn(M, K, device='cuda', dtype=torch.float16); b = torch.randn(N, K, device='cuda', dtype=torch.float16)
    for _ in range(warm): c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for rep in range(2):
    for cin, cout, warm, iters in ((3072, 3072, 30, 10), (1024, 1024, 250, 40)):
        us, ghz = ours(cin, cout, warm, iters)
        fl = 2.0 * B * T * cin * cout
        print(json.dumps(dict(kernel='ring GEMM + bias / ReLU / BatchNorm epilogue', K=cin, N=cout, rows=B * T, us=round(us, 1), tflops=round(fl / us / 1e6, 1), clock_ghz=round(ghz, 3))), flush=True)
        usv = vendor(B * T, cout, cin, warm, iters)
        print(json.dumps(dict(kernel='torch.matmul fp16 (hipBLASLt), no epilogue', K=cin, N=cout, rows=B * T, us=round(usv, 1), tflops=round(fl / usv / 1e6, 1))), flush=True)
PY
