"""The two passes over the MFA output x [B, T, 3072] of the ASP head -- hidden 1x1 conv with fused input statistics (pooling.py:104-117), then the
attentive pooling pass (pooling.py:118-127) -- on the whole batch and on batch SLICES (hidden(slice) -> pool(slice) -> next slice): does the second
pass of a slice find its x in the 256 MiB Infinity Cache when the slice is small enough (128 utterances = 235 MB)?  Layer-level C ABI, HIP events;
x is rewritten (as the MFA layer would have just done) before every timed chain, outside the timed span.  python tools/bench_asp_chain.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
import layer_checks as lc
lib = _hip.bind_partial(ctypes.CDLL(os.environ['MV_PROBE_LIB'])) if os.environ.get('MV_PROBE_LIB') else _hip.lib()
B, T, C, A = 256, 298, 3072, 128
src = (torch.randn(B, T, C, device='cuda') * 1.5 + 0.3).half()
x = torch.empty_like(src)
w1 = torch.randn(A, C, 1, device='cuda') * (2.0 / C) ** 0.5
p1 = lc.pack_weight(lib, w1)
w2 = (torch.rand(C, A, device='cuda') * 2 - 1) * 0.08
p2 = lc.pack_weight(lib, (w2 * 1.4426950408889634).reshape(C, A, 1))
bound = float(w2.abs().sum(1).max()) * 1.4427 * 1.001
h = torch.empty(B, T, A, dtype=torch.float16, device='cuda')
gmean = src.float().mean(1)
out = torch.empty(B, 2 * C, device='cuda')
nin = lib.mv_conv1d_in_stats_elems(B, T, C)
isum, isq = torch.empty(nin, device='cuda'), torch.empty(nin, device='cuda')
st = _hip.current_stream(x)
tiles_per_utt = nin // (B * C)


def hidden(b0, nb):
    d = _hip.MvConv1dDesc()
    d.x, d.x_dtype, d.ldx = x[b0].data_ptr(), _hip.MV_DT_F16, C
    d.w_packed = p1.data_ptr()
    d.y, d.y_dtype, d.ldy = h[b0].data_ptr(), _hip.MV_DT_F16, A
    d.B, d.T_in, d.T_out, d.cin, d.cout, d.k, d.dilation, d.stride = nb, T, T, C, A, 1, 1, 1
    d.pad_mode = _hip.MV_PAD_REFLECT
    d.in_stat_sum, d.in_stat_sq = isum[b0 * tiles_per_utt * C:].data_ptr(), isq[b0 * tiles_per_utt * C:].data_ptr()
    _hip.check(lib.mv_conv1d_forward(ctypes.byref(d), st), lib)


def pool(b0, nb):
    _hip.check(lib.mv_asp_pool_f16(h[b0].data_ptr(), p2.data_ptr(), x[b0].data_ptr(), C, gmean[b0].data_ptr(), C, out[b0].data_ptr(), nb, T, C, A, bound, st), lib)


def chain(slices):
    nb = B // slices
    for s in range(slices):
        hidden(s * nb, nb)
        pool(s * nb, nb)


for slices in (1, 2, 4, 1, 2, 4):
    for _ in range(2):
        x.copy_(src)
        chain(slices)
    torch.cuda.synchronize()
    tot, n = 0.0, 8
    for _ in range(n):
        x.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chain(slices)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(json.dumps(dict(slices=slices, chain_us=round(tot / n * 1e3, 1))), flush=True)
# the two passes alone on the whole batch, for reference
for name, fn in (('hidden', lambda: hidden(0, B)), ('pool', lambda: pool(0, B))):
    tot, n = 0.0, 8
    for _ in range(n):
        x.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(json.dumps(dict(kernel=name, us=round(tot / n * 1e3, 1))), flush=True)
