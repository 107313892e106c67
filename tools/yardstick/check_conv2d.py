"""Parity of the fp32 conv2d yard-stick (tools/yardstick/conv2d_f32.hip) against F.conv2d in fp32 -- the layer cases the product's test-suite ran
while the ERes2Net family lived on these kernels (rounds 2-3).  On a GPU box:   python tools/yardstick/check_conv2d.py"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, 'tools'), os.path.join(ROOT, 'tests'), ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
import torch.nn.functional as F
from yardstick import binding
from mvector import _hip
from layer_checks import _stream


def conv2d_case(cdll, device, B=2, H=10, W=37, cin=16, cout=16, ks=3, stride=1, stride_w=0, x2_mode=0, epi=0, with_res=False,
                lo=0.0, hi=20.0, seed=0):
    """mv_conv2d_forward against F.conv2d in fp32 (ERes2Net layer: conv -> folded BN -> epilogue)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    r16 = lambda n: -(-n // 16) * 16
    cin_a = cin if x2_mode != 2 else cin // 2          # concat: two operands of cin/2 channels each
    lda = r16(cin_a) + 8
    xa = rn(B, H, W, lda)
    xb = rn(B, H, W, lda) if x2_mode else None
    w = rn(cout, cin, ks, ks) * (2.0 / (cin * ks * ks)) ** 0.5
    bn_scale = torch.rand(cout, generator=g) + 0.5
    bias = rn(cout) * 0.3
    p = ks // 2
    sw = stride_w or stride
    Ho, Wo = (H + 2 * p - ks) // stride + 1, (W + 2 * p - ks) // sw + 1
    c16 = r16(cout)
    ldy = c16 + 4
    res = rn(B, Ho, Wo, c16) if (with_res or epi == 2) else None
    res2 = rn(B, Ho, Wo, c16) if epi == 2 else None
    for t in (res, res2):   # operands are maps of the same model: their padded channels are zero
        if t is not None:
            t[..., cout:] = 0.0
    dev = lambda t: None if t is None else t.to(device).contiguous()
    xad, xbd, resd, res2d = dev(xa), dev(xb), dev(res), dev(res2)
    wd, sd = dev(w), dev(bn_scale)
    if x2_mode == 2:
        # packed K axis = [r16(cin_a) channels of x | r16(cin_a) channels of x2]
        wfull = torch.zeros(cout, 2 * r16(cin_a), ks, ks)
        wfull[:, :cin_a] = w[:, :cin_a]
        wfull[:, r16(cin_a):r16(cin_a) + cin_a] = w[:, cin_a:]
        wd = dev(wfull)
        cin_k = 2 * r16(cin_a)
    else:
        cin_k = cin
    n = cdll.mv_conv2d_packed_elems(cout, cin_k, ks)
    packed = torch.zeros(n, dtype=torch.float32, device=device)
    _hip.check(cdll.mv_conv2d_pack_weight(wd.data_ptr(), sd.data_ptr(), cout, cin_k, ks, packed.data_ptr(), _stream(wd)), cdll)
    biasd = torch.zeros(c16, device=device)
    biasd[:cout] = dev(bias)
    y = torch.full((B, Ho, Wo, ldy), 7.0, dtype=torch.float32, device=device)
    d = binding.MvConv2dDesc()
    d.x, d.ldx = xad.data_ptr(), lda
    d.x2, d.ldx2, d.x2_mode, d.cin1 = (xbd.data_ptr() if x2_mode else None), lda, x2_mode, r16(cin_a)
    d.w, d.bias = packed.data_ptr(), biasd.data_ptr()
    d.res, d.ldres = (resd.data_ptr() if res is not None else None), c16
    d.res2, d.ldres2 = (res2d.data_ptr() if res2 is not None else None), c16
    d.y, d.ldy = y.data_ptr(), ldy
    d.B, d.H, d.W, d.cin16, d.cout16, d.ks, d.stride, d.epi = B, H, W, r16(cin_k), c16, ks, stride, epi
    d.stride_w = stride_w
    d.lo, d.hi = lo, hi
    _hip.check(cdll.mv_conv2d_forward(ctypes.byref(d), _stream(xad)), cdll)
    if device != 'cpu':
        torch.cuda.synchronize()

    xin = xa.float()[..., :cin_a]
    if x2_mode == 1:
        xin = xin + xb.float()[..., :cin_a]
    elif x2_mode == 2:
        xin = torch.cat([xin, xb.float()[..., :cin_a]], dim=-1)
    weff = w * bn_scale.view(-1, 1, 1, 1)
    ref = F.conv2d(xin.permute(0, 3, 1, 2), weff, bias, stride=(stride, sw), padding=p).permute(0, 2, 3, 1)
    if epi == 0:
        if with_res:
            ref = ref + res.float()[..., :cout]
        ref = ref.clamp(lo, hi)
    elif epi == 1:
        ref = F.silu(ref)
    else:
        t = torch.tanh(ref)
        ref = res.float()[..., :cout] * (1 + t) + res2.float()[..., :cout] * (1 - t)
    got = y.cpu().float()
    assert torch.all(got[..., c16:] == 7.0), 'kernel wrote outside its channel slice'
    if c16 > cout and epi != 2:
        assert torch.all(got[..., cout:c16] == (0.0 if epi != 0 else min(max(0.0, lo), hi))), 'padded channels must stay zero'
    err = (got[..., :cout] - ref).abs().max().item()
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert err < tol, f'conv2d mismatch {err} (tol {tol})'
    return err


CONV2D_CASES = [
    dict(),                                                                  # 3x3 16 -> 16
    dict(cin=13, cout=13, x2_mode=1),                                        # ERes2NetV2 width 13 (padded), sp + spx[i]
    dict(cin=32, cout=32, ks=1, stride=2, H=8, W=41),                        # strided 1x1 (conv1 / shortcut of a stage's first block)
    dict(cin=64, cout=128, ks=3, stride=2, H=9, W=150, hi=65504.0, lo=-65504.0),  # layer1_downsample: no BN / activation
    dict(cin=32, cout=64, ks=1, with_res=True, W=300, H=3),                  # conv3 + bn3 + residual + ReLU20, 3 time tiles
    dict(cin=128, cout=16, ks=1, x2_mode=2, epi=1),                          # AFF local_att[0:3]: cat -> 1x1 -> BN -> SiLU
    dict(cin=16, cout=64, ks=1, epi=2),                                      # AFF local_att[3:5] + fusion
    dict(cin=104, cout=104, ks=3, H=5, W=40, B=1),                           # width 104: two K chunks, 7 channel blocks -> NB 1
    dict(cin=256, cout=512, ks=3, stride=2, H=6, W=20, B=1, hi=65504.0, lo=-65504.0),  # layer3_downsample
    dict(cin=48, cout=144, ks=3, H=3, W=20, B=1),                            # 9 channel blocks: two uneven tiles of 5 (one clamped block)
    dict(cin=48, cout=208, ks=1, H=2, W=37, B=2, with_res=True),             # 1x1: 13 blocks -> 4 tiles of 4 (3 clamped blocks)
    dict(cin=32, cout=48, ks=1, stride=2, H=5, W=37, B=2),                   # strided 1x1 on odd sizes: 5 x 37 -> 3 x 19
    dict(cin=16, cout=32, ks=3, stride=2, H=5, W=33, B=1, x2_mode=1),        # strided 3x3 on odd sizes with the input sum
    dict(cin=32, cout=32, ks=3, stride=2, stride_w=1, H=9, W=45, B=2, hi=65504.0),   # CAM++ head (fp32 form): stride on the frequency axis only
    dict(cin=32, cout=32, ks=1, stride=2, stride_w=1, H=8, W=37, B=2, hi=65504.0, lo=-65504.0),  # its 1x1 shortcut conv
]



if __name__ == '__main__':
    cdll = binding.load()
    for idx, cfg in enumerate(CONV2D_CASES):
        print(idx, cfg, conv2d_case(cdll, 'cuda', seed=idx, **cfg), flush=True)
    print('conv2d yard-stick: all', len(CONV2D_CASES), 'cases within tolerance')
