"""In-kernel timeline of cam_dense_layer_kernel (text-edited copy, tools/probe/libcam_trace.so; never shipped): s_memtime of wave 0 of
workgroup 100 at the phase boundaries.  usage: python tools/probe_camdense.py  (build);  MV_PROBE_LIB=tools/probe/libcam_trace.so python tools/probe_camdense.py run"""
import ctypes, glob, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
EV = ['start', 'bn1 tables in LDS', 'parameters + first x stage landed', 'stage loop done', 'h written', 'partial sums', 'ctx', 'FC1', 'gate', 'k=3 conv + stores issued']


def build():
    d = '/tmp/probe_cam'
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'camdense.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    p = os.path.join(d, 'camdense.hip')
    s = open(p).read()
    def ins(marker, ev, before=False, nth=0):
        nonlocal s
        idx = -1
        for _ in range(nth + 1):
            idx = s.index(marker, idx + 1)
        code = f'    CD_T({ev});\n'
        pos = idx if before else s.index('\n', idx) + 1
        s = s[:pos] + code + s[pos:]
    s = s.replace('namespace mv {\n\nconstexpr int CD_THREADS', 'namespace mv {\n__device__ unsigned long long g_cd_trace[64 * 16];\n'
                  '#define CD_T(ev) do { if (blockIdx.x == 100 && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); '
                  'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_cd_trace[a.seq * 16 + ev] = t_; } } while (0)\nconstexpr int CD_THREADS', 1)
    ins('    const int nst = a.cin_pad / 64;', 0)
    ins('    // ---- parameters of the later phases are requested NOW', 1, before=True)
    ins('    transform(0);\n    float4v acc[2][5];', 2, before=True)
    ins('    wait_vm<0>();   // only padding transfers are left', 3, before=True)
    ins('    // ---- phase B: context gate per 100-frame segment ----', 4, before=True)
    ins('        __syncthreads();\n        if (tid < CD_MAX_SEG * CD_BN) {', 5)
    ins('        {   // g1 = ReLU(Wa ctx + ba)', 6, before=True)
    ins('        {   // gate = sigmoid(Wb g1 + bb)', 7, before=True)
    ins('    // ---- phase C: y = conv_k3(h) * gate', 8, before=True)
    s = s.replace('                *reinterpret_cast<half4v*>(xb + (int64_t)t * a.ldx + a.cin + co) = hv;\n            }\n        }\n    }\n}',
                  '                *reinterpret_cast<half4v*>(xb + (int64_t)t * a.ldx + a.cin + co) = hv;\n            }\n        }\n    }\n    CD_T(9);\n}', 1)
    s = s.replace('}  // namespace mv', 'extern "C" int mv_cam_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cd_trace), sizeof(g_cd_trace)); }\n}  // namespace mv', 1)
    s = s.replace('    int T2, cin, cin_pad, dil, seg_len;\n};', '    int T2, cin, cin_pad, dil, seg_len, seq;\n};', 1)
    s = s.replace('a.seg_len = seg_len;', 'a.seg_len = seg_len; static int seq_ = 0; a.seq = seq_++ % 52;', 1)
    open(p, 'w').write(s)
    obj = '/tmp/probe_cam/cam.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-I', d, '-I',
                           os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/camdense.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libcam_trace.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


def run():
    sys.path[:0] = [REPO, os.path.join(REPO, 'tests'), PKG]
    import numpy as np
    import torch
    from mvector import _hip
    _hip._lib = _hip.bind(ctypes.CDLL(os.path.join(REPO, 'tools', 'probe', 'libcam_trace.so')))
    import bench
    dev = torch.device('cuda', 0)
    featurizer, model, _ = bench.build('campp', dev)
    g = torch.Generator().manual_seed(1)
    wav = (0.1 * torch.randn([256, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(featurizer(wav))
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16, dtype=np.uint64)
    assert _hip._lib.mv_cam_trace_read(ctypes.c_void_p(buf.ctypes.data)) == 0
    tr = buf.reshape(64, 16).astype(np.int64)
    print('cam_dense_layer_kernel timeline, workgroup 100 wave 0, s_memtime ticks between events (one forward of 256 x 3 s):')
    print('events: ' + ' | '.join(f'{i}={n}' for i, n in enumerate(EV)))
    tot = np.zeros(9)
    for l in range(52):
        d = np.diff(tr[l, :10])
        tot += d
        if l in (0, 5, 11, 12, 24, 35, 36, 44, 51):
            print(f'  layer {l:2d}: ' + ' '.join(f'{int(v):6d}' for v in d) + f'   total {int(tr[l, 9] - tr[l, 0]):6d}')
    print('  mean   : ' + ' '.join(f'{int(v / 52):6d}' for v in tot) + f'   total {int(tot.sum() / 52):6d}')


if __name__ == '__main__':
    (run if len(sys.argv) > 1 and sys.argv[1] == 'run' else build)()
