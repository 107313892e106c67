"""In-kernel timeline of cam_dense_block_kernel (text-edited copy -> tools/probe/libcamblk_trace.so; never shipped): s_memtime of wave 0
of workgroup 100 at the phase boundaries of every layer.  usage: python tools/probe_camblock.py (build) | python tools/probe_camblock.py run"""
import ctypes, glob, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TID = int(os.environ.get('MV_PROBE_TID', '0'))   # the traced thread of workgroup 100 (its wave's view): 0 = wave 0, 448 = wave 7
LIBNAME = f'libcamblk_trace_t{TID}.so'
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
EV = ['layer entry', 'entry wait passed', 'entry barrier passed', 'x(2) requested, stage 0 transformed', 'stage loop done', 'h written', 'context done', 'k=3 MFMAs done', 'stores + next k=3 weights requested']


def build():
    d = '/tmp/probe_camblk'
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'camblock.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    p = os.path.join(d, 'camblock.hip')
    s = open(p).read()

    def ins(marker, ev, before=False):
        nonlocal s
        assert s.count(marker) == 1, (marker, s.count(marker))
        idx = s.index(marker)
        pos = idx if before else s.index('\n', idx) + 1
        s = s[:pos] + f'        CB_T({ev});\n' + s[pos:]
    assert s.count('constexpr int CB_THREADS = 512;') == 1
    s = s.replace('constexpr int CB_THREADS', '__device__ unsigned long long g_cb_trace[3 * 24 * 16];\n__device__ int g_cb_blk;\n'
                  '#define CB_T(ev) do { if (blockIdx.x == 100 && threadIdx.x == %d) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); '
                  'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_cb_trace[(a.blk * 24 + l) * 16 + ev] = t_; } } while (0)\nconstexpr int CB_THREADS' % TID, 1)
    s = s.replace('    int nlayers, T2, dil, seg_len;\n};', '    int nlayers, T2, dil, seg_len, blk;\n};', 1)
    s = s.replace('    a.seg_len = seg_len;\n', '    a.seg_len = seg_len;\n    static int blk_ = 0; a.blk = blk_++ % 3;\n', 1)
    ins('        const float* lbn_t = lbn_s + CB_MAX_CIN;', 0)
    ins('        __syncthreads();   // every wave\'s stores are in L2', 1, before=True)
    ins('        __syncthreads();   // every wave\'s stores are in L2', 2)
    ins('        float4v acc[2][5];', 3, before=True)
    ins('        wait_vm<0>();   // only padding transfers are left', 4, before=True)
    ins('        // ---- phase B: context gate per 100-frame segment ----', 5, before=True)
    ins('        // The next layer\'s BN1 tables (requested at the start of this tail, read by its transform() behind the entry barrier) go to LDS HERE', 6, before=True)
    ins('            const int co = ct * 16 + 4 * fg_t;', 7, before=True)
    ins('            load_wl(Ln);                       // the k = 3 weights of this layer are consumed (twelve requests)', 8)
    s = s.replace('}  // namespace mv', 'extern "C" int mv_camblk_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cb_trace), sizeof(g_cb_trace)); }\n}  // namespace mv', 1)
    # CB_T(6) sits inside `if (more)`: also stamp it for the last layer
    open(p, 'w').write(s)
    obj = d + '/cb.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-inline-asm', '-DNDEBUG', '-I', d, '-I',
                           os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/camblock.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', LIBNAME)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


def run():
    sys.path[:0] = [REPO, os.path.join(REPO, 'tests'), PKG]
    import numpy as np
    import torch
    from mvector import _hip
    _hip._lib = _hip.bind(ctypes.CDLL(os.path.join(REPO, 'tools', 'probe', LIBNAME)))
    import bench
    dev = torch.device('cuda', 0)
    featurizer, model, _ = bench.build('campp', dev)
    g = torch.Generator().manual_seed(1)
    wav = (0.1 * torch.randn([256, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(featurizer(wav))
    torch.cuda.synchronize()
    buf = np.zeros(3 * 24 * 16, dtype=np.uint64)
    lib = ctypes.CDLL(os.path.join(REPO, 'tools', 'probe', LIBNAME))
    assert lib.mv_camblk_trace_read(ctypes.c_void_p(buf.ctypes.data)) == 0
    tr = buf.reshape(3, 24, 16).astype(np.int64)[:, :, :9]
    print('cam_dense_block_kernel timeline, workgroup 100 wave 0, s_memtime ticks between events:')
    print('events: ' + ' | '.join(f'{i}={n}' for i, n in enumerate(EV)))
    for blk, n in enumerate((12, 24, 16)):
        tot = np.zeros(8)
        for l in range(n - 1):
            d = np.diff(tr[blk, l, :9])
            tot += d
            if l in (0, 1, n // 2, n - 2):
                print(f'  block {blk} layer {l:2d}: ' + ' '.join(f'{int(v):6d}' for v in d) + f'   layer {int(tr[blk, l + 1, 0] - tr[blk, l, 0]):6d}')
        print(f'  block {blk} mean   : ' + ' '.join(f'{int(v / (n - 1)):6d}' for v in tot) + f'   layer {int((tr[blk, n - 1, 0] - tr[blk, 0, 0]) / (n - 1)):6d}')


if __name__ == '__main__':
    (run if len(sys.argv) > 1 and sys.argv[1] == 'run' else build)()
