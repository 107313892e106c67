"""Upper bound of what a faster matrix pipe could give the ERes2Net family (3-pass hi/lo fp16 split = 5.3 x less
matrix time than v_mfma_f32_16x16x4_f32) WITHOUT writing the split kernels: text-edited copies of conv2d.hip in which the 1x1 kernel,
the 3x3 kernel or both issue only ONE of every four fp32 MFMAs (k4 == 0; results are wrong on purpose, loads / stores / staging
unchanged).  If the bench barely moves, the layers are bound by operand delivery and the split cannot pay.
usage: python tools/probe_conv2d.py            (build tools/probe/libconv2d_{q1x1,q3x3,qboth}.so)
       python tools/probe_conv2d.py run <lib>  (bench line of the 54.9 M ERes2NetV2 at 64 x 3 s through that library; lib = product | q1x1 | ...)"""
import ctypes, glob, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
ONE_X_ONE = ('''            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
    }
    conv2d_epilogue<NB>(a, acc, ho_u, wo_u, b, co0, j16, q);''', '''            for (int k4 = 0; k4 < 1; ++k4)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
    }
    conv2d_epilogue<NB>(a, acc, ho_u, wo_u, b, co0, j16, q);''')
THREE = ('''                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
        };''', '''                for (int k4 = 0; k4 < 1; ++k4)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
        };''')
VARIANTS = {'q1x1': [ONE_X_ONE], 'q3x3': [THREE], 'qboth': [ONE_X_ONE, THREE]}


def build(name):
    d = '/tmp/probe_conv2d/' + name
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'conv2d.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    p = os.path.join(d, 'conv2d.hip')
    s = open(p).read()
    for old, new in VARIANTS[name]:
        assert s.count(old) == 1, (name, s.count(old))
        s = s.replace(old, new)
    open(p, 'w').write(s)
    obj = d + '.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-I', d, '-I',
                           os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/conv2d.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libconv2d_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


def run(libname):
    sys.path[:0] = [REPO, PKG]
    import time
    import torch
    from mvector import _hip
    if libname != 'product':
        _hip._lib = _hip.bind(ctypes.CDLL(os.path.join(REPO, 'tools', 'probe', 'libconv2d_%s.so' % libname)))
    import bench
    dev = torch.device('cuda', 0)
    for model, B in (('eres2netv2_w96s4', 64), ('eres2netv2', 256)):
        featurizer, net, _ = bench.build(model, dev)
        g = torch.Generator().manual_seed(1)
        wav = (0.1 * torch.randn([B, bench.SAMPLES], generator=g)).clamp(-1, 1).to(dev)
        with torch.no_grad():
            feats = featurizer(wav)
            for _ in range(2):
                net(feats)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 4
            for _ in range(n):
                net(feats)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print(f'{libname:8s} {model:18s} B={B}: {dt * 1e3:8.2f} ms per forward = {B / dt:8.1f} utt/s', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        for n in VARIANTS:
            build(n)
