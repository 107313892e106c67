"""Bounded experiment on fbank_tile_kernel: the second radix-16 stage of the 256-point FFT on the fp32 matrix pipe
(v_mfma_f32_16x16x4_f32, exact fp32) beside the VALU, as TIMING variants -- text-edited copies of fbank.hip, results wrong on purpose
downstream of the stage (the pairing of the real-FFT post-processing is left as it is), instruction mix and dependencies kept:
  mfma2     after the stage-1 twiddle the wave's four frames go to LDS as today (row stride 68 instead of 65: conflict-free for both
            access patterns); per frame the 16 x 16 complex DFT along n2 is C[32 x 16] = A[32 x 32] . B[32 x 16] in real form = 16
            v_mfma_f32_16x16x4_f32 (A = cos / sin table in 16 registers, B = four 8-byte LDS reads per lane); the in-lane fft16 of
            stage 2 is gone.  An UPPER bound of the gain: the cross-row partner exchange a correct version needs is not charged.
  nostage2  stage 2 removed altogether (transpose kept): what a free, perfectly overlapped stage 2 would give.
  base      the unmodified source through the same build.
usage: python tools/probe_fbank_mfma.py                 (build tools/probe/libfbank_{mfma2,nostage2,base}.so)
       MV_PROBE_LIB=tools/probe/libfbank_mfma2.so python tools/bench_fbank.py"""
import glob, os, shutil, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')

OLD_STAGE2 = '''        // ---- the one transpose ----
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1];
        MV_WAVE_FENCE();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_read_single(tw_read + n2);
        MV_WAVE_FENCE();
        // ---- stage 2 -> z[k2] = Z[l16 + 16 k2] (halved) ----
        fft16(z);
        // ---- paired real-input post-processing ----'''
assert OLD_STAGE2.count('fft16(z);') == 1
MFMA2 = '''        // ---- the one transpose ----
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1];
        MV_WAVE_FENCE();
        // ---- PROBE: stage 2 on the matrix pipe.  B[K = 8 m + 4 c + kg][k1] = Y_c[k1][n2 = 4 m + kg] of frame f ----
        {
            const cplx* mm_read = reinterpret_cast<const cplx*>(wslot) + l16 * FBT_ROW + fs;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float4v c0 = float4v{0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const cplx y = lds_read_single(mm_read + 16 * f + 4 * m);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr_ar[2 * m], y[0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr_ai[2 * m], y[0], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr_ar[2 * m + 1], y[1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr_ai[2 * m + 1], y[1], c1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) z[4 * f + r] = cmake(c0[r], c1[r]);
            }
        }
        MV_WAVE_FENCE();
        // ---- paired real-input post-processing ----'''
NOSTAGE2 = OLD_STAGE2.replace('        fft16(z);\n', '')
OLD_CONST = '''    float* wslot = xbuf + wave * FBT_SLOT_FLOATS;
    cplx* tw_write = '''
NEW_CONST = '''    float pr_ar[8], pr_ai[8];  // PROBE: A[k2 = l16][K = 4 s + kg], s = 2 m + c, n2 = 4 m + kg: real rows (cos, sin), imaginary rows (-sin, cos)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int n2 = 4 * (s >> 1) + fs;
        float sn, cs;
        sincospif((float)((n2 * l16) & 15) * 0.125f, &sn, &cs);
        pr_ar[s] = (s & 1) ? sn : cs;
        pr_ai[s] = (s & 1) ? cs : -sn;
    }
    float* wslot = xbuf + wave * FBT_SLOT_FLOATS;
    cplx* tw_write = '''
VARIANTS = {
    'mfma2': [(OLD_STAGE2, MFMA2), (OLD_CONST, NEW_CONST), ('constexpr int FBT_ROW = 65; ', 'constexpr int FBT_ROW = 68; ')],
    'nostage2': [(OLD_STAGE2, NOSTAGE2)],
    'base': [],  # the product source through the same build (A/B control)
}


def build(name):
    d = '/tmp/probe_fbank/' + name
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'fbank.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    p = os.path.join(d, 'fbank.hip')
    s = open(p).read()
    for old, new in VARIANTS[name]:
        assert s.count(old) == 1, (name, old[:40], s.count(old))
        s = s.replace(old, new)
    open(p, 'w').write(s)
    obj = os.path.join(d, 'fbank.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-save-temps=obj', '-fno-slp-vectorize', '-fno-signed-zeros',  # fbank.hip's own `// hipcc-flags:` line
                          
                           '-I', d, '-I', os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/fbank.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libfbank_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


if __name__ == '__main__':
    for n in VARIANTS:
        build(n)
