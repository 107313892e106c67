"""Occupancy variants of fbank_tile_kernel as TEXT-EDITED copies -- 12 / 16 waves per workgroup instead of 8:
  * the per-wave LDS slot shrinks from 8320 to 4672 bytes: the 16 x 16 transpose goes through it one component (re, then im) at a time
    ([16 rows][4 frames x 16 + 1] floats), the power rows keep their layout;
  * no second register set for prefetched samples (the quad's samples are loaded at the top of the iteration; more waves hide the latency);
  * the mel weights (40 registers) come from global memory (L1) at use, the split twiddles (16 registers) from an LDS table.
usage: python tools/fbank_waves_variant.py [name:waves ...]   -> tools/probe/libfbankw_<name>.so + register / spill counts
CORRECT results (the same arithmetic; the time mean is summed over `waves` slots, so its last bits differ from the product's)."""
import glob, os, re, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')


def make(name, waves, lean=True, bounds=None):
    d = '/tmp/fbw/' + name
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d + '/arch')
    for f in glob.glob(PKG + '/csrc/*.h') + [PKG + '/csrc/fbank.hip']:
        shutil.copy(f, d)
    shutil.copy(PKG + '/csrc/arch/gfx950.h', d + '/arch')
    p = d + '/common.h'
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    s = open(d + '/fbank.hip').read()

    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (name, old[:60], s.count(old))
        s = s.replace(old, new)
    rep('constexpr int FBT_WAVES = 8;', 'constexpr int FBT_WAVES = %d;' % waves)
    if bounds:
        rep('__global__ __launch_bounds__(FBT_WAVES * 64) void fbank_tile_kernel', '__global__ __launch_bounds__(FBT_WAVES * 64, %d) void fbank_tile_kernel' % bounds)
    if lean:
        rep('constexpr int FBT_SLOT_FLOATS = 16 * FBT_ROW * 2;  // 2080 floats = 8320 B per wave',
            'constexpr int FBT_SLOT_FLOATS = 4 * 292;  // power rows; the transpose goes through it one component at a time')
        rep('    float* tile = ltw1 + 512;                                      // [tile_rows][nbins]',
            '    float* lct2 = ltw1 + 512;                                      // [256][2] cos, sin of pi k / 256\n    float* tile = lct2 + 512;                                      // [tile_rows][nbins]')
        rep('''    float2v ctw2[8];        // (cos, sin) of pi k / 256 at k = l16 + 16 j
#pragma unroll
    for (int j = 0; j < 8; ++j) ctw2[j] = *reinterpret_cast<const float2v*>(a.tab.tw512 + 2 * (l16 + 16 * j));
    float4v mb0[G0], mb1[G1];  // mel weights of this lane's (block, filter): 4 bins per group
#pragma unroll
    for (int g = 0; g < G0; ++g) mb0[g] = *reinterpret_cast<const float4v*>(a.tab.melb + (size_t)g * 256 + lane * 4);
#pragma unroll
    for (int g = 0; g < G1; ++g) mb1[g] = *reinterpret_cast<const float4v*>(a.tab.melb + (size_t)(G0 + g) * 256 + lane * 4);
''', '''    for (int i = tid; i < 512; i += THREADS) lct2[i] = a.tab.tw512[i];
    const float* cct2 = lct2 + 2 * l16;   // (cos, sin) of pi k / 256 at k = l16 + 16 j: + 32 j
    const float* melw = a.tab.melb + lane * 4;   // this lane's mel weights, group g at + 256 g (L1 resident)
''')
        rep('''    cplx* tw_write = reinterpret_cast<cplx*>(wslot) + lane;                       // element (k1, frame fs, n2 = l16) at + k1 * FBT_ROW
    const cplx* tw_read = reinterpret_cast<const cplx*>(wslot) + l16 * FBT_ROW + 16 * fs;  // element (k1 = l16, fs, n2) at + n2
''', '''    float* tw_write = wslot + lane;                            // component of element (k1, frame fs, n2 = l16) at + k1 * FBT_ROW
    const float* tw_read = wslot + l16 * FBT_ROW + 16 * fs;    // component of element (k1 = l16, fs, n2) at + n2
''')
        rep('''        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1];
        MV_WAVE_FENCE();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_read_single(tw_read + n2);
        MV_WAVE_FENCE();
''', '''        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1][0];
        MV_WAVE_FENCE();
        float zre[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) zre[n2] = *(const volatile __attribute__((address_space(3))) float*)(tw_read + n2);
        MV_WAVE_FENCE();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1][1];
        MV_WAVE_FENCE();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = cmake(zre[n2], *(const volatile __attribute__((address_space(3))) float*)(tw_read + n2));
        MV_WAVE_FENCE();
''')
        rep('            const float c = ctw2[j][0], s = ctw2[j][1];      // w = c - i s',
            '            const float2v cs2 = lds_load_unmerged(reinterpret_cast<const float2v*>(cct2 + 32 * j));\n            const float c = cs2[0], s = cs2[1];      // w = c - i s')
        rep('            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);\n',
            '            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);\n            const float4v mw = *reinterpret_cast<const float4v*>(melw + 256 * g);\n')
        rep('acc0[c] = fb_mfma4(av[c], mb0[g][c], g == 0 ? zero4 : acc0[c]);', 'acc0[c] = fb_mfma4(av[c], mw[c], g == 0 ? zero4 : acc0[c]);')
        rep('            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);\n',
            '            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);\n            const float4v mw = *reinterpret_cast<const float4v*>(melw + 256 * (G0 + g));\n')
        rep('acc1[c] = fb_mfma4(av[c], mb1[g][c], g == 0 ? zero4 : acc1[c]);', 'acc1[c] = fb_mfma4(av[c], mw[c], g == 0 ? zero4 : acc1[c]);')
        rep('        if (q + FBT_WAVES < nquads) load_quad(q + FBT_WAVES, r_next);\n', '')
        rep('''    float3u ra[NG], rb[NG];
    if (qbeg + wave < nquads) load_quad(qbeg + wave, ra);
    for (int q = qbeg + wave; q < nquads; q += 2 * FBT_WAVES) {
        process_quad(q, ra, rb);
        if (q + FBT_WAVES < nquads) process_quad(q + FBT_WAVES, rb, ra);
    }''', '''    float3u ra[NG];
    for (int q = qbeg + wave; q < nquads; q += FBT_WAVES) {
        load_quad(q, ra);
        process_quad(q, ra, ra);
    }''')
        rep('    return ((size_t)mv::FBT_WAVES * mv::FBT_SLOT_FLOATS + mv::FBT_WIN_FLOATS + 512) * sizeof(float);',
            '    return ((size_t)mv::FBT_WAVES * mv::FBT_SLOT_FLOATS + mv::FBT_WIN_FLOATS + 512 + 512) * sizeof(float);')
    open(d + '/fbank.hip', 'w').write(s)
    obj = d + '/fbank.o'
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-fno-slp-vectorize',
                        '-fno-signed-zeros', '-save-temps=obj', '-I', d, '-I', PKG + '/csrc', '-x', 'hip', '-c', d + '/fbank.hip', '-o', obj], capture_output=True, text=True)
    if r.returncode:
        print(name, 'FAILED', r.stderr[-1500:])
        return
    asm = open(glob.glob(d + '/*gfx950.s')[0]).read()
    res = dict(re.findall(r'\.set _ZN2mv17fbank_tile_kernelILi13ELb1ELi7ELi3EEEvNS_9FbankArgsE\.(num_vgpr|private_seg_size), (\d+)', asm))
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/fbank.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libfbankw_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print(f'{name}: {waves} waves, VGPRs {res.get("num_vgpr")}, scratch bytes per lane {res.get("private_seg_size")} -> {out}')


if __name__ == '__main__':
    for spec in (sys.argv[1:] or ['w8:8', 'w12:12', 'w16:16']):
        n, w = spec.split(':')
        make(n, int(w))
