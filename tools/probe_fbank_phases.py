"""Where do fbank_tile_kernel's 48 us go?  Timing variants (text-edited copies of fbank.hip; results wrong on purpose unless noted) and an
in-kernel s_memtime timeline, for the occupancy question of VERDICT r3 item 5.  Nothing here is shipped.
  base      the product source through the same build (A/B control)
  trace     s_memtime at 9 points of a quad iteration (workgroup 100, wave 0 and wave 5) -> `python tools/probe_fbank_phases.py run`
  noload    every quad reads the utterance's FIRST frames (L1 / L2 hits): what the HBM side of the sample loads costs
  nomel     the mel MFMAs and their ten LDS operand reads removed
  nofft     both in-register fft16 removed (transposes, twiddles, post-processing kept)
  notile    log-mel rows not kept in LDS / not written (no tile stores, no CMN sweep)
  occ4      __launch_bounds__(512, 4): <= 128 VGPRs (spills to scratch) and no LDS block (tile_rows = 0, two-pass CMN through global memory)
            -> two workgroups = 16 waves per CU.  CORRECT results.  A pessimistic probe of "4 waves per SIMD".
usage: python tools/probe_fbank_phases.py            build tools/probe/libfbankp_<name>.so
       MV_PROBE_LIB=tools/probe/libfbankp_noload.so python tools/bench_fbank.py
       python tools/probe_fbank_phases.py run        print the timeline of the trace build"""
import ctypes, glob, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')

EV = ['entry', 'samples landed + DC/pre-emphasis/window', 'stage-1 fft16 + twiddle', 'transpose (write + read back)', 'stage-2 fft16',
      'post-processing + power rows written', 'mel MFMAs', 'log + sums + tile stores']
TRACE_DEF = ('namespace mv {\n__device__ unsigned long long g_fb_trace[2 * 16 * 9];\n'
             '#define FB_T(ev) do { if (blockIdx.x == 100 && (wave == 0 || wave == 5) && fb_it < 16) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); '
             'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) g_fb_trace[((wave == 5) * 16 + fb_it) * 9 + ev] = t_; } } while (0)\n')


def edits(name):
    E = []
    if name == 'trace':
        E += [('namespace mv {\n', TRACE_DEF, 'first'),
              ("    float csum0 = 0.0f, csum1 = 0.0f;   // this wave's column sums (slot", "    int fb_it = 0;\n    float csum0 = 0.0f, csum1 = 0.0f;   // this wave's column sums (slot", 'only'),
              ('        float x0[NG], x1[NG];\n#pragma unroll\n        for (int n1 = 0; n1 < NG; ++n1) {\n            const int idx = 32 * n1 + 2 * l16;\n            const bool full = NG == 13 ? n1 < 12 : 32 * n1 + 32 <= a.win;\n            x0[n1] = full',
               '        FB_T(0);\n        float x0[NG], x1[NG];\n#pragma unroll\n        for (int n1 = 0; n1 < NG; ++n1) {\n            const int idx = 32 * n1 + 2 * l16;\n            const bool full = NG == 13 ? n1 < 12 : 32 * n1 + 32 <= a.win;\n            x0[n1] = full', 'last'),
              ('        if (q + FBT_WAVES < nquads) load_quad(q + FBT_WAVES, r_next);\n', '        FB_T(1);\n        if (q + FBT_WAVES < nquads) load_quad(q + FBT_WAVES, r_next);\n', 'only'),
              ('        // ---- the one transpose ----\n', '        FB_T(2);\n        // ---- the one transpose ----\n', 'last'),
              ('        // ---- stage 2 -> z[k2] = Z[l16 + 16 k2] (halved) ----\n        fft16(z);\n', '        FB_T(3);\n        // ---- stage 2 -> z[k2] = Z[l16 + 16 k2] (halved) ----\n        fft16(z);\n        FB_T(4);\n', 'only'),
              ('        // ---- mel filters on the matrix pipe (weights in registers) ----\n', '        FB_T(5);\n        // ---- mel filters on the matrix pipe (weights in registers) ----\n', 'only'),
              ('        MV_WAVE_FENCE();  // the power rows are consumed: the next quad\'s transpose may overwrite them\n        // pass 1:',
               '        MV_WAVE_FENCE();  // the power rows are consumed: the next quad\'s transpose may overwrite them\n        FB_T(6);\n        // pass 1:', 'only'),
              ('    float3u ra[NG], rb[NG];\n    if (qbeg + wave < nquads) load_quad(qbeg + wave, ra);\n    for (int q = qbeg + wave; q < nquads; q += 2 * FBT_WAVES) {\n        process_quad(q, ra, rb);\n        if (q + FBT_WAVES < nquads) process_quad(q + FBT_WAVES, rb, ra);\n    }',
               '    float3u ra[NG], rb[NG];\n    if (qbeg + wave < nquads) load_quad(qbeg + wave, ra);\n    for (int q = qbeg + wave; q < nquads; q += 2 * FBT_WAVES) {\n        process_quad(q, ra, rb);\n        FB_T(7);\n        ++fb_it;\n        if (q + FBT_WAVES < nquads) {\n            process_quad(q + FBT_WAVES, rb, ra);\n            FB_T(7);\n            ++fb_it;\n        }\n    }\n    FB_T(8);', 'only'),
              ('}  // namespace mv\n\n// ------------------------------------------------------------------------------------------ host side',
               'extern "C" int mv_fbank_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fb_trace), sizeof(g_fb_trace)); }\n}  // namespace mv\n\n// ------------------------------------------------------------------------------------------ host side', 'only')]
    elif name == 'noload':
        E += [('        const int f = f_raw < T ? f_raw : T - 1;  // surplus slots recompute the last frame; nothing of theirs is kept\n',
               '        const int f = fs + 0 * (f_raw < T ? f_raw : T - 1);  // PROBE: always the first four frames\n', 'last')]
    elif name == 'nomel':
        E += [('            for (int c = 0; c < 4; ++c) acc0[c] = fb_mfma4(av[c], mb0[g][c], g == 0 ? zero4 : acc0[c]);',
               '            for (int c = 0; c < 4; ++c) acc0[c] = (g == 0 ? zero4 : acc0[c]) + float4v{pk[c], pk[c + 4], pp[c], pp[c + 4]};  // PROBE: no MFMA, no operand read', 'only'),
              ('            for (int c = 0; c < 4; ++c) acc1[c] = fb_mfma4(av[c], mb1[g][c], g == 0 ? zero4 : acc1[c]);',
               '            for (int c = 0; c < 4; ++c) acc1[c] = (g == 0 ? zero4 : acc1[c]) + float4v{pp[c], pk[c + 4], pk[c], pp[c + 4]};  // PROBE', 'only'),
              ('            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);\n', '', 'only'),
              ('            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);\n', '', 'only')]
    elif name == 'nofft':
        E += [('        // ---- stage 1 + twiddle ----\n        fft16(z);\n', '        // ---- stage 1 + twiddle ----\n', 'last'),
              ('        // ---- stage 2 -> z[k2] = Z[l16 + 16 k2] (halved) ----\n        fft16(z);\n', '', 'only')]
    elif name == 'notile':
        E += [('        if (q * 4 < tile_rows) {  // uniform (tile_rows is a multiple of 4): LDS block [t][m]',
               '        if (q < 0) {  // PROBE: no tile stores', 'only'),
              ('        } else {\n            auto d0 = MV_AS_GLOBAL(float, orow + row0 + m0);', '        } else if (q < -1) {\n            auto d0 = MV_AS_GLOBAL(float, orow + row0 + m0);', 'only'),
              ('    const bool second_pass = a.cmn || a.lens_ratio != nullptr || a.num_samples != nullptr;\n    if (!second_pass && tile_rows == 0) return;\n',
               '    if (csum0 + csum1 != 12345.0f) return;  // PROBE: no CMN sweep\n    const bool second_pass = a.cmn || a.lens_ratio != nullptr || a.num_samples != nullptr;\n', 'only')]
    elif name == 'noloop':
        E += [('    for (int q = qbeg + wave; q < nquads; q += 2 * FBT_WAVES) {\n        process_quad(q, ra, rb);', '    for (int q = qbeg + wave; q < 0; q += 2 * FBT_WAVES) {\n        process_quad(q, ra, rb);', 'only')]
    elif name == 'notrans':
        E += [('        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1];\n        MV_WAVE_FENCE();\n#pragma unroll\n        for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_read_single(tw_read + n2);\n',
               '        for (int k1 = 0; k1 < 0; ++k1) tw_write[k1 * FBT_ROW] = z[k1];\n        MV_WAVE_FENCE();\n', 'only')]
    elif name == 'nopost':
        E += [('            const float t = dpp_mov_all<DPP_ROW_MIRROR>(src[c]);\n                bp[c] = dpp_mov<DPP_ROW_SHR1>(own_alt[c], t);', '            bp[c] = src[c] + own_alt[c];', 'only')]
    elif name == 'nowin':
        E += [('            const float2v w2 = lds_load_unmerged(reinterpret_cast<const float2v*>(cwin + 32 * n1));\n            const float y0 = fmaf(npre, r[n1][0], x0[n1]) - dc;',
               '            const float2v w2 = float2v{0.5f, 0.25f + 0.01f * n1};\n            const float y0 = fmaf(npre, r[n1][0], x0[n1]) - dc;', 'last')]
    elif name == 'notw':
        E += [('            const float2v tw = lds_load_unmerged(reinterpret_cast<const float2v*>(ctw1 + 32 * k1));\n            z[k1] = cmul_conjtw(z[k1], tw[0], tw[1]);',
               '            const float2v tw = float2v{0.7f, 0.1f * k1};\n            z[k1] = cmul_conjtw(z[k1], tw[0], tw[1]);', 'last')]
    elif name == 'nolog':
        E += [('            v0[r] = fb_log2(fmaxf(a0[r], 1.1920928955078125e-07f)) * 0.69314718055994531f;\n            v1[r] = fb_log2(fmaxf(a1[r], 1.1920928955078125e-07f)) * 0.69314718055994531f;',
               '            v0[r] = fmaxf(a0[r], 1.1920928955078125e-07f) * 0.69314718055994531f;\n            v1[r] = fmaxf(a1[r], 1.1920928955078125e-07f) * 0.69314718055994531f;', 'last')]
    elif name == 'nopower':
        E += [('        for (int j = 0; j < 8; ++j) p_own[16 * j] = pk[j];\n        p_par0[16 * 15] = pp[0];\n#pragma unroll\n        for (int j = 1; j < 8; ++j) p_par[16 * (15 - j)] = pp[j];\n',
               '        for (int j = 0; j < 1; ++j) p_own[16 * j] = pk[0] + pk[1] + pk[2] + pk[3] + pk[4] + pk[5] + pk[6] + pk[7] + pp[0] + pp[1] + pp[2] + pp[3] + pp[4] + pp[5] + pp[6] + pp[7];\n', 'only')]
    elif name == 'occ4':
        E += [('__global__ __launch_bounds__(FBT_WAVES * 64) void fbank_tile_kernel', '__global__ __launch_bounds__(FBT_WAVES * 64, 4) void fbank_tile_kernel', 'only'),
              ('            a.tile_rows = (int)(plan.fit < plan.need ? plan.fit : plan.need);\n', '            a.tile_rows = 0;  // PROBE: no LDS block -> 70 KB per workgroup, two workgroups per CU\n', 'only')]
    return E


def build(name):
    d = '/tmp/probe_fbankp/' + name
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
    for old, new, which in edits(name):
        n = s.count(old)
        assert n >= 1 and (which != 'only' or n == 1), (name, old[:60], n)
        i = s.index(old) if which in ('only', 'first') else s.rindex(old)
        s = s[:i] + new + s[i + len(old):]
    open(p, 'w').write(s)
    obj = os.path.join(d, 'fbank.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-fno-slp-vectorize',
                           '-fno-signed-zeros', '-I', d, '-I', os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/fbank.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libfbankp_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


def run():
    sys.path[:0] = [REPO, PKG]
    import numpy as np
    import torch
    from mvector import _hip
    path = os.path.join(REPO, 'tools', 'probe', 'libfbankp_trace.so')
    cdll = _hip.bind_partial(ctypes.CDLL(path))
    fb = _hip.Fbank(dict(sample_frequency=16000, num_mel_bins=80), cdll=cdll)
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn([256, 48000], generator=g)).clamp(-1, 1).cuda()
    for _ in range(3):
        fb(wav)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fb(wav)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(2 * 16 * 9, dtype=np.uint64)
    assert ctypes.CDLL(path).mv_fbank_trace_read(ctypes.c_void_p(buf.ctypes.data)) == 0
    tr = buf.reshape(2, 16, 9).astype(np.int64)
    print(f'fbank_tile_kernel timeline (workgroup 100; launch {e0.elapsed_time(e1) * 1e3:.1f} us with the probes in), s_memtime ticks per phase of a quad iteration:')
    for wi, w in enumerate((0, 5)):
        its = [i for i in range(16) if tr[wi, i, 7] > 0]
        print(f' wave {w}: {len(its)} iterations, first entry -> last end {tr[wi, its[-1], 7] - tr[wi, 0, 0]} ticks')
        for i in its:
            d = np.diff(tr[wi, i, :8])
            gap = tr[wi, i, 0] - tr[wi, i - 1, 7] if i else 0
            print(f'  it {i:2d}: ' + ' '.join(f'{int(x):6d}' for x in d) + f'   total {int(tr[wi, i, 7] - tr[wi, i, 0]):6d}  gap-before {int(gap):5d}')
        dd = np.array([np.diff(tr[wi, i, :8]) for i in its[1:-1]])
        if len(dd):
            print('  mean : ' + ' '.join(f'{x:6.0f}' for x in dd.mean(0)))
    print(' phases: ' + ' | '.join(f'{i}->{i + 1} {EV[i + 1]}' for i in range(7)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'run':
        run()
    else:
        names = sys.argv[1:] or ['base', 'trace', 'noload', 'nomel', 'nofft', 'notile', 'noloop', 'notrans', 'nopost', 'nowin', 'notw', 'nolog', 'nopower']
        for n in names:
            build(n)
