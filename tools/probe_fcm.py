"""Timing-probe builds of the fused FCM block kernel WITHOUT probe code in the product sources: every variant is a text-edited copy
of fcmblock.hip / arch/gfx950.h under /tmp, compiled and linked with the product's other objects into
tools/probe/libfcm_<name>.so (never shipped; results of the probe kernels are wrong on purpose).  usage: python tools/probe_fcm.py [names]"""
import glob, os, re, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
W = '/tmp/probe_fcm'
HIPCC = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG']


def re_sub(pat, rep, count=0):
    return lambda s: re.subn(pat, rep, s, count=count, flags=re.S)


def lit(old, new):
    return lambda s: (s.replace(old, new), s.count(old))


MFMA = r'v_mfma_f32_16x16x32_f16 %[0-9]+, %[0-9]+, %[0-9]+, %[0-9]+'
VARIANTS = {
    # name: {file: [edit, ...]}
    'nomfma': {'arch/gfx950.h': [re_sub(MFMA, 's_nop 0')]},
    'nolds': {'arch/gfx950.h': [re_sub(r'ds_read_b128 %[0-9]+, %[0-9]+( offset:[0-9]+)?(?=[\\"])', 's_nop 0')]},
    'nostore': {'fcmblock.hip': [lit('if (16 * k < jlim) *MV_AS_GLOBAL(half4v, yrow + k * tile_bytes + ylane) = o;', '(void)o;')]},
    'noload': {'fcmblock.hip': [lit('auto issue_part = [&](const RowReq& r, int i) __attribute__((always_inline)) {', 'auto issue_part = [&](const RowReq& r, int i) __attribute__((always_inline)) {\n            if (i >= 0) return;')]},
    # in-kernel timeline: s_memtime at four points of a producer / consumer step of workgroup 100 (tools/bench_fcm.py prints it)
    'trace': {'fcmblock.hip': [
        lit('namespace mv {\n\nconstexpr int FBK_C = 32;', 'namespace mv {\n__device__ unsigned long long g_fbk_trace[8 * 48 * 4];\n'
            '#define FBK_T(ev) do { if (blockIdx.x == 100 && lane == 0 && i < 48) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); '
            'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_fbk_trace[(wave8 * 48 + i) * 4 + ev] = t_; } } while (0)\n'
            'constexpr int FBK_C = 32;'),
        lit('            lds_barrier();        // ... in every wave; the consumers', '            lds_barrier(); FBK_T(0);       // ... in every wave; the consumers'),
        lit('            for (int s = 0; s < SF; ++s) req[s] = next_row();\n', '            for (int s = 0; s < SF; ++s) req[s] = next_row();\n            FBK_T(1);\n'),
        lit('            mfma_hazard_pad();\n            mfma_hazard_pad();\n            // mid row i', '            FBK_T(2);\n            mfma_hazard_pad();\n            mfma_hazard_pad();\n            // mid row i'),
        lit('            cslot += SF;\n            cslot = cslot >= G::RING ? cslot - G::RING : cslot;\n        }\n        lds_barrier();  // the consumers', '            FBK_T(3);\n            cslot += SF;\n            cslot = cslot >= G::RING ? cslot - G::RING : cslot;\n        }\n        lds_barrier();  // the consumers'),
        lit('        lds_barrier();\n        half8v bq[2][NT];', '        lds_barrier(); FBK_T(0);\n        half8v bq[2][NT];'),
        lit('        const bool sc = i < nsteps, sm = i > 0;  // uniform', '        FBK_T(1);\n        const bool sc = i < nsteps, sm = i > 0;  // uniform'),
        lit('        if (sm) {\n#pragma unroll\n            for (int dt = 0; dt < 3; ++dt)', '        FBK_T(2);\n        if (sm) {\n#pragma unroll\n            for (int dt = 0; dt < 3; ++dt)'),
        lit('        cslot += SF;\n        cslot = cslot >= G::RING ? cslot - G::RING : cslot;\n    };', '        FBK_T(3);\n        cslot += SF;\n        cslot = cslot >= G::RING ? cslot - G::RING : cslot;\n    };'),
        lit('extern "C" {', 'extern "C" {\nint mv_fcm_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mv::g_fbk_trace), sizeof(mv::g_fbk_trace)); }'),
    ]},
    # one group reduced to its barriers (and, for the producers, the row transfers): the other group's path alone
    'prodonly': {'fcmblock.hip': [lit('        lds_barrier();\n        half8v bq[2][NT];', '        lds_barrier();\n        if (i >= 0) return;\n        half8v bq[2][NT];')]},
    'consonly': {'fcmblock.hip': [lit('            float4v acc1[2][NT];', '            if (i >= 0) continue;\n            float4v acc1[2][NT];')]},
    # no barrier coupling between the groups (races on the mid slots: timing only)
    'nobarrier': {'fcmblock.hip': [lit('lds_barrier();', 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')]},
}


def build(name):
    d = os.path.join(W, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'fcmblock.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    text = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(text)
    for f, edits in VARIANTS[name].items():
        p = os.path.join(d, f)
        s = open(p).read()
        for e in edits:
            s, n = e(s)
            assert n > 0, (name, f, 'edit did not apply')
        open(p, 'w').write(s)
    obj = os.path.join(W, name + '.o')
    subprocess.check_call(HIPCC + ['-I', d, '-I', os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', os.path.join(d, 'fcmblock.hip'), '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/fcmblock.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libfcm_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


if __name__ == '__main__':
    from concurrent.futures import ThreadPoolExecutor
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(build, names))
