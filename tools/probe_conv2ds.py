"""In-kernel timeline of conv2ds_kernel (text-edited copy -> tools/probe/libconv2ds_trace.so; never shipped): s_memtime of consumer wave 0 and of the
first producer wave of workgroup 37 at the boundaries of its first 48 stages.
usage: python tools/probe_conv2ds.py (build) | python tools/probe_conv2ds.py run [B]   (MV_BENCH_SHAPES selects the layers of tools/bench_conv2d.py)"""
import ctypes, glob, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
NST = 48


def build():
    d = '/tmp/probe_conv2ds'
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'arch'))
    for f in glob.glob(os.path.join(PKG, 'csrc', '*.h')) + [os.path.join(PKG, 'csrc', 'conv2ds.hip')]:
        shutil.copy(f, d)
    shutil.copy(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h'), os.path.join(d, 'arch'))
    p = os.path.join(d, 'common.h')
    t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
    open(p, 'w').write(t)
    p = os.path.join(d, 'conv2ds.hip')
    s = open(p).read()

    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (a[:70], s.count(a))
        s = s.replace(a, b)
    rep('constexpr int CS_SEGS = 8;', '__device__ unsigned long long g_cs_trace[2 * %d * 4];\n__device__ unsigned long long g_cs_wg[1024][2];\n'
        '#define CS_T(role, st, ev) do { if (blockIdx.x == 37 && lane == 0 && (st) < %d) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); '
        'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_cs_trace[((role) * %d + (st)) * 4 + (ev)] = t_; } } while (0)\nconstexpr int CS_SEGS = 8;' % (NST, NST, NST))
    # producer (first producer wave only): before the wait, after the wait, after the barrier, after the issue
    rep('            wait_vm_dyn((issued - gs - 1) * pp);\n            lds_barrier();\n            if (issued < nstages) {\n                issue_stage();\n                ++issued;\n            }\n',
        '            if (pw == 0) CS_T(1, gs, 0);\n            wait_vm_dyn((issued - gs - 1) * pp);\n            if (pw == 0) CS_T(1, gs, 1);\n            lds_barrier();\n            if (pw == 0) CS_T(1, gs, 2);\n'
        '            if (issued < nstages) {\n                issue_stage();\n                ++issued;\n            }\n            if (pw == 0) CS_T(1, gs, 3);\n')
    # consumer wave 0: before the barrier, after it, after the MFMAs, after the epilogue (tile ends only)
    rep('        lds_barrier();  // stage gs has landed; every wave has left the slot the producers refill next\n',
        '        if (wave == 0) CS_T(0, gs, 0);\n        lds_barrier();  // stage gs has landed; every wave has left the slot the producers refill next\n        if (wave == 0) CS_T(0, gs, 1);\n')
    rep('        if (++c < nst) continue;\n', '        if (wave == 0) CS_T(0, gs, 2);\n        if (++c < nst) continue;\n')
    rep("                        if (ok[u]) *reinterpret_cast<uint4v*>(a.y2 + pixo[u] * a.ldy2 * 2 + coff) = w2;\n                    }\n                }\n        }\n    }\n    if (track) s16_peak_commit(a.peak, pk);",
        "                        if (ok[u]) *reinterpret_cast<uint4v*>(a.y2 + pixo[u] * a.ldy2 * 2 + coff) = w2;\n                    }\n                }\n        }\n        if (wave == 0) CS_T(0, gs, 3);\n    }\n"
        "    if (tid == 0) g_cs_wg[blockIdx.x & 1023][1] = __builtin_amdgcn_s_memtime();\n    if (track) s16_peak_commit(a.peak, pk);")
    rep("    const int HWo = a.Ho * a.Wo;\n\n    if (wave >= a.ncons) {", "    const int HWo = a.Ho * a.Wo;\n    if (tid == 0) g_cs_wg[blockIdx.x & 1023][0] = __builtin_amdgcn_s_memtime();\n\n    if (wave >= a.ncons) {")
    rep('}  // namespace mv\n\nextern "C" {', 'extern "C" int mv_conv2ds_occupancy(int ks, int nbw, int threads, int lds) { int n = -1; hipError_t e;\n'
        '  if (ks == 3 && nbw == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv2ds_kernel<3, 1, 8, 768, false>, threads, (size_t)lds);\n'
        '  else if (ks == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv2ds_kernel<3, 2, 8, 512, false>, threads, (size_t)lds);\n'
        '  else if (nbw == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv2ds_kernel<1, 1, 8, 768, false>, threads, (size_t)lds);\n'
        '  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv2ds_kernel<1, 2, 8, 512, false>, threads, (size_t)lds);\n  return e == hipSuccess ? n : -(int)e; }\n'
        'extern "C" int mv_conv2ds_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cs_trace), sizeof(g_cs_trace)); }\n'
        'extern "C" int mv_conv2ds_wg_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cs_wg), sizeof(g_cs_wg)); }\n'
        'extern "C" int mv_conv2ds_trace_clear() { static unsigned long long z[2 * %d * 4]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cs_trace), z, sizeof(z)); }\n}  // namespace mv\n\nextern "C" {' % NST)
    open(p, 'w').write(s)
    obj = d + '/cs.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-inline-asm', '-DNDEBUG', '-I', d, '-I',
                           os.path.join(PKG, 'csrc'), '-x', 'hip', '-c', p, '-o', obj])
    objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/conv2ds.hip.o')]
    out = os.path.join(REPO, 'tools', 'probe', 'libconv2ds_trace.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
    print('built', out)


def run():
    sys.path[:0] = [REPO, PKG]
    import torch
    from mvector import _hip
    lib = ctypes.CDLL(os.path.join(REPO, 'tools', 'probe', 'libconv2ds_trace.so'))
    cdll = _hip.bind(lib)
    _hip._lib = cdll
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    for ks, nbw, threads, lds_ in [(3, 1, 320, 63488), (3, 1, 320, 40000), (3, 1, 256, 63488), (3, 1, 384, 63488), (3, 1, 448, 63488), (3, 2, 448, 63488), (1, 2, 512, 149504), (1, 1, 320, 63488)]:
        print('occupancy: ks %d nbw %d threads %d lds %d -> %d workgroups per CU' % (ks, nbw, threads, lds_, lib.mv_conv2ds_occupancy(ks, nbw, threads, lds_)))
    import importlib.util
    spec = importlib.util.spec_from_file_location('bc', os.path.join(REPO, 'tools', 'bench_conv2d.py'))
    src = open(os.path.join(REPO, 'tools', 'bench_conv2d.py')).read()
    # one launch of the split form per layer, then the timeline
    st = lambda: _hip.current_stream(torch.empty(1, device='cuda'))
    ns = {'__file__': os.path.join(REPO, 'tools', 'bench_conv2d.py')}
    exec(src.split('for name, H, W, cin, cout, ks, stride, with_res in SHAPES:')[0].replace("B = int(sys.argv[1]) if len(sys.argv) > 1 else 16", "B = %d" % B), ns)
    for name, H, W, cin, cout, ks, stride, with_res in ns['SHAPES']:
        g = torch.Generator().manual_seed(1)
        p = ks // 2
        Ho, Wo = (H + 2 * p - ks) // stride + 1, (W + 2 * p - ks) // stride + 1
        x = torch.randn(B, H, W, cin, generator=g).clamp(0, 20).cuda()
        w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
        bias = torch.zeros(cout).cuda()
        res = torch.randn(B, Ho, Wo, cout, generator=g).cuda() if with_res else None
        sp = lambda t: None if t is None else (lambda o: (_hip.check(cdll.mv_map_split_f32(t.data_ptr(), o.data_ptr(), t.numel(), st()), cdll), o)[1])(torch.empty_like(t))
        xs, rs = sp(x), sp(res)
        ys = torch.empty(B, Ho, Wo, cout, device='cuda')
        pks = torch.zeros(cdll.mv_conv2ds_packed_elems(cout, cin, ks), device='cuda')
        osc = ctypes.c_float(0)
        _hip.check(cdll.mv_conv2ds_pack_weight(w.data_ptr(), None, cout, cin, ks, pks.data_ptr(), ctypes.byref(osc), st()), cdll)
        e = _hip.MvConv2dsDesc()
        e.x, e.ldx, e.w, e.bias, e.oscale, e.y, e.ldy = xs.data_ptr(), cin, pks.data_ptr(), bias.data_ptr(), osc.value, ys.data_ptr(), cout
        e.res, e.ldres = (rs.data_ptr() if with_res else None), cout
        e.B, e.H, e.W, e.cin16, e.cout16, e.ks, e.stride, e.epi, e.lo, e.hi = B, H, W, cin, cout, ks, stride, 0, 0.0, 20.0
        for k in ('nbw', 'rows', 'ring', 'wgs', 'spw', 'nprod'):
            setattr(e, k + '_hint', int(os.environ.get('MV_PROBE_' + k.upper(), '0')))
        for _ in range(2):
            _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)
        torch.cuda.synchronize()
        lib.mv_conv2ds_trace_clear()
        _hip.check(cdll.mv_conv2ds_forward(ctypes.byref(e), st()), cdll)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (2 * NST * 4))()
        lib.mv_conv2ds_trace_read(buf)
        wg = (ctypes.c_ulonglong * 2048)()
        lib.mv_conv2ds_wg_read(wg)
        starts = sorted(wg[2 * i] for i in range(1024) if wg[2 * i + 1])
        ends = sorted(wg[2 * i + 1] for i in range(1024) if wg[2 * i + 1])
        if starts:
            k0 = starts[0]
            q = lambda a, f: (a[min(len(a) - 1, int(f * len(a)))] - k0) / 100.0
            print('   workgroups %d: start min / median / max %.1f / %.1f / %.1f, end min / median / max %.1f / %.1f / %.1f (same units, from the first start)' %
                  (len(starts), q(starts, 0), q(starts, 0.5), q(starts, 0.9999), q(ends, 0), q(ends, 0.5), q(ends, 0.9999)))
        t0 = min(v for v in buf if v)
        us = lambda v: '%7.2f' % ((v - t0) / 100.0) if v else '      -'    # unit: 100 ticks of s_memtime (r12q: the kernel's 167 us are 412 k ticks)
        print('== %s (B=%d): consumer wave 0 [arrive, barrier passed, MFMAs done, epilogue done] | producer 0 [before wait, landed, barrier passed, issued]  (units of 100 s_memtime ticks ~ 0.04 us)' % (name, B))
        for sidx in range(NST):
            c = [buf[(0 * NST + sidx) * 4 + k] for k in range(4)]
            pr = [buf[(1 * NST + sidx) * 4 + k] for k in range(4)]
            if not any(c) and not any(pr):
                break
            print('  stage %2d  C %s   P %s' % (sidx, ' '.join(us(v) for v in c), ' '.join(us(v) for v in pr)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'run':
        run()
    else:
        build()
