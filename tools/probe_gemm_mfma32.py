"""TIMING probe for VERDICT r3 item 4: the ring GEMM's K stage on v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_16x16x32_f16.
A text-edited copy of arch/gfx950.h + conv1d.hip (tools/probe/libconv_mfma32.so; never shipped, RESULTS ARE WRONG ON PURPOSE): in
conv1d_ring_persistent_kernel every group of eight 16x16x32 MFMAs of a K-stage step (two A fragments x four B fragments into eight 4-register
accumulators = 128 matrix-pipe cycles) becomes four 32x32x16 MFMAs on the same operand registers into two 16-register accumulators
(4 x 32 = 128 cycles): the same fragment reads, the same LDS traffic, the same accumulator footprint (eight 16-register tiles per wave), the
same FLOPs, half the MFMA issue slots -- what a correct 32x32 variant (other fragment layout, other epilogue) could gain at most in the K loop.
usage: python tools/probe_gemm_mfma32.py ; MV_PROBE_LIB=tools/probe/libconv_mfma32.so python tools/bench_conv.py  (beside the product run)"""
import glob, os, re, shutil, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
d = '/tmp/probe_mfma32'
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/arch')
for f in glob.glob(PKG + '/csrc/*.h') + [PKG + '/csrc/conv1d.hip']:
    shutil.copy(f, d)
p = d + '/common.h'
t = open(p).read().replace('"../../include/mvector_hip.h"', '"%s/include/mvector_hip.h"' % REPO)
open(p, 'w').write(t)

# ---- arch header: a second step function on 16-register accumulators
s = open(PKG + '/csrc/arch/gfx950.h').read()
STEP32 = r'''
// PROBE: four 32x32x16 MFMAs on the operands of eight 16x16x32 ones (two 16-register accumulators)
template <int WAIT>
__device__ __forceinline__ void mfma8_step32(float16v& q0, float16v& q1, const half8v& a0, const half8v& a1, const half8v (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %5, %1\n\t"
                 "v_mfma_f32_32x32x16_f16 %0, %2, %6, %0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %7, %1"
                 : "+v"(q0), "+v"(q1)
                 : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(WAIT)
                 : "memory");
}
'''
marker = '#undef MV_MFMA8\n'
assert s.count(marker) == 1
s = s.replace(marker, marker + STEP32)
open(d + '/arch/gfx950.h', 'w').write(s)

# ---- conv1d.hip: a copy of the stage on float16v accumulators, used by the ring kernel only
c = open(d + '/conv1d.hip').read()
i0 = c.index('template <class F>\n__device__ __forceinline__ void mma_stage_8x4(')
i1 = c.index('\n}\n', i0) + 3
stage = c[i0:i1]
stage_q = stage.replace('mma_stage_8x4(', 'mma_stage_8x4_q(').replace('float4v (&acc)[8][4]', 'float16v (&acc)[8]').replace('mfma8_step<', 'mfma8_step32<')
assert stage_q.count('mfma8_step32<') == 8
c = c[:i1] + '\n' + stage_q + c[i1:]
k0 = c.index('__global__ __launch_bounds__(512) void conv1d_ring_persistent_kernel(ConvArgs a) {')
k1 = c.index('\n}\n', k0) + 3
ring = c[k0:k1]


def rrep(a, b, n=1):
    global ring
    assert ring.count(a) == n, (a[:70], ring.count(a))
    ring = ring.replace(a, b)


rrep('    float4v acc[MI][NI];', '    float16v acc[MI];')
rrep('#pragma unroll\n                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = b4;',
     'acc[mi] = __builtin_shufflevector(__builtin_shufflevector(b4, b4, 0, 1, 2, 3, 0, 1, 2, 3), __builtin_shufflevector(b4, b4, 0, 1, 2, 3, 0, 1, 2, 3), '
     '0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);')
rrep('mma_stage_8x4(wt, xt, wc, wn, lane, acc, ', 'mma_stage_8x4_q(wt, xt, wc, wn, lane, acc, ')
VIEW = '''{
        float4v accv[8][4];
#pragma unroll
        for (int mi_ = 0; mi_ < 8; ++mi_) {
            accv[mi_][0] = __builtin_shufflevector(acc[mi_], acc[mi_], 0, 1, 2, 3);
            accv[mi_][1] = __builtin_shufflevector(acc[mi_], acc[mi_], 4, 5, 6, 7);
            accv[mi_][2] = __builtin_shufflevector(acc[mi_], acc[mi_], 8, 9, 10, 11);
            accv[mi_][3] = __builtin_shufflevector(acc[mi_], acc[mi_], 12, 13, 14, 15);
        }
        persistent_epilogue<0>(a, hp, e_n0, e_co0, wc, wn, lane, accv);
    }'''
rrep('if (pending) persistent_epilogue<0>(a, hp, e_n0, e_co0, wc, wn, lane, acc);', 'if (pending) ' + VIEW)
rrep('    persistent_epilogue<0>(a, hp, e_n0, e_co0, wc, wn, lane, acc);', '    ' + VIEW)
c = c[:k0] + ring + c[k1:]
open(d + '/conv1d.hip', 'w').write(c)

obj = d + '/conv1d.o'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-inline-asm', '-DNDEBUG', '-save-temps=obj',
                       '-I', d, '-I', PKG + '/csrc', '-x', 'hip', '-c', d + '/conv1d.hip', '-o', obj], cwd=d)
asm = open(glob.glob(d + '/*gfx950.s')[0]).read()
k = asm[asm.index('\n_ZN2mv29conv1d_ring_persistent_kernelENS_8ConvArgsE:'):]
k = k[:k.index('s_endpgm')]
first, last = k.index('v_mfma'), k.rindex('v_mfma')
print('ring kernel: 32x32x16 MFMAs', k.count('v_mfma_f32_32x32x16_f16'), '| 16x16x32 MFMAs', k.count('v_mfma_f32_16x16x32_f16'), '| v_mov_b32 between the first and the last MFMA',
      k[first:last].count('v_mov_b32'), '| scratch ops', k.count('scratch_'))
print(re.findall(r'\.set _ZN2mv29conv1d_ring_persistent_kernelENS_8ConvArgsE\.(num_vgpr|private_seg_size), (\d+)', asm))
objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if not o.endswith('/conv1d.hip.o')]
out = os.path.join(REPO, 'tools', 'probe', 'libconv_mfma32.so')
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
print('built', out)
