// Does gfx950 honour MODE.FP16_OVFL (bit 23: an overflowed fp16 result is clamped to +-65504 instead of +-inf, true infinities kept)?
// hipcc --offload-arch=gfx950 -O2 tools/fp16_ovfl_probe.hip -o /tmp/p && /tmp/p
// If it does, the fp16 saturation the conv / Res2Net epilogues spell out (v_med3_f32 per value, v_pk_min / v_pk_max per pair) is one s_setreg per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const float* in, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n s_nop 4" ::: "memory");
    const float a = in[0], b = in[1];
    _Float16 h = (_Float16)a;                       // v_cvt_f16_f32
    half2v p = {(_Float16)a, (_Float16)b};          // v_cvt_pk / two cvts
    half2v q = {(_Float16)in[2], (_Float16)in[3]};
    half2v s = p + q;                               // v_pk_add_f16
    half2v m = q * q;                               // v_pk_mul_f16
    if (threadIdx.x == 0) {
        out[0] = (float)h; out[1] = (float)p[0]; out[2] = (float)p[1]; out[3] = (float)s[0]; out[4] = (float)s[1]; out[5] = (float)m[0]; out[6] = (float)m[1];
    }
}
int main() {
    float *d, *din; hipMalloc(&d, 64); hipMalloc(&din, 16);
    const float cases[][4] = {{1.0e6f, -7.0e4f, 60000.0f, -60000.0f}, {65520.0f, 65519.0f, 300.0f, -300.0f}, {INFINITY, -INFINITY, 1.0f, 2.0f}, {NAN, 3.0f, 65504.0f, 65504.0f}};
    for (int ovfl = 0; ovfl < 2; ++ovfl)
        for (auto& c : cases) {
            hipMemcpy(din, c, 16, hipMemcpyHostToDevice);
            k<<<1, 64>>>(d, din, ovfl);
            float h[7]; hipMemcpy(h, d, 28, hipMemcpyDeviceToHost);
            printf("FP16_OVFL=%d in=(%g %g %g %g): cvt %g | pair (%g %g) | pk_add (%g %g) | pk_mul (%g %g)\n", ovfl, c[0], c[1], c[2], c[3], h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
        }
    return 0;
}
