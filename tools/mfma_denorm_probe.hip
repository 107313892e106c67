// Does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs?  hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float a_val, float b_val) {
    half8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }   // k = 0 only
    float4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[][2] = {{1.0f, 3.0e-6f}, {3.0e-6f, 1024.0f}, {1.0f, 6.0e-8f}, {2.0e-5f, 2.0e-5f}};
    for (auto& v : vals) {
        k<<<1, 64>>>(d, v[0], v[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma=%.9g  expected=%.9g\n", v[0], v[1], h, (float)(_Float16)v[0] * (float)(_Float16)v[1]);
    }
    return 0;
}
