// Layout probe for v_mfma_f32_4x4x1_16B_f32 (prints which lane supplies each D element).  hipcc --offload-arch=gfx950 -o mfma4x4 mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    float4v z = {0, 0, 0, 0};
    float4v d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.0f, z, 0, 0, 0);   // D = A-lane id of the row supplier
    float4v d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(lane + 1), z, 0, 0, 0);   // D = B-lane id of the column supplier
    for (int r = 0; r < 4; ++r) {
        out[lane * 8 + r] = d1[r];
        out[lane * 8 + 4 + r] = d2[r];
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 8 * 4);
    probe<<<1, 64>>>(d);
    float h[512];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: A-supplier of D[r]:", l);
        for (int r = 0; r < 4; ++r) printf(" %2d", (int)h[l * 8 + r] - 1);
        printf("   B-supplier:");
        for (int r = 0; r < 4; ++r) printf(" %2d", (int)h[l * 8 + 4 + r] - 1);
        printf("\n");
    }
    return 0;
}
