// Semantics probe for v_permlane16_swap_b32 (gfx950): prints, per lane, which (operand, lane) each result came from.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    unsigned h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("row %d: result0[lane %d] = %u   result1[lane %d] = %u\n", row, row * 16, h[row * 16], row * 16, h[64 + row * 16]);
    return 0;
}
