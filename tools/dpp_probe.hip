// Lane-layout probe for the DPP controls fbank_tile_kernel relies on (row_mirror, row_shr:1 with `old`, row_shl:4/8):
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
// prints, for lanes 0..15 of a row, which lane each control reads (value = source lane id; -1 = kept `old`).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false); }

__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    out[0 * 64 + lane] = dpp<0x140>(-1, lane);   // row_mirror
    out[1 * 64 + lane] = dpp<0x111>(-1, lane);   // row_shr:1
    out[2 * 64 + lane] = dpp<0x104>(-1, lane);   // row_shl:4
    out[3 * 64 + lane] = dpp<0x108>(-1, lane);   // row_shl:8
    out[4 * 64 + lane] = dpp<0x121>(-1, lane);   // row_ror:1
    out[5 * 64 + lane] = __builtin_amdgcn_mov_dpp(lane, 0x140, 0xf, 0xf, false);
}

int main() {
    int* d;
    hipMalloc(&d, 6 * 64 * sizeof(int));
    probe<<<1, 64>>>(d);
    int h[6 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"row_mirror", "row_shr:1 ", "row_shl:4 ", "row_shl:8 ", "row_ror:1 ", "mov_dpp mir"};
    for (int r = 0; r < 6; ++r) {
        printf("%s:", names[r]);
        for (int l = 16; l < 32; ++l) printf(" %3d", h[r * 64 + l] < 0 ? -1 : h[r * 64 + l] - 16);
        printf("\n");
    }
    return 0;
}
