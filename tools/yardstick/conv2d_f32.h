/* NOT PART OF THE PRODUCT ABI.  fp32-operand form of the ERes2Net conv layers (rounds 2-3): the exact-fp32 yard-stick of tools/bench_conv2d.py and
 * tools/yardstick/check_conv2d.py, built by tools/yardstick/build.py.  Semantics: include/mvector_hip.h, "2-D convolution layers". */
#pragma once
#include "../../include/mvector_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pack [Cout][Cin][k][k] fp32 (nn.Conv2d layout) times an optional per-output-channel scale -> fp32 [cout16][k*k][cin16] */
int64_t mv_conv2d_packed_elems(int32_t cout, int32_t cin, int32_t ks);
int mv_conv2d_pack_weight(const float* w, const float* out_scale, int32_t cout, int32_t cin, int32_t ks, float* packed,
                          mv_stream_t stream);
typedef struct MvConv2dDesc {
    const float* x;      /* [B, H, W, ldx] */
    const float* x2;     /* optional second input */
    int32_t x2_mode;     /* 0 none, 1 added, 2 concatenated behind the first cin1 channels */
    int32_t cin1;
    int64_t ldx, ldx2;
    const float* w;      /* packed [cout16][ks*ks][cin16] */
    const float* bias;   /* [cout16] */
    const float* res;    /* epi 0: optional residual; epi 2: first AFF operand; [B, Ho, Wo, ldres] */
    const float* res2;   /* epi 2: second AFF operand */
    int64_t ldres, ldres2;
    float* y;            /* [B, Ho, Wo, ldy], Ho = (H + 2*(ks/2) - ks)/stride + 1, Wo likewise with stride_w */
    int64_t ldy;
    int32_t B, H, W, cin16, cout16, ks, stride, epi;
    float lo, hi;
    int32_t cin_alg, cout_alg; /* channel counts of the layer before padding (0: same as cin16 / cout16): profile accounting only */
    int32_t stride_w;          /* stride along W when it differs from `stride` (then the stride along H); 0 = same.  The CAM++ head
                                * (campplus.py:221-292) strides the frequency axis only: stride 2, stride_w 1 */
} MvConv2dDesc;
int mv_conv2d_forward(const MvConv2dDesc* d, mv_stream_t stream);

#ifdef __cplusplus
}
#endif
