// CAM++ front-end (FCM, mvector/models/campplus.py:221-292): 3x3 2-D convolutions with 32 feature maps over the
// (frequency, time) plane, BatchNorm folded into the weights (BN follows every conv directly, campplus.py:250-251,285,288).
//
// Layout: feature maps are channel-last fp16 [B, F, T, 32] -- one (frequency, time) position is 64 contiguous bytes and a
// 3x3 tap is exactly one MFMA 16x16x32 K step (32 input maps).  A workgroup owns one (utterance, output frequency) row and
// walks it in 128-frame tiles: the three input rows (+1 frame halo) are staged in LDS once, the ten [32 x 32] weight
// matrices (nine taps + the optional 1x1 shortcut tap of BasicResBlock) live in registers as MFMA A fragments for the
// whole kernel, each wave produces 32 positions x 32 maps.  The residual branch is fused: either the strided 1x1
// shortcut conv (+BN) as a tenth tap on the block input, or the identity add in the epilogue, then ReLU.
#include "kernels.h"

namespace mv {

constexpr int FCM_TT = 128;  // frames per tile
constexpr int FCM_C = 32;    // feature maps

// ---- first conv: one input map (the fp32 features [B, T, F], read transposed), K = 9 -> plain VALU -----------------
__global__ __launch_bounds__(256) void fcm_conv1_kernel(const float* feats, half_t* out, const float* w, const float* bias,
                                                        int B, int T, int F) {
    __shared__ float sw[FCM_C * 9 + FCM_C];
    for (int i = threadIdx.x; i < FCM_C * 9; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < FCM_C; i += 256) sw[FCM_C * 9 + i] = bias[i];
    __syncthreads();
    const int64_t total = (int64_t)B * F * T * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i & 3);
        const int64_t p = i >> 2;  // (b*F + f)*T + t
        const int t = (int)(p % T);
        const int f = (int)((p / T) % F);
        const int b = (int)(p / ((int64_t)T * F));
        float x[9];
#pragma unroll
        for (int df = 0; df < 3; ++df)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int ff = f + df - 1, tt = t + dt - 1;
                x[df * 3 + dt] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? feats[((int64_t)b * T + tt) * F + ff] : 0.0f;
            }
        half8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = cc * 8 + e;
            float acc = sw[FCM_C * 9 + co];
#pragma unroll
            for (int j = 0; j < 9; ++j) acc += sw[co * 9 + j] * x[j];
            acc = fminf(fmaxf(acc, 0.0f), 65504.0f);
            o[e] = (half_t)acc;
        }
        *reinterpret_cast<half8v*>(out + p * FCM_C + cc * 8) = o;
    }
}

struct FcmConvArgs {
    const half_t* x;   // [B, Fin, T, 32]
    const half_t* x2;  // optional [B, F2, T, 32]
    const half_t* w;   // [ntaps][32 co][32 ci] fp16, BN folded
    const float* bias; // [32]
    half_t* y;
    int64_t y_sB, y_sF, y_sT;
    int B, T, Fin, Fout, sf, F2, sf2, mode2;  // mode2: 0 none, 1 = 1x1 shortcut tap on x2, 2 = identity add of x2
};

__device__ __forceinline__ int fcm_lds_off(int row_pos, int chunk) { return row_pos * 64 + ((chunk ^ ((row_pos >> 1) & 3)) << 4); }

__global__ __launch_bounds__(256) void fcm_conv3x3_kernel(FcmConvArgs a) {
    __shared__ __attribute__((aligned(16))) char sx[3 * (FCM_TT + 2) * 64];
    __shared__ __attribute__((aligned(16))) char sx2[FCM_TT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: workgroup i runs on XCD i % 8 and every XCD has its own L2.  Neighbouring output rows share two of their
    // three input rows, so each XCD gets one contiguous run of the (utterance, output row) space: the shared rows are then L2
    // hits on the XCD that fetched them instead of being fetched by three L2s (measured before: ~3 x the algorithmic reads,
    // the whole FCM stack at the HBM / Infinity-Cache bandwidth limit).
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, per = nwg >> 3, rem = nwg & 7;
    const int logical = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const int b = logical / a.Fout;
    const int fo = logical - b * a.Fout;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntaps = a.mode2 == 1 ? 10 : 9;

    half8v wf[10][2];
#pragma unroll
    for (int tap = 0; tap < 10; ++tap)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (tap < ntaps) {
                wf[tap][mi] = *reinterpret_cast<const half8v*>(a.w + ((tap * FCM_C + mi * 16 + fr) * FCM_C + 8 * fg));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[tap][mi][e] = (half_t)0.0f;
            }
        }
    float bias[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mi][r] = a.bias[mi * 16 + 4 * fg + r];

    const half8v zero8 = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f,
                                (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
    for (int t0 = 0; t0 < a.T; t0 += FCM_TT) {
        // ---- stage the three input rows (one frame of halo on each side) and the residual row ----
        for (int i = tid; i < 3 * (FCM_TT + 2) * 4; i += 256) {
            const int c4 = i & 3;
            const int rp = i >> 2;  // r*(TT+2) + pi
            const int r = rp / (FCM_TT + 2);
            const int pi = rp - r * (FCM_TT + 2);
            const int fin = fo * a.sf + r - 1;
            const int t = t0 + pi - 1;
            half8v v = zero8;
            if (fin >= 0 && fin < a.Fin && t >= 0 && t < a.T)
                v = *reinterpret_cast<const half8v*>(a.x + (((int64_t)b * a.Fin + fin) * a.T + t) * FCM_C + c4 * 8);
            *reinterpret_cast<half8v*>(sx + fcm_lds_off(rp, c4)) = v;
        }
        if (a.mode2 != 0) {
            const int f2 = fo * a.sf2;
            for (int i = tid; i < FCM_TT * 4; i += 256) {
                const int c4 = i & 3;
                const int pi = i >> 2;
                const int t = t0 + pi;
                half8v v = zero8;
                if (t < a.T) v = *reinterpret_cast<const half8v*>(a.x2 + (((int64_t)b * a.F2 + f2) * a.T + t) * FCM_C + c4 * 8);
                *reinterpret_cast<half8v*>(sx2 + fcm_lds_off(pi, c4)) = v;
            }
        }
        __syncthreads();
        float4v acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = float4v{bias[mi][0], bias[mi][1], bias[mi][2], bias[mi][3]};
#pragma unroll
        for (int df = 0; df < 3; ++df)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int rp = df * (FCM_TT + 2) + wave * 32 + ni * 16 + fr + dt;
                    const half8v bf = *reinterpret_cast<const half8v*>(sx + fcm_lds_off(rp, fg));
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[df * 3 + dt][mi], bf, acc[mi][ni], 0, 0, 0);
                }
            }
        if (a.mode2 == 1) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const half8v bf = *reinterpret_cast<const half8v*>(sx2 + fcm_lds_off(wave * 32 + ni * 16 + fr, fg));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[9][mi], bf, acc[mi][ni], 0, 0, 0);
            }
        }
        // ---- epilogue: lane holds maps co..co+3 of one position ----
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int pos = wave * 32 + ni * 16 + fr;
            const int t = t0 + pos;
            if (t < a.T) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int co = mi * 16 + 4 * fg;
                    float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                    if (a.mode2 == 2) {
                        // identity residual: element co of the row is in chunk co/8 at offset co%8
                        const half4v rv = *reinterpret_cast<const half4v*>(sx2 + fcm_lds_off(pos, co >> 3) + (co & 7) * 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    }
                    half4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)fminf(fmaxf(v[r], 0.0f), 65504.0f);
                    *reinterpret_cast<half4v*>(a.y + (int64_t)b * a.y_sB + (int64_t)fo * a.y_sF + (int64_t)t * a.y_sT + co) = o;
                }
            }
        }
        __syncthreads();
    }
}

int fcm_conv1_launch(const float* feats, half_t* out, const float* w, const float* bias, int B, int T, int F,
                     hipStream_t stream) {
    const int64_t total = (int64_t)B * F * T * 4;
    const int grid = (int)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    MV_LAUNCH(fcm_conv1_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, feats, out, w, bias, B, T, F);
    return check_launch("fcm_conv1_kernel");
}

int fcm_conv3x3_launch(const half_t* x, int Fin, int sf, const half_t* x2, int F2, int sf2, int mode2, const half_t* w,
                       const float* bias, half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int B, int T, int Fout,
                       hipStream_t stream) {
    MV_REQUIRE(x != nullptr && w != nullptr && bias != nullptr && y != nullptr, "fcm_conv3x3: null tensor");
    MV_REQUIRE(mode2 == 0 || x2 != nullptr, "fcm_conv3x3: residual input missing");
    MV_REQUIRE((int64_t)B * Fout < ((int64_t)1 << 31), "fcm_conv3x3: grid too large");
    FcmConvArgs a;
    a.x = x;
    a.x2 = x2;
    a.w = w;
    a.bias = bias;
    a.y = y;
    a.y_sB = y_sB;
    a.y_sF = y_sF;
    a.y_sT = y_sT;
    a.B = B;
    a.T = T;
    a.Fin = Fin;
    a.Fout = Fout;
    a.sf = sf;
    a.F2 = F2;
    a.sf2 = sf2;
    a.mode2 = mode2;
    MV_LAUNCH(fcm_conv3x3_kernel, ((unsigned)(B * Fout), 1, 1), (256, 1, 1), 0, stream, a);
    return check_launch("fcm_conv3x3_kernel");
}

}  // namespace mv
