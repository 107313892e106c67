// Stem conv and temporal statistics pooling of the ERes2Net family and of the CAM++ exact head (mvector/models/eres2net.py:196-201,250;
// mvector/models/pooling.py:130-148), on fp32 maps and on the S16 maps of conv2ds.hip (s16map.h).  The layers between them run conv2ds.hip.
// (Rounds 2-3 evaluated those layers here with fp32 maps and weights on v_mfma_f32_16x16x4_f32; since round 4 no model handle runs that form, and
// since round 5 it lives under tools/yardstick/ -- the exact-fp32 yard-stick of tools/bench_conv2d.py -- not in the product library.)
#include "kernels.h"
#include "s16map.h"

namespace mv {

// ---- first conv of ERes2Net (eres2net.py:196-201, 250): one input map = the fp32 features [B, T, F] read transposed,
// 3x3, zero padding, BatchNorm folded, plain ReLU.  K = 9 -> VALU; a lane produces 8 output maps of one position.
// S16: the output map in the split-fp16 form of conv2ds.hip (s16map.h) instead of fp32
template <bool S16>
__global__ __launch_bounds__(256) void conv2d_first_kernel(const float* feats, float* out, const float* w, const float* bias,
                                                           int B, int T, int F, int C, unsigned* peak) {
    MV_DYN_SMEM(smem);
    float* sw = reinterpret_cast<float*>(smem);  // [C][9] weights, then [C] bias
    for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < C; i += 256) sw[C * 9 + i] = bias[i];
    __syncthreads();
    const int groups = C >> 3;
    const int64_t total = (int64_t)B * F * T * groups;
    float pk = 0.0f;   // S16: largest scaled value this lane stores (s16map.h)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % groups);
        const int64_t pix = i / groups;  // (b*F + f)*T + t
        const int t = (int)(pix % T);
        const int f = (int)((pix / T) % F);
        const int b = (int)(pix / ((int64_t)T * F));
        float x[9];
#pragma unroll
        for (int df = 0; df < 3; ++df)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int ff = f + df - 1, tt = t + dt - 1;
                x[df * 3 + dt] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? feats[((int64_t)b * T + tt) * F + ff] : 0.0f;
            }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = cg * 8 + e;
            float acc = sw[C * 9 + co];
#pragma unroll
            for (int j = 0; j < 9; ++j) acc += sw[co * 9 + j] * x[j];
            o[e] = fmaxf(acc, 0.0f);
        }
        if (S16) {
            half_t* unit = reinterpret_cast<half_t*>(out) + (pix * C + (cg >> 1) * 16) * 2 + (cg & 1) * 8;
            s16_store4(unit, float4v{o[0], o[1], o[2], o[3]});
            s16_store4(unit + 4, float4v{o[4], o[5], o[6], o[7]});
            pk = s16_peak_of(s16_peak_of(pk, float4v{o[0], o[1], o[2], o[3]}), float4v{o[4], o[5], o[6], o[7]});
        } else {
            *reinterpret_cast<float4v*>(out + pix * C + cg * 8) = float4v{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<float4v*>(out + pix * C + cg * 8 + 4) = float4v{o[4], o[5], o[6], o[7]};
        }
    }
    if (S16 && peak != nullptr) s16_peak_commit(peak, pk * CS_XSCALE);   // (uniform condition: every lane arrives)
}

int conv2d_first_launch(const float* feats, float* out, const float* w, const float* bias, int B, int T, int F, int C,
                        hipStream_t stream) {
    MV_REQUIRE(feats != nullptr && out != nullptr && w != nullptr && bias != nullptr, "conv2d_first: null pointer");
    MV_REQUIRE(C > 0 && C % 8 == 0 && C <= 1024, "conv2d_first: output maps must be a multiple of 8");
    const int64_t total = (int64_t)B * F * T * (C / 8);
    const int grid = (int)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384);
    MV_LAUNCH(conv2d_first_kernel<false>, (grid, 1, 1), (256, 1, 1), (size_t)C * 10 * sizeof(float), stream, feats, out, w, bias, B, T, F, C, static_cast<unsigned*>(nullptr));
    return check_launch("conv2d_first_kernel");
}

int conv2d_first_s16_launch(const float* feats, half_t* out, const float* w, const float* bias, int B, int T, int F, int C, hipStream_t stream, unsigned* peak) {
    MV_REQUIRE(feats != nullptr && out != nullptr && w != nullptr && bias != nullptr, "conv2d_first_s16: null pointer");
    MV_REQUIRE(C > 0 && C % 16 == 0 && C <= 1024, "conv2d_first_s16: output maps must be a multiple of 16");
    const int64_t total = (int64_t)B * F * T * (C / 8);
    const int grid = (int)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384);
    MV_LAUNCH(conv2d_first_kernel<true>, (grid, 1, 1), (256, 1, 1), (size_t)C * 10 * sizeof(float), stream, feats, reinterpret_cast<float*>(out), w, bias, B,
              T, F, C, peak);
    return check_launch("conv2d_first_kernel");
}

// ---- temporal statistics pooling (mvector/models/pooling.py:130-148) over channel-last maps [B, H, W, C]:
// stats[b, c*H + h] = mean over W, stats[b, C*H + c*H + h] = sqrt(unbiased var + 1e-8) -- the reference flattens [B, C, H].
// One workgroup per (utterance, frequency row); moments about the first time step (see time_stats_kernel).
template <bool S16>
__global__ __launch_bounds__(256) void tstp_kernel(const float* x, int64_t ld, int H, int W, int C, int Creal, float* stats) {
    __shared__ float red[2][256];
    const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + ((int64_t)b * H + h) * W * ld;
    for (int c0 = 0; c0 < Creal; c0 += 64) {  // 64 channels x 4 time phases per pass
        const int c = c0 + (tid & 63), ph = tid >> 6;
        float s1 = 0.0f, s2 = 0.0f, k = 0.0f;
        if (c < Creal) {
            const half_t* xh = reinterpret_cast<const half_t*>(xr);
            k = S16 ? s16_load1(xh, 0, c) : xr[c];
            for (int w = ph; w < W; w += 4) {
                const float d = (S16 ? s16_load1(xh, (int64_t)w * ld, c) : xr[(int64_t)w * ld + c]) - k;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
        }
        red[0][tid] = s1;
        red[1][tid] = s2;
        __syncthreads();
        if (tid < 64 && c < Creal) {
            const float z1 = red[0][tid] + red[0][tid + 64] + red[0][tid + 128] + red[0][tid + 192];
            const float z2 = red[1][tid] + red[1][tid + 64] + red[1][tid + 128] + red[1][tid + 192];
            const float var = fmaxf(z2 - z1 * z1 / (float)W, 0.0f) / (float)(W - 1);
            float* sb = stats + (int64_t)b * 2 * Creal * H;
            sb[c * H + h] = k + z1 / (float)W;
            sb[Creal * H + c * H + h] = sqrtf(var + 1e-8f);
        }
        __syncthreads();
    }
    (void)C;
}

int tstp_launch(const float* x, int64_t ld, int B, int H, int W, int C, float* stats, hipStream_t stream) {
    MV_REQUIRE(x != nullptr && stats != nullptr && B > 0 && H > 0 && W > 1 && C > 0 && ld >= C, "tstp: bad argument");
    MV_REQUIRE(H <= 65535 && B <= 65535, "tstp: grid too large");
    MV_LAUNCH(tstp_kernel<false>, ((unsigned)H, (unsigned)B, 1), (256, 1, 1), 0, stream, x, ld, H, W, (int)ld, C, stats);
    return check_launch("tstp_kernel");
}

int tstp_s16_launch(const half_t* x, int64_t ld, int B, int H, int W, int C, float* stats, hipStream_t stream) {
    MV_REQUIRE(x != nullptr && stats != nullptr && B > 0 && H > 0 && W > 1 && C > 0 && ld >= C && ld % 16 == 0, "tstp_s16: bad argument");
    MV_REQUIRE(H <= 65535 && B <= 65535, "tstp_s16: grid too large");
    MV_LAUNCH(tstp_kernel<true>, ((unsigned)H, (unsigned)B, 1), (256, 1, 1), 0, stream, reinterpret_cast<const float*>(x), ld, H, W, (int)ld, C, stats);
    return check_launch("tstp_kernel");
}

}  // namespace mv

extern "C" {

int mv_conv2d_first(const float* feats, float* out, const float* w, const float* bias, int32_t B, int32_t T, int32_t F, int32_t C,
                    mv_stream_t stream) {
    return mv::conv2d_first_launch(feats, out, w, bias, B, T, F, C, static_cast<hipStream_t>(stream));
}

int mv_tstp_f32(const float* x, int64_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* stats, mv_stream_t stream) {
    return mv::tstp_launch(x, ld, B, H, W, C, stats, static_cast<hipStream_t>(stream));
}

int mv_conv2d_first_s16(const float* feats, void* out, const float* w, const float* bias, int32_t B, int32_t T, int32_t F, int32_t C,
                        mv_stream_t stream) {
    return mv::conv2d_first_s16_launch(feats, static_cast<half_t*>(out), w, bias, B, T, F, C, static_cast<hipStream_t>(stream));
}

int mv_tstp_s16(const void* x, int64_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* stats, mv_stream_t stream) {
    return mv::tstp_s16_launch(static_cast<const half_t*>(x), ld, B, H, W, C, stats, static_cast<hipStream_t>(stream));
}

}  // extern "C"
