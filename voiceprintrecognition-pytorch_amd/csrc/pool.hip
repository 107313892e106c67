// Time-axis reductions and the SE / attentive-statistics pooling kernels (channel-last fp16 activations).
//
//   time_stats_kernel    mean (and std) over T per (utterance, channel): SE squeeze (ecapa_tdnn.py:79),
//                        ASP global context (pooling.py:104-109), CAM++ StatsPool (campplus.py:27-33)
//   seg_mean_kernel      CAM context = mean over T + mean over 100-frame segments (campplus.py:94-111)
//   se_gate_residual     out = gate[b, c] * y + residual (ecapa_tdnn.py:84,143), written into a channel slice
//                        of the aggregation buffer so torch.cat (ecapa_tdnn.py:273) never materialises
//   asp_pool_kernel      attention logits (128 -> C projection on MFMA), softmax over time and the weighted
//                        mean / std (pooling.py:117-125) in one kernel: the [B, C, T] logits never exist
#include "common.h"

namespace mv {

// Grid of the element-wise passes (one 16-byte group per thread): ONE group per thread, no grid-stride walk over a capped grid.  Round 6 (r15n / o):
// se_gate_residual over [256 x 298, 1024] with 2048 / 4096 / 8192 / 16384 / 38144 (= all) workgroups 82.7 / 82.1 / 78.9 / 76.6 / 73.1 us -- the kernels keep their
// loops (a cap of 2^20 workgroups still bounds the grid), the hardware's workgroup dispatcher spreads the memory traffic better than a fixed stride does.
constexpr int64_t EW_MAX_GRID = (int64_t)1 << 20;
static inline int ew_grid(int64_t groups) {
    const int64_t g = (groups + 255) / 256;
    return (int)(g < 1 ? 1 : (g < EW_MAX_GRID ? g : EW_MAX_GRID));
}


// ------------------------------------------------------------------------------------------------
// mean / std over time.  Workgroup = (utterance, 128-channel group): lane (c16 = lane & 15) owns 8 channels, the four
// 16-lane groups of a wave and the four waves take interleaved time steps (16 rows in flight per workgroup), so a
// [B=256, C=1024] reduction runs 2048 workgroups.  ONE pass over x: the moments are taken about the channel's first
// time step k = x[b, 0, c] (s1 = sum (x - k), s2 = sum (x - k)^2), which conditions the variance like the reference's
// (x - mean)^2 form -- |mean - k| is of the order of the spread -- and makes a constant channel's variance exactly zero:
//   mean = k + s1 / T,   var = (s2 - s1^2 / T) / (T or T - 1)
// Optional pre-activation: v = relu(v * in_scale[c] + in_shift[c]) (CAM++ out_nonlinear, campplus.py:344-345).
// PIPE: four rows in flight per lane (small grids; costs registers, i.e. waves per SIMD, which a full batch needs more than the overlap)
// LEAN: the mean alone of a plain tensor (no pre-activation, no std: the SE squeeze) -- conversion, subtraction, addition per value.  The general form spends
// seven vector operations per value (a run-time select around the pre-activation, the second moment) and is bound by them, not by HBM: 11.4 M vector
// instructions per launch at the headline shape = 21 us of issue in a 29 us kernel (PMC r15ay).  Same additions in the same order: the same mean.
template <bool PIPE, bool LEAN = false>
__global__ __launch_bounds__(256) void time_stats_kernel(const half_t* x, int64_t ld, int T, int C, float* mean,
                                                         float* stdv, int64_t ld_out, int unbiased, float clamp_eps,
                                                         const float* in_scale, const float* in_shift) {
    __shared__ float red[2][4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, rp = lane >> 4;
    const int b = blockIdx.y;
    const int cg0 = blockIdx.x * 128;
    const int c0 = cg0 + c16 * 8;
    const bool active = c0 < C;
    const int nvalid = active ? (C - c0 < 8 ? C - c0 : 8) : 0;
    const half_t* xb = x + (int64_t)b * T * ld + c0;
    const int t_first = wave * 4 + rp;  // this lane's rows: t_first, t_first + 16, ...
    float s1[8], s2[8], k8[8], isc[8], ish[8];
    const bool pre = !LEAN && in_scale != nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = s2[e] = k8[e] = 0.0f;
        isc[e] = (pre && e < nvalid) ? in_scale[c0 + e] : 1.0f;
        ish[e] = (pre && e < nvalid) ? in_shift[c0 + e] : 0.0f;
    }
    auto val = [&](half_t h, int e) {
        const float v = (float)h;
        return pre ? fmaxf(v * isc[e] + ish[e], 0.0f) : v;
    };
    if (active) {
        if (nvalid == 8) {
            const half8v v0 = *reinterpret_cast<const half8v*>(xb);
#pragma unroll
            for (int e = 0; e < 8; ++e) k8[e] = val(v0[e], e);
        } else {
            for (int e = 0; e < nvalid; ++e) k8[e] = val(xb[e], e);
        }
        int t = t_first;
        if (PIPE && nvalid == 8) {
            // small grids only: four rows in flight per lane, same order of the additions -- with one utterance on the chip the pass is
            // a chain of dependent load trips, 19 of them for 3 s: 13 -> 10 us for 610 KB.  A full batch (2048 workgroups) streams at the HBM rate
            // with one load per lane in flight and LOSES with four (29 -> 38 us at 256 x 3 s, r12l: fewer waves per SIMD)
            for (; t + 48 < T; t += 64) {
                half8v v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const half8v*>(xb + (int64_t)(t + 16 * u) * ld);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = val(v4[u][e], e) - k8[e];
                        s1[e] += d;
                        if (!LEAN) s2[e] = fmaf(d, d, s2[e]);
                    }
                }
            }
        }
        for (; t < T; t += 16) {
            const half_t* p = xb + (int64_t)t * ld;
            if (nvalid == 8) {
                const half8v v = *reinterpret_cast<const half8v*>(p);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = val(v[e], e) - k8[e];
                    s1[e] += d;
                    if (!LEAN) s2[e] = fmaf(d, d, s2[e]);
                }
            } else {
                for (int e = 0; e < nvalid; ++e) {
                    const float d = val(p[e], e) - k8[e];
                    s1[e] += d;
                    if (!LEAN) s2[e] = fmaf(d, d, s2[e]);
                }
            }
        }
    }
    // sum over the 4 row groups of the wave, then over waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a1 = s1[e], a2 = s2[e];
        a1 += __shfl_xor(a1, 16);
        a1 += __shfl_xor(a1, 32);
        if (!LEAN) {
            a2 += __shfl_xor(a2, 16);
            a2 += __shfl_xor(a2, 32);
        }
        if (rp == 0) {
            red[0][wave][c16 * 8 + e] = a1;
            if (!LEAN) red[1][wave][c16 * 8 + e] = a2;
        }
    }
    __syncthreads();
    if (tid < 128 && cg0 + tid < C) {
        // k of channel tid: recomputed by its owner thread (one 2-byte load)
        float kc = (float)x[(int64_t)b * T * ld + cg0 + tid];
        if (pre) kc = fmaxf(kc * in_scale[cg0 + tid] + in_shift[cg0 + tid], 0.0f);
        const float z1 = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
        const float z2 = LEAN ? 0.0f : red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
        mean[(int64_t)b * ld_out + cg0 + tid] = kc + z1 / (float)T;
        if (!LEAN && stdv != nullptr) {
            const float denom = unbiased ? (float)(T - 1) : (float)T;
            float var = fmaxf(z2 - z1 * z1 / (float)T, 0.0f) / denom;
            if (clamp_eps > 0.0f) var = fmaxf(var, clamp_eps);
            stdv[(int64_t)b * ld_out + cg0 + tid] = sqrtf(var);
        }
    }
}

int time_stats_launch(const half_t* x, int64_t ld, int B, int T, int C, float* mean, float* stdv, int64_t ld_out,
                      int unbiased, float clamp_eps, hipStream_t stream, const float* in_scale, const float* in_shift) {
    MV_REQUIRE(x != nullptr && mean != nullptr && B > 0 && T > 0 && C > 0, "time_stats: bad argument");
    MV_REQUIRE(ld > 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "time_stats: rows must be 16-byte aligned");
    MV_REQUIRE(ld_out >= C, "time_stats: output leading dimension");
    if (unbiased) MV_REQUIRE(T > 1, "time_stats: unbiased std needs T > 1");
    const bool lean = stdv == nullptr && in_scale == nullptr;
    if (ceil_div(C, 128) * (int64_t)B <= 1024) {
        MV_LAUNCH(time_stats_kernel<true>, ((unsigned)ceil_div(C, 128), (unsigned)B, 1), (256, 1, 1), 0, stream, x, ld, T, C, mean, stdv,
                  ld_out, unbiased, clamp_eps, in_scale, in_shift);
    } else if (lean) {
        MV_LAUNCH((time_stats_kernel<false, true>), ((unsigned)ceil_div(C, 128), (unsigned)B, 1), (256, 1, 1), 0, stream, x, ld, T, C, mean, stdv,
                  ld_out, unbiased, clamp_eps, in_scale, in_shift);
    } else {
        MV_LAUNCH(time_stats_kernel<false>, ((unsigned)ceil_div(C, 128), (unsigned)B, 1), (256, 1, 1), 0, stream, x, ld, T, C, mean, stdv,
                  ld_out, unbiased, clamp_eps, in_scale, in_shift);
    }
    return check_launch("time_stats_kernel");
}

// ------------------------------------------------------------------------------------------------
// CAM context: ctx[b, s, c] = mean_t x[b, t, c] + mean_{t in segment s} x[b, t, c]  (last segment divides by its
// true length, campplus.py:103 avg_pool1d(ceil_mode=True)).  One workgroup per (utterance, 128-channel group): lane
// (c16 = tid & 15) owns 8 channels, the 16 row phases (tid >> 4) walk each segment's frames 16 at a time.
__global__ __launch_bounds__(256) void seg_mean_kernel(const half_t* x, int64_t ld, int T, int C, int seg_len, int nseg,
                                                       float* ctx) {
    __shared__ float red[16][128];
    __shared__ float tot[128];
    const int tid = threadIdx.x;
    const int c16 = tid & 15, ph = tid >> 4;
    const int b = blockIdx.y;
    const int cg0 = blockIdx.x * 128;
    const int c0 = cg0 + c16 * 8;
    const bool active = c0 + 8 <= C;  // C is a multiple of 8 (checked by the launcher)
    const half_t* xb = x + (int64_t)b * T * ld + c0;
    if (tid < 128) tot[tid] = 0.0f;
    for (int s = 0; s < nseg; ++s) {
        const int t0 = s * seg_len;
        const int t1 = t0 + seg_len < T ? t0 + seg_len : T;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
        if (active) {
            for (int t = t0 + ph; t < t1; t += 16) {
                const half8v v = *reinterpret_cast<const half8v*>(xb + (int64_t)t * ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        }
        __syncthreads();  // previous segment's readers of `red` are done
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ph][c16 * 8 + e] = acc[e];
        __syncthreads();
        if (tid < 128) {
            float v = 0.0f;
#pragma unroll
            for (int p = 0; p < 16; ++p) v += red[p][tid];
            tot[tid] += v;
            if (cg0 + tid < C) ctx[((int64_t)b * nseg + s) * C + cg0 + tid] = v / (float)(t1 - t0);
        }
    }
    if (tid < 128 && cg0 + tid < C) {
        const float gm = tot[tid] / (float)T;
        for (int s = 0; s < nseg; ++s) ctx[((int64_t)b * nseg + s) * C + cg0 + tid] += gm;
    }
}

int seg_mean_launch(const half_t* x, int64_t ld, int B, int T, int C, int seg_len, float* ctx, hipStream_t stream) {
    MV_REQUIRE(x != nullptr && ctx != nullptr && B > 0 && T > 0 && C > 0 && seg_len > 0, "seg_mean: bad argument");
    MV_REQUIRE(C % 8 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "seg_mean: rows must be 16-byte aligned");
    const int nseg = (int)ceil_div(T, seg_len);
    MV_LAUNCH(seg_mean_kernel, ((unsigned)ceil_div(C, 128), (unsigned)B, 1), (256, 1, 1), 0, stream, x, ld, T, C, seg_len, nseg,
              ctx);
    return check_launch("seg_mean_kernel");
}

// ------------------------------------------------------------------------------------------------
// out[n, c] = gate[b(n), c] * y[n, c] + res[n, c]   (8 channels per thread, grid-stride)
__global__ __launch_bounds__(256) void se_gate_residual_kernel(const half_t* y, int64_t ldy, const float* gate,
                                                               const half_t* res, int64_t ldr, half_t* out, int64_t ldo,
                                                               int T, int C, int64_t n_rows) {
    const int cgroups = C / 8;
    const int64_t total = n_rows * cgroups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / cgroups;
        const int c = (int)(i - n * cgroups) * 8;
        const int b = (int)(n / T);
        const half8v yv = *reinterpret_cast<const half8v*>(y + n * ldy + c);
        const half8v rv = *reinterpret_cast<const half8v*>(res + n * ldr + c);
        const float4v g0 = *reinterpret_cast<const float4v*>(gate + (int64_t)b * C + c);
        const float4v g1 = *reinterpret_cast<const float4v*>(gate + (int64_t)b * C + c + 4);
        half8v ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = e < 4 ? g0[e] : g1[e - 4];
            float v = g * (float)yv[e] + (float)rv[e];
            v = fminf(fmaxf(v, -65504.0f), 65504.0f);
            ov[e] = (half_t)v;
        }
        *reinterpret_cast<half8v*>(out + n * ldo + c) = ov;
    }
}

int se_gate_residual_launch(const half_t* y, int64_t ldy, const float* gate, const half_t* res, int64_t ldr, half_t* out,
                            int64_t ldo, int B, int T, int C, hipStream_t stream) {
    MV_REQUIRE(y != nullptr && gate != nullptr && res != nullptr && out != nullptr, "se_gate_residual: null tensor");
    MV_REQUIRE(C % 8 == 0 && ldy % 8 == 0 && ldr % 8 == 0 && ldo % 8 == 0, "se_gate_residual: channels must be a multiple of 8");
    const int64_t n_rows = (int64_t)B * T;
    const int64_t total = n_rows * (C / 8);
    const int grid = ew_grid(total);
    MV_LAUNCH(se_gate_residual_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, y, ldy, gate, res, ldr, out, ldo, T, C, n_rows);
    return check_launch("se_gate_residual_kernel");
}

// ------------------------------------------------------------------------------------------------
// dst[n, 0:C] = src[n, 0:C] for channel-last fp16 rows with different leading dimensions (Res2Net slice 0
// pass-through, ecapa_tdnn.py:43-44).
__global__ __launch_bounds__(256) void copy_slice_kernel(const half_t* src, int64_t lds_, half_t* dst, int64_t ldd, int C,
                                                         int64_t n_rows) {
    const int cgroups = C / 8;
    const int64_t total = n_rows * cgroups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / cgroups;
        const int c = (int)(i - n * cgroups) * 8;
        *reinterpret_cast<half8v*>(dst + n * ldd + c) = *reinterpret_cast<const half8v*>(src + n * lds_ + c);
    }
}

// dst[n, 0:C] = (fp16) src[n, 0:C]; columns C..ldd-1 of dst are zero-filled (row padding up to a multiple of 8)
__global__ __launch_bounds__(256) void cast_rows_f32_f16_kernel(const float* src, int64_t lds_, half_t* dst, int64_t ldd,
                                                                int64_t n_rows, int C) {
    const int64_t total = n_rows * ldd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / ldd;
        const int c = (int)(i - n * ldd);
        float v = c < C ? src[n * lds_ + c] : 0.0f;
        v = fminf(fmaxf(v, -65504.0f), 65504.0f);
        dst[i] = (half_t)v;
    }
}

int cast_rows_f32_f16_launch(const float* src, int64_t lds_, half_t* dst, int64_t ldd, int64_t n_rows, int C,
                             hipStream_t stream) {
    MV_REQUIRE(src != nullptr && dst != nullptr && ldd >= C, "cast_rows: bad argument");
    const int64_t total = n_rows * ldd;
    const int grid = ew_grid(total);
    MV_LAUNCH(cast_rows_f32_f16_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, src, lds_, dst, ldd, n_rows, C);
    return check_launch("cast_rows_f32_f16_kernel");
}

// dst[b, t', :] = (fp16) src[b, reflect(t' - pad), :] for t' in [0, T + 2 pad): the features with the first conv's reflect
// padding materialised (12 MB), so that the k taps of an output step are ONE contiguous run of k*C values (see
// EcapaModel::forward, block 0)
__global__ __launch_bounds__(256) void cast_reflect_pad_kernel(const float* src, half_t* dst, int B, int T, int C, int pad) {
    const int Tp = T + 2 * pad, C8 = C >> 3;                 // C % 8 == 0: one lane = 8 channels (2 x 16 B in, 16 B out)
    const int total = B * Tp * C8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c8 = i % C8, r = i / C8;
        const int b = r / Tp;
        int t = r - b * Tp - pad;
        t = t < 0 ? -t : (t >= T ? 2 * (T - 1) - t : t);
        const float4v* s4 = reinterpret_cast<const float4v*>(src + ((int64_t)b * T + t) * C) + 2 * c8;
        const float4v lo = s4[0], hi = s4[1];
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        half8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)fminf(fmaxf(v[j], -65504.0f), 65504.0f);
        *reinterpret_cast<half8v*>(dst + (int64_t)i * 8) = o;
    }
}

int cast_reflect_pad_launch(const float* src, half_t* dst, int B, int T, int C, int pad, hipStream_t stream) {
    MV_REQUIRE(src != nullptr && dst != nullptr && pad >= 0 && pad < T && C % 8 == 0, "cast_reflect_pad: bad argument");
    const int64_t total = (int64_t)B * (T + 2 * pad) * (C / 8);
    MV_REQUIRE(total < (int64_t)1 << 31, "cast_reflect_pad: batch too large");
    const int grid = ew_grid(total);
    MV_LAUNCH(cast_reflect_pad_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, src, dst, B, T, C, pad);
    return check_launch("cast_reflect_pad_kernel");
}

int copy_slice_launch(const half_t* src, int64_t lds_, half_t* dst, int64_t ldd, int C, int64_t n_rows, hipStream_t stream) {
    MV_REQUIRE(C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "copy_slice: channels must be a multiple of 8");
    const int64_t total = n_rows * (C / 8);
    const int grid = ew_grid(total);
    MV_LAUNCH(copy_slice_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, src, lds_, dst, ldd, C, n_rows);
    return check_launch("copy_slice_kernel");
}

// ------------------------------------------------------------------------------------------------
// Deferred epilogue of the ASP hidden layer (pooling.py:80-84, 110-117): the context columns of asp.tdnn multiply [mean; std] of
// the utterance -- a per-utterance bias -- so the 1x1 conv over x can run BEFORE the statistics are known (and collect them from
// its own x tiles, MvConv1dDesc.in_stat_*); this pass then adds the bias and applies ReLU -> BatchNorm -> tanh:
//   h[b, t, a] = tanh(relu(z[b, t, a] + row_bias[b, a]) * scale[a] + shift[a])      in place on fp16 [B*T, A], A % 8 == 0
// (z is stored as fp16: its rounding error, 2^-11 |z|, reaches h multiplied by tanh' = 1 - h^2 -- at most ~2e-4 around |z| ~ 0.7,
// the size of the rounding of h itself, and vanishing where |z| is large; fp32 z costs the conv its staged, row-contiguous
// epilogue: measured 134 vs 110 us)
__global__ __launch_bounds__(256) void asp_hidden_act_kernel(half_t* zh, const float* row_bias, const float* scale, const float* shift,
                                                             int64_t total8, int T, int A8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / A8;
        const int a8 = (int)(i - row * A8);
        const int b = (int)(row / T);
        const half8v v = *reinterpret_cast<const half8v*>(zh + i * 8);
        const float* rb = row_bias + ((int64_t)b * A8 + a8) * 8;
        half8v o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4v r4 = *reinterpret_cast<const float4v*>(rb + 4 * q);
            const float4v sc = *reinterpret_cast<const float4v*>(scale + a8 * 8 + 4 * q), sh = *reinterpret_cast<const float4v*>(shift + a8 * 8 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * q + e] = (half_t)tanhf(fmaxf((float)v[4 * q + e] + r4[e], 0.0f) * sc[e] + sh[e]);
        }
        *reinterpret_cast<half8v*>(zh + i * 8) = o;
    }
}

// y[n, c] = fp16(relu(x[n, c] * scale[c] + shift[c])) for channel-last fp16 rows: a pre-activation (BatchNorm + ReLU in front of a 1x1 conv,
// campplus.py:139-141 / 186-189) written out once so that the conv behind it can take the direct global -> LDS path.  Same rounding point as the
// conv kernels' transform-on-load (fp32 affine + ReLU, then one rounding to fp16 with saturation).
__global__ __launch_bounds__(256) void bn_relu_rows_kernel(const half_t* x, int64_t ldx, const float* scale, const float* shift, half_t* y,
                                                           int64_t ldy, int64_t n_rows, int C8) {
    const int64_t total = n_rows * C8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / C8;
        const int c = (int)(i - n * C8) * 8;
        const half8v v = *reinterpret_cast<const half8v*>(x + n * ldx + c);
        half8v o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4v sc = *reinterpret_cast<const float4v*>(scale + c + 4 * q), sh = *reinterpret_cast<const float4v*>(shift + c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * q + e] = (half_t)fminf(fmaxf((float)v[4 * q + e] * sc[e] + sh[e], 0.0f), 65504.0f);
        }
        *reinterpret_cast<half8v*>(y + n * ldy + c) = o;
    }
}

int bn_relu_rows_launch(const half_t* x, int64_t ldx, const float* scale, const float* shift, half_t* y, int64_t ldy, int64_t n_rows, int C,
                        hipStream_t stream) {
    MV_REQUIRE(x != nullptr && y != nullptr && scale != nullptr && shift != nullptr, "bn_relu_rows: null tensor");
    MV_REQUIRE(n_rows > 0 && C > 0 && C % 8 == 0 && ldx > 0 && ldy > 0 && ldx % 8 == 0 && ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(shift) & 15) == 0,
               "bn_relu_rows: rows and parameters must be 16-byte aligned, channels a multiple of 8");
    const int64_t total = n_rows * (C / 8);
    const int grid = ew_grid(total);
    MV_LAUNCH(bn_relu_rows_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, x, ldx, scale, shift, y, ldy, n_rows, C / 8);
    return check_launch("bn_relu_rows_kernel");
}

int asp_hidden_act_launch(half_t* zh, const float* row_bias, const float* scale, const float* shift, int B, int T, int A, hipStream_t stream) {
    MV_REQUIRE(zh != nullptr && row_bias != nullptr && scale != nullptr && shift != nullptr, "asp_hidden_act: null tensor");
    MV_REQUIRE(B > 0 && T > 0 && A > 0 && A % 8 == 0, "asp_hidden_act: bad geometry");
    const int64_t total8 = (int64_t)B * T * (A / 8);
    const int grid = ew_grid(total8);
    MV_LAUNCH(asp_hidden_act_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, zh, row_bias, scale, shift, total8, T, A / 8);
    return check_launch("asp_hidden_act_kernel");
}

// ------------------------------------------------------------------------------------------------
// Attentive statistics pooling tail.  Workgroup = (64-channel tile, 4 utterances), one utterance per wave, 16-row time
// tiles.  Per tile:  logits[c, t] = W2[c, :] . h[b, t, :]  (+ b2[c], which is constant over t and cancels in the softmax
// over time, so it is never added)  on MFMA 16x16x32 (A = W2 rows staged once in LDS,
// B = h rows loaded straight into fragment layout).  The W2 rows are fed in the order
//     MFMA tile mi, row i  <->  channel 16*(i>>2) + 4*mi + (i&3)
// so the 16 logits a lane ends up with (4 tiles x 4 rows) are 16 CONSECUTIVE channels of one time step: its x values
// are two 16-byte loads and a 16-lane group reads whole 128-byte lines.
// W2 arrives pre-multiplied by log2(e) (AspLayer::create), so the softmax weight is one v_exp_f32.
// ONE pass over x with the sums
//   s0 = sum e,  s1 = sum e*(x - g),  s2 = sum e*(x - g)^2        (g = global channel mean: keeps the second moment
//   well conditioned and makes a constant channel's variance exactly zero)
//   mean = g + s1/s0,  std = sqrt(clamp(s2/s0 - (s1/s0)^2, 1e-12))      (pooling.py:91-94,122-125)
// Two forms of e:
//   NOMAX  e = 2^logit.  h = tanh(.) is bounded by 1, so |logit| <= sum_k |W2[c,k]|; when that bound (in log2
//          units) is <= 60 for every channel -- checked once at create; real checkpoints sit near 10-15 -- nothing can
//          overflow or vanish in fp32 and the softmax needs no max subtraction at all: 10 VALU operations per element;
//   online running max m per lane and channel, e = 2^(logit - m), sums rescaled when m grows (any weights).
// The [B, 9C, T] attention input and the [B, C, T] logits of the reference never exist.
struct AspArgs {
    const half_t* h;    // [B, T, A]
    const half_t* w2;   // packed [C_pad][1][A_pad], times log2(e)
    const half_t* x;    // [B, T, ldx]
    int64_t ldx;
    const float* gmean; // [B, gmean_ld] or null
    int64_t gmean_ld;
    float* out;         // [B, 2C]: mean | std
    int B, T, C, C_pad, A, A_pad;
    float eps;
};

__device__ __forceinline__ float asp_exp2(float v) { return exp2_fast(v); }  // v_exp_f32: arguments are <= 0 (online) or within +-60 (NOMAX)

__device__ __forceinline__ void asp_merge(float& m, float& s0, float& s1, float& s2, float om, float o0, float o1, float o2) {
    const float nm = fmaxf(m, om);
    const float ra = asp_exp2(m - nm), rb = asp_exp2(om - nm);
    s0 = s0 * ra + o0 * rb;
    s1 = s1 * ra + o1 * rb;
    s2 = s2 * ra + o2 * rb;
    m = nm;
}

template <int KS, bool NOMAX>
__global__ __launch_bounds__(256) void asp_pool_kernel(AspArgs a) {
    // Workgroup = (64-channel tile, 4 utterances): the W2 tile is staged once and shared, then every WAVE pools one whole
    // utterance on its own -- no barrier and no cross-wave merge after the prologue.
    __shared__ __attribute__((aligned(16))) half_t wlds[64 * KS * 32];  // W2 tile, fragment-major: [mi][kk][lane][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y * 4 + wave;
    const int c0 = blockIdx.x * 64;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntiles = (a.T + 15) / 16;

    // stage the 64 x A_pad weight tile in MFMA A-fragment order: slot (mi, kk, lane): A row i = lane & 15 is channel
    // c0 + 16*(i>>2) + 4*mi + (i&3), k = kk*32 + 8*(lane>>4)
    for (int i = tid; i < 4 * KS * 64; i += 256) {
        const int l = i & 63, kk = (i >> 6) % KS, mi = i / (64 * KS);
        const int ar = l & 15;
        const int row = c0 + 16 * (ar >> 2) + 4 * mi + (ar & 3);
        half8v v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)0.0f;
        if (row < a.C_pad) v = *reinterpret_cast<const half8v*>(a.w2 + (int64_t)row * a.A_pad + kk * 32 + 8 * (l >> 4));
        *reinterpret_cast<half8v*>(wlds + (size_t)i * 8) = v;
    }
    __syncthreads();
    if (b >= a.B) return;
    const half_t* hb = a.h + (int64_t)b * a.T * a.A;
    const half_t* xb = a.x + (int64_t)b * a.T * a.ldx;

    // this lane's 16 channels: c0 + 16*fg + 4*mi + r
    const int cl = c0 + 16 * fg;
    // (the bias b2[c] is constant over time, so it cancels in the softmax over time: it is never added)
    float4v g4[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cl + 4 * mi + r;
            g4[mi][r] = (a.gmean != nullptr && c < a.C) ? a.gmean[(int64_t)b * a.gmean_ld + c] : 0.0f;
        }
    }
    const bool xvec = c0 + 64 <= a.C && (a.ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;

    auto load_h = [&](int t0, half8v (&hf)[KS]) {
        const int t = t0 + fr < a.T ? t0 + fr : a.T - 1;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int k = kk * 32 + 8 * fg;
            if (k + 8 <= a.A) {
                hf[kk] = *reinterpret_cast<const half8v*>(hb + (int64_t)t * a.A + k);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) hf[kk][e] = (k + e < a.A) ? hb[(int64_t)t * a.A + k + e] : (half_t)0.0f;
            }
        }
    };
    auto load_x = [&](int t0, half8v (&xv)[2]) {
        const int t = t0 + fr < a.T ? t0 + fr : a.T - 1;
        const half_t* p = xb + (int64_t)t * a.ldx + cl;
        if (xvec) {
            xv[0] = *reinterpret_cast<const half8v*>(p);
            xv[1] = *reinterpret_cast<const half8v*>(p + 8);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) xv[e >> 3][e & 7] = (cl + e < a.C) ? p[e] : (half_t)0.0f;
        }
    };

    float m[4][4], s0[4][4], s1[4][4], s2[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m[mi][r] = -3.0e38f;
            s0[mi][r] = s1[mi][r] = s2[mi][r] = 0.0f;
        }
    {
        half8v hcur[KS], hnext[KS];
        half8v xcur[2], xnext[2];
        load_h(0, hcur);
        load_x(0, xcur);
        for (int tt = 0; tt < ntiles; ++tt) {
            const bool more = tt + 1 < ntiles;
            if (more) {  // next tile's rows are in flight while this tile computes
                load_h((tt + 1) * 16, hnext);
                load_x((tt + 1) * 16, xnext);
            }
            // rows beyond T (last tile only) get a logit of -inf: e = 0, they add nothing
            const float voff = tt * 16 + fr < a.T ? 0.0f : -INFINITY;
            // the W2 fragments are re-read from LDS every tile (the opaque copy keeps the compiler from parking all
            // 16 of them in 64 VGPRs, which would halve the number of resident waves)
            const lds_half_ptr wl = lds_opaque_half_ptr(wlds + lane * 8);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float4v l = float4v{voff, voff, voff, voff};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const half8v wf = lds_load_half8(wl, (mi * KS + kk) * 512);
                    l = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, hcur[kk], l, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 4 * mi + r;
                    const float d = (float)xcur[ci >> 3][ci & 7] - g4[mi][r];
                    if (NOMAX) {
                        const float e = asp_exp2(l[r]);
                        const float ed = e * d;
                        s0[mi][r] += e;
                        s1[mi][r] += ed;
                        s2[mi][r] = fmaf(ed, d, s2[mi][r]);
                    } else {
                        const float nm = fmaxf(m[mi][r], l[r]);  // finite from the first valid row on; -3e38 before
                        const float rs = asp_exp2(m[mi][r] - nm);
                        const float e = asp_exp2(l[r] - nm);
                        const float ed = e * d;
                        s0[mi][r] = s0[mi][r] * rs + e;
                        s1[mi][r] = s1[mi][r] * rs + ed;
                        s2[mi][r] = s2[mi][r] * rs + ed * d;
                        m[mi][r] = nm;
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) hcur[kk] = hnext[kk];
                xcur[0] = xnext[0];
                xcur[1] = xnext[1];
            }
        }
    }
    // ---- merge the 16 time lanes that share channels; lane fr == 0 writes its 16 channels ----
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (NOMAX) {
                s0[mi][r] = row16_sum(s0[mi][r]);
                s1[mi][r] = row16_sum(s1[mi][r]);
                s2[mi][r] = row16_sum(s2[mi][r]);
            } else {
#pragma unroll
                for (int sh = 1; sh <= 8; sh <<= 1) {
                    const float om = __shfl_xor(m[mi][r], sh), o0 = __shfl_xor(s0[mi][r], sh);
                    const float o1 = __shfl_xor(s1[mi][r], sh), o2 = __shfl_xor(s2[mi][r], sh);
                    asp_merge(m[mi][r], s0[mi][r], s1[mi][r], s2[mi][r], om, o0, o1, o2);
                }
            }
            const int c = cl + 4 * mi + r;
            if (fr == 0 && c < a.C) {
                const float m1 = s1[mi][r] / s0[mi][r];
                const float var = s2[mi][r] / s0[mi][r] - m1 * m1;
                a.out[(int64_t)b * 2 * a.C + c] = g4[mi][r] + m1;
                a.out[(int64_t)b * 2 * a.C + a.C + c] = sqrtf(fmaxf(var, a.eps));
            }
        }
}

// ---- ring form (round 2): the shipped shapes (whole 64-channel tiles, 16-byte aligned rows, A = 64 or 128, bounded logits).
// Workgroup = (ONE utterance, four ADJACENT 64-channel tiles), one tile per wave, no barrier:
//   * the four waves walk the same rows at the same pace, so the workgroup reads 512 contiguous bytes of every x row (the register
//     form has four utterances per workgroup: isolated 128-byte segments 6 KiB apart);
//   * BOTH streams of a wave -- its x rows (16 x 128 B per tile) and the h rows (16 x 2A B) -- travel global -> LDS directly into
//     PRIVATE rings (x: four tiles, three ahead; h: three tiles, two ahead): no load registers, up to 14 KiB in flight per wave, one
//     counted s_waitcnt per tile.  The transfers are issued from inline assembly (glds16_untracked): with the builtin -- or with h
//     in ordinary registers -- the compiler adds its own waits, which either drain the rings (vmcnt(0) in front of every LDS load
//     that may alias a transfer) or force half of the just-requested h rows to land (it counts only the loads it knows about):
//     measured 185-195 us for every such variant against 211-225 us for the register form;
//   * the W2 tile of the wave (64 x A fp16) lives in REGISTERS as MFMA A fragments for the whole utterance;
//   * the element math runs on float2 (v_pk_add/mul/fma_f32): 2 cvt + 2 exp + 5 packed operations per channel pair.
// LDS tiles: x 16 rows x 128 B, 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7); h 16 rows x 2A B, chunk c of row r
// at position c ^ (r & (2A/16 - 1)): the 16 lanes of a fragment group read 16 different bank groups.
// Ring depth (round 6, r15j/k: the kernel alone at 256 x 3 s): 4 tiles = 48 KiB per workgroup = three workgroups per CU 149-158 us, 6 tiles (rounds 2-5: 72 KiB, two per
// CU) 158-163, 5 tiles 152-158, 8 tiles (96 KiB, one per CU) 228-234: occupancy beats prefetch depth.
constexpr int ASP_RING = 4;                          // tiles per ring: a step requests h(tt + ASP_RING - 1) and x(tt + ASP_RING)
constexpr int ASP_XRING = ASP_RING, ASP_HRING = ASP_RING;
constexpr int ASP_YOUNGER = 2 + 3 * (ASP_RING - 2);  // transfers of a wave younger than its part of h(tt + 1) when step tt waits (8 for four tiles, 14 for six)

template <int KS>
__global__ __launch_bounds__(256) void asp_pool_ring_kernel(AspArgs a) {
    constexpr int HROW = KS * 64;          // bytes per h row (A_pad fp16)
    constexpr int HTILE = 16 * HROW;       // 2 or 4 KiB
    constexpr int HTR = HTILE / 1024;      // transfers per h tile
    constexpr int HCH = HROW / 16;         // 16-byte chunks per h row (8 or 16)
    MV_DYN_SMEM(smem);  // the workgroup's h ring (ASP_HRING tiles), then four private x rings (ASP_XRING tiles each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = MV_UNIFORM(tid >> 6);
    // (measured and dropped, r12m: an XCD-aware order that gives the C / 256 workgroups of one utterance the same id mod 8 -- its h rows through one
    //  L2 instead of six -- 160.2 / 160.5 / 167.5 us against 160.5 / 154.6 / 158.2 in the plain order)
    const int b = blockIdx.y;
    const int c0 = (blockIdx.x * 4 + wave) * 64;  // (C is a multiple of 256 here: every wave has a tile and reaches every barrier)
    const int fr = lane & 15, fg = lane >> 4;
    const int ntiles = (a.T + 15) / 16;
    const half_t* hb = a.h + (int64_t)b * a.T * a.A;
    const half_t* xb = a.x + (int64_t)b * a.T * a.ldx + c0;
    const int cl = c0 + 16 * fg;

    half8v wf[4][KS];   // W2 tile of this wave (loaded below, behind the first ring requests)
    float2v g2[4][2];   // global channel means of this lane's 16 channels
    const unsigned hring = MV_UNIFORM(lds_addr(smem));
    const unsigned xring = hring + (unsigned)(ASP_HRING * HTILE + wave * (ASP_XRING * 2048));
    // x transfers: lane l of transfer j (0 / 1) lands at (row 8j + (l >> 3), chunk position l & 7).  Every transfer has its own
    // running source pointer (one 64-bit add per tile instead of a clamp, a 64-bit multiply and two adds); tiles that reach beyond
    // T (the last one, and the ones the ring requests past the end) re-read row T - 1 through the slow path: the counts stay exact.
    const half_t* xsrc[2];
    int xrow[2], xch[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        xrow[j] = 8 * j + (lane >> 3);
        xch[j] = ((lane & 7) ^ ((xrow[j] >> 1) & 7)) * 8;
        xsrc[j] = xb + (int64_t)xrow[j] * a.ldx + xch[j];
    }
    const int64_t xstep = 16 * a.ldx;
    auto issue_x = [&](int tile, int slot) {
        const bool whole = tile * 16 + 16 <= a.T;  // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const half_t* src = xsrc[j];
            if (!whole) {
                const int t = tile * 16 + xrow[j];
                src = xb + (int64_t)(t < a.T ? t : a.T - 1) * a.ldx + xch[j];
            }
            glds16_untracked(src, xring + (unsigned)(slot * 2048 + j * 1024));
            xsrc[j] += xstep;
        }
    };
    // h transfers: the tile is shared by the four waves; wave w moves transfer j = w % HTR (A = 64: two transfers, moved twice --
    // identical bytes to the same place -- so that every wave issues exactly one per tile and the waits can be counted):
    // lane l lands at (row (64 / HCH) * j + l / HCH, chunk position l % HCH)
    const int hj = wave % HTR;
    const int hrow = (64 / HCH) * hj + lane / HCH;
    const int hch = ((lane % HCH) ^ (hrow & (HCH - 1))) * 8;
    const half_t* hsrc = hb + (int64_t)hrow * a.A + hch;
    const int hstep = 16 * a.A;
    auto issue_h = [&](int tile, int slot) {
        const bool whole = tile * 16 + 16 <= a.T;
        const half_t* src = hsrc;
        if (!whole) {
            const int t = tile * 16 + hrow;
            src = hb + (int64_t)(t < a.T ? t : a.T - 1) * a.A + hch;
        }
        glds16_untracked(src, hring + (unsigned)(slot * HTILE + hj * 1024));
        hsrc += hstep;
    };
    // this lane's reads: x row fr, chunks 2 fg, 2 fg + 1; h row fr, chunks 4 kk + fg
    const int sw = (fr >> 1) & 7;
    const unsigned xoff0 = (unsigned)(fr * 128 + (((2 * fg) ^ sw) << 4)), xoff1 = (unsigned)(fr * 128 + (((2 * fg + 1) ^ sw) << 4));
    unsigned hoff[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) hoff[kk] = (unsigned)(fr * HROW + (((4 * kk + fg) ^ (fr & (HCH - 1))) << 4));

    float2v s0[4][2], s1[4][2], s2[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int q = 0; q < 2; ++q) s0[mi][q] = s1[mi][q] = s2[mi][q] = float2v{0.0f, 0.0f};

    // Software pipeline: step tt requests h(tt+5) and x(tt+6), fetches tile tt+1 from the rings into registers and runs ITS MFMAs,
    // then does the element math of tile tt on the logits the previous step left behind -- the matrix pipe and the LDS round trip of a
    // tile work under the VALU block of the tile before it.  Request order = steps -6 .. -1 of the same rule: x0 | h0 x1 | h1 x2 | ...
    // | h4 x5, then step tt: h(tt+5) x(tt+6).  Younger than this wave's part of h(tt+1) (requested in step tt-4, with x(tt+2) behind
    // it) when step tt waits: 2 + 4 steps x 3 = 14 transfers; x(tt+1) is older still.  After the wait a barrier: every wave's part of
    // h(tt+1) has landed, and every wave is done reading h(tt) -- whose slot h(tt+6) overwrites one step later.
    auto fetch = [&](int tile, int hslot, int xslot, float4v (&l)[4], half8v& xv0, half8v& xv1) {
        MV_LOCKSTEP_POINT();  // (lanes read what other lanes of the wave transferred)
        const unsigned xt = xring + (unsigned)(xslot * 2048);
        const unsigned ht = hring + (unsigned)(hslot * HTILE);
        half8v hf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) lds_read1(hf[kk], ht + hoff[kk]);
        lds_read1(xv0, xt + xoff0);
        lds_read1(xv1, xt + xoff1);
        MV_LOCKSTEP_POINT();  // (... and the slots are overwritten by later transfers)
        const float voff = tile * 16 + fr < a.T ? 0.0f : -INFINITY;
        if constexpr (KS == 4) lds_wait<2>(hf[0], hf[1], hf[2], hf[3]); else lds_wait<2>(hf[0], hf[1]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            l[mi] = float4v{voff, voff, voff, voff};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) l[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][kk], hf[kk], l[mi], 0, 0, 0);
        }
        lds_wait<0>(xv0, xv1);
    };
    auto pool = [&](const float4v (&l)[4], const half8v& xv0, const half8v& xv1) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ci = 4 * mi + 2 * q;
                const half8v& xv = ci < 8 ? xv0 : xv1;
                const float2v xf = float2v{(float)xv[ci & 7], (float)xv[(ci & 7) + 1]};
                const float2v d = xf - g2[mi][q];
                const float2v e = float2v{asp_exp2(l[mi][2 * q]), asp_exp2(l[mi][2 * q + 1])};
                const float2v ed = e * d;
                s0[mi][q] += e;
                s1[mi][q] += ed;
                s2[mi][q] = __builtin_elementwise_fma(ed, d, s2[mi][q]);
            }
        }
    };
    issue_x(0, 0);
#pragma unroll
    for (int i = 0; i < ASP_RING - 1; ++i) {
        issue_h(i, i);
        issue_x(i + 1, i + 1);
    }
    // W2 tile as A fragments: tile mi, row i = fr  <->  channel c0 + 16 * (i >> 2) + 4 * mi + (i & 3), k = kk * 32 + 8 * fg
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            wf[mi][kk] = *MV_GLOBAL_PTR(half8v, a.w2 + (int64_t)(c0 + 16 * (fr >> 2) + 4 * mi + (fr & 3)) * a.A_pad + kk * 32 + 8 * fg);
    {
        float4v g4[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            g4[mi] = a.gmean != nullptr ? *MV_GLOBAL_PTR(float4v, a.gmean + (int64_t)b * a.gmean_ld + cl + 4 * mi) : float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            g2[mi][0] = float2v{g4[mi][0], g4[mi][1]};
            g2[mi][1] = float2v{g4[mi][2], g4[mi][3]};
        }
    }
    // (The ring requests above are already in flight: the latency of these loads and of the first tiles overlap.)
    // The loads above are the only ones the compiler tracks.  Touch every register they deliver: the compiler then waits for them HERE
    // (s_waitcnt vmcnt(0) in front of the first touch) instead of re-checking them inside the loop on every iteration, where its
    // vmcnt(0) would drain the rings.
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) MV_OPAQUE(wf[mi][kk]);
        MV_OPAQUE(g2[mi][0]);
        MV_OPAQUE(g2[mi][1]);
    }

    float4v la[4], lb[4];
    half8v xa0, xa1, xb0, xb1;
    wait_vm<ASP_YOUNGER>();
    lds_barrier();
    fetch(0, 0, 0, la, xa0, xa1);
    int hs = ASP_RING - 1, xs = 0;  // ring slots of h(tt+5) and x(tt+6)
    auto step = [&](int tt, float4v (&lcur)[4], half8v& xc0, half8v& xc1, float4v (&lnext)[4], half8v& xn0, half8v& xn1) {
        issue_h(tt + ASP_RING - 1, hs);
        issue_x(tt + ASP_RING, xs);
        wait_vm<ASP_YOUNGER>();
        lds_barrier();
        hs = hs == ASP_HRING - 1 ? 0 : hs + 1;   // now the slot of h(tt)  ... and of h(tt+6) next step
        xs = xs == ASP_XRING - 1 ? 0 : xs + 1;   // the slot of x(tt+1)   ... and of x(tt+7) next step
        int h1 = hs + 1;                         // slot of h(tt+1)
        h1 = h1 == ASP_HRING ? 0 : h1;
        fetch(tt + 1, h1, xs, lnext, xn0, xn1);
        pool(lcur, xc0, xc1);
    };
    int tt = 0;
#pragma unroll 1
    for (; tt + 1 < ntiles; tt += 2) {  // the two register sets trade roles: no copies
        step(tt, la, xa0, xa1, lb, xb0, xb1);
        step(tt + 1, lb, xb0, xb1, la, xa0, xa1);
    }
    if (tt < ntiles) pool(la, xa0, xa1);   // odd tile count: the last tile's logits are in the first set
    wait_vm<0>();  // the rings belong to the workgroup's LDS allocation: nothing may still be landing when the wave ends
    // 48 sums over the 16 time lanes of a row, then mean / std by lane fr == 0 (v_rcp / v_sqrt: 1 ulp, far inside the 2e-5 bar)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z0 = row16_sum(s0[mi][r >> 1][r & 1]);
            const float z1 = row16_sum(s1[mi][r >> 1][r & 1]);
            const float z2 = row16_sum(s2[mi][r >> 1][r & 1]);
            if (fr == 0) {
                const int c = cl + 4 * mi + r;
                const float inv = rcp_fast(z0);
                const float m1 = z1 * inv;
                const float var = z2 * inv - m1 * m1;
                a.out[(int64_t)b * 2 * a.C + c] = g2[mi][r >> 1][r & 1] + m1;
                a.out[(int64_t)b * 2 * a.C + a.C + c] = sqrt_fast(fmaxf(var, a.eps));
            }
        }
}

template <int KS>
static void asp_pool_dispatch(const AspArgs& a, unsigned gx, unsigned gy, bool nomax, hipStream_t stream) {
    if (nomax) {
        MV_LAUNCH((asp_pool_kernel<KS, true>), (gx, gy, 1), (256, 1, 1), 0, stream, a);
    } else {
        MV_LAUNCH((asp_pool_kernel<KS, false>), (gx, gy, 1), (256, 1, 1), 0, stream, a);
    }
}

// w2_packed carries the factor log2(e); logit_bound_log2 = max_c sum_k |W2[c,k]| * log2(e), < 0 = unknown
int asp_pool_launch(const half_t* h, const half_t* w2_packed, const half_t* x, int64_t ldx, const float* gmean,
                    int64_t gmean_ld, float* out, int B, int T, int C, int A, float logit_bound_log2, hipStream_t stream) {
    MV_REQUIRE(h != nullptr && w2_packed != nullptr && x != nullptr && out != nullptr, "asp_pool: null tensor");
    MV_REQUIRE(B > 0 && T > 0 && C > 0 && A > 0, "asp_pool: bad geometry");
    MV_REQUIRE(A % 8 == 0 && A <= 256, "asp_pool: attention width must be a multiple of 8 and <= 256");
    MV_REQUIRE(ldx % 4 == 0 && C % 4 == 0, "asp_pool: channels must be a multiple of 4");
    MV_REQUIRE(ldx > 0 && (gmean == nullptr || gmean_ld >= 0), "asp_pool: leading dimensions must not be negative");
    AspArgs a;
    a.h = h;
    a.w2 = w2_packed;
    a.x = x;
    a.ldx = ldx;
    a.gmean = gmean;
    a.gmean_ld = gmean_ld;
    a.out = out;
    a.B = B;
    a.T = T;
    a.C = C;
    a.A = A;
    a.C_pad = (int)round_up(C, 32);
    a.A_pad = (int)round_up(A, 64);
    a.eps = 1e-12f;
    const unsigned gx = (unsigned)ceil_div(C, 64);
    const bool nomax = logit_bound_log2 >= 0.0f && logit_bound_log2 <= 60.0f;
    // ring form: whole tiles, aligned rows, no partial h fragments (other widths / unbounded logits: the register form below)
    const bool ring_ok = nomax && C % 256 == 0 && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && A == a.A_pad &&
                         (gmean == nullptr || (gmean_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(gmean) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(h) & 15) == 0;
    if (ring_ok && (a.A_pad == 64 || a.A_pad == 128)) {
        const unsigned gxw = (unsigned)ceil_div(C, 256);
        if (a.A_pad == 64) {
            MV_LAUNCH((asp_pool_ring_kernel<2>), (gxw, (unsigned)B, 1), (256, 1, 1), 4 * ASP_XRING * 2048 + ASP_HRING * 2048, stream, a);
        } else {
            MV_LAUNCH((asp_pool_ring_kernel<4>), (gxw, (unsigned)B, 1), (256, 1, 1), 4 * ASP_XRING * 2048 + ASP_HRING * 4096, stream, a);
        }
        return check_launch("asp_pool_ring_kernel");
    }
    switch (a.A_pad / 32) {
        case 2: asp_pool_dispatch<2>(a, gx, (unsigned)ceil_div(B, 4), nomax, stream); break;
        case 4: asp_pool_dispatch<4>(a, gx, (unsigned)ceil_div(B, 4), nomax, stream); break;
        case 6: asp_pool_dispatch<6>(a, gx, (unsigned)ceil_div(B, 4), nomax, stream); break;
        default: asp_pool_dispatch<8>(a, gx, (unsigned)ceil_div(B, 4), nomax, stream); break;
    }
    return check_launch("asp_pool_kernel");
}

}  // namespace mv

extern "C" {

int mv_asp_pool_f16(const void* h, const void* w2_packed, const void* x, int64_t ldx, const float* gmean, int64_t gmean_ld,
                    float* out, int32_t B, int32_t T, int32_t C, int32_t A, float logit_bound_log2, mv_stream_t stream) {
    return mv::asp_pool_launch(reinterpret_cast<const half_t*>(h), reinterpret_cast<const half_t*>(w2_packed),
                               reinterpret_cast<const half_t*>(x), ldx, gmean, gmean_ld, out, B, T, C, A, logit_bound_log2,
                               static_cast<hipStream_t>(stream));
}

int mv_bn_relu_rows_f16(const void* x, int64_t ldx, const float* scale, const float* shift, void* y, int64_t ldy, int64_t n_rows,
                        int32_t C, mv_stream_t stream) {
    return mv::bn_relu_rows_launch(reinterpret_cast<const half_t*>(x), ldx, scale, shift, reinterpret_cast<half_t*>(y), ldy, n_rows, C,
                                   static_cast<hipStream_t>(stream));
}

int mv_time_stats_f16(const void* x, int64_t ld, int32_t B, int32_t T, int32_t C, float* mean, float* std,
                      int32_t unbiased, float clamp_eps, mv_stream_t stream) {
    return mv::time_stats_launch(reinterpret_cast<const half_t*>(x), ld, B, T, C, mean, std, C, unbiased, clamp_eps,
                                 static_cast<hipStream_t>(stream), nullptr, nullptr);
}

}  // extern "C"
