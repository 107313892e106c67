// Implicit-GEMM 1-D convolution on MFMA (fp16 inputs, fp32 accumulate) with a fused epilogue.
//
// One kernel serves every dense contraction of the backbones:
//   * TDNNBlock = conv -> ReLU -> BatchNorm (mvector/models/utils.py:115-138), with the reflect "same"
//     padding of the Conv1d wrapper (utils.py:98-103) done as index mirroring in the loader;
//   * Res2Net steps (ecapa_tdnn.py:39-51): second input added on load, channel slices addressed by
//     pointer offset + leading dimension, so torch.chunk / torch.cat never materialise;
//   * the ASP projections (pooling.py:80-84,117) incl. the per-utterance context bias;
//   * CAM++ TDNN / dense layers (campplus.py:41-68,114-150): zero padding, stride, pre-activation
//     BatchNorm+ReLU applied to the input on load, context gate multiplied in the epilogue.
//
// Layout: activations are channel-last fp16 [B, T, C] (C contiguous), so one time step is one GEMM row
// whose K elements are contiguous; weights are pre-packed fp16 [Cout_pad][tap][Cin_pad].  The GEMM is
//   D[co, n] = sum_{tap, ci} W[co, tap, ci] * X[row(n, tap), ci],   n = b*T_out + t,
// with W as the MFMA A operand and X as the B operand, so that each lane ends up with 4 consecutive
// output channels of one time step (one 8-byte fp16 store).
//
// Tile: 128 (co) x 128 (n) x 64 (K) per 256-thread workgroup, 2 x 2 waves, 4 x 4 MFMA 16x16x32 tiles per
// wave; global -> registers -> XOR-swizzled LDS (conflict-free ds_read_b128), double buffered, one barrier
// per K step.  Workgroups are numbered so that all co-tiles of one n-tile run on the same XCD (L2 reuse
// of the activation tile; the weights are shared by everyone).
#include "common.h"

namespace mv {

constexpr int CV_TC = 128;  // output channels per tile
constexpr int CV_TN = 128;  // time steps per tile
constexpr int CV_BK = 64;   // K elements per stage
constexpr int CV_THREADS = 256;
constexpr int CV_LDS_BYTES = 2 * (CV_TC + CV_TN) * CV_BK * 2;  // 64 KiB

struct ConvArgs {
    const void* x;
    const void* x2;
    int64_t ldx, ldx2;
    const float* in_scale;
    const float* in_shift;
    const half_t* w;
    const float* bias;
    const float* row_bias;
    const float* scale;
    const float* shift;
    const float* gate;
    void* y;
    int64_t ldy;
    int B, T_in, T_out, cin, cin_pad, cout, cout_pad, k, dil, stride, pad, pad_mode;
    int pre_act, post_act, y_f16, gate_seg_len, gate_nseg;
    int n_rows, n_tiles, co_tiles;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MV_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MV_ACT_TANH) return tanhf(v);
    if (act == MV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][64] fp16 tile
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <typename InT>
struct RawChunk;
template <>
struct RawChunk<half_t> {
    half8v v;
};
template <>
struct RawChunk<float> {
    float4v lo, hi;
};

template <typename InT>
__device__ __forceinline__ void load_raw(RawChunk<InT>& r, const InT* p, int nvalid);

template <>
__device__ __forceinline__ void load_raw<half_t>(RawChunk<half_t>& r, const half_t* p, int nvalid) {
    if (nvalid >= 8) {
        r.v = *reinterpret_cast<const half8v*>(p);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[e] = e < nvalid ? p[e] : (half_t)0.0f;
    }
}
template <>
__device__ __forceinline__ void load_raw<float>(RawChunk<float>& r, const float* p, int nvalid) {
    if (nvalid >= 8) {
        r.lo = *reinterpret_cast<const float4v*>(p);
        r.hi = *reinterpret_cast<const float4v*>(p + 4);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r.lo[e] = e < nvalid ? p[e] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) r.hi[e] = (e + 4) < nvalid ? p[e + 4] : 0.0f;
    }
}

__device__ __forceinline__ void zero_raw(RawChunk<half_t>& r) {
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = (half_t)0.0f;
}
__device__ __forceinline__ void zero_raw(RawChunk<float>& r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r.lo[e] = 0.0f;
        r.hi[e] = 0.0f;
    }
}

__device__ __forceinline__ float raw_get(const RawChunk<half_t>& r, int e) { return (float)r.v[e]; }
__device__ __forceinline__ float raw_get(const RawChunk<float>& r, int e) { return e < 4 ? r.lo[e] : r.hi[e - 4]; }

__device__ __forceinline__ half_t to_half_sat(float v) {
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);  // saturate instead of producing inf
    return (half_t)v;
}

template <typename InT>
__global__ __launch_bounds__(CV_THREADS) void conv1d_mfma_kernel(ConvArgs a) {
    MV_DYN_SMEM(smem);
    // ---- XCD-aware tile assignment: workgroup id -> (n_tile, co_tile) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int seq = bid >> 3;
    const int n_tile = xcd + 8 * (seq / a.co_tiles);
    const int co_tile = seq % a.co_tiles;
    if (n_tile >= a.n_tiles) return;  // whole workgroup leaves before any barrier

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1;  // wave position along co
    const int wn = wave & 1;   // wave position along n
    const int n0 = n_tile * CV_TN;
    const int co0 = co_tile * CV_TC;

    // ---- loader mapping: thread owns 16-byte chunk kc of rows lrow + 32*i ----
    const int kc = tid & 7;
    const int lrow = tid >> 3;
    int xb[4], xt[4];  // batch index / output time of the 4 activation rows of this thread (-1: out of range)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + lrow + 32 * i;
        if (n < a.n_rows) {
            xb[i] = n / a.T_out;
            xt[i] = n - xb[i] * a.T_out;
        } else {
            xb[i] = -1;
            xt[i] = 0;
        }
    }
    const InT* xbase = reinterpret_cast<const InT*>(a.x);
    const InT* x2base = reinterpret_cast<const InT*>(a.x2);
    const int kstages_per_tap = a.cin_pad / CV_BK;
    const int nstages = a.k * kstages_per_tap;

    RawChunk<InT> xr[4], x2r[4];
    half8v wr[4];

    auto issue_loads = [&](int s) {
        const int tap = s / kstages_per_tap;
        const int c = (s - tap * kstages_per_tap) * CV_BK + kc * 8;
        const int nvalid = a.cin - c;  // channels of this chunk that exist
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool ok = xb[i] >= 0 && nvalid > 0;
            int tin = xt[i] * a.stride - a.pad + tap * a.dil;
            if (tin < 0 || tin >= a.T_in) {
                if (a.pad_mode == MV_PAD_REFLECT) {
                    tin = tin < 0 ? -tin : 2 * (a.T_in - 1) - tin;
                } else {
                    ok = false;
                }
            }
            if (ok) {
                const int64_t row = (int64_t)xb[i] * a.T_in + tin;
                load_raw<InT>(xr[i], xbase + row * a.ldx + c, nvalid);
                if (x2base != nullptr) load_raw<InT>(x2r[i], x2base + row * a.ldx2 + c, nvalid);
            } else {
                zero_raw(xr[i]);
                if (x2base != nullptr) zero_raw(x2r[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + lrow + 32 * i;
            if (co < a.cout_pad) {
                wr[i] = *reinterpret_cast<const half8v*>(a.w + ((int64_t)co * a.k + tap) * a.cin_pad + c);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wr[i][e] = (half_t)0.0f;
            }
        }
    };

    auto store_lds = [&](int s, int buf) {
        char* wt = smem + buf * ((CV_TC + CV_TN) * CV_BK * 2);
        char* xtile = wt + CV_TC * CV_BK * 2;
        const int tap = s / kstages_per_tap;
        const int c = (s - tap * kstages_per_tap) * CV_BK + kc * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lrow + 32 * i;
            half8v hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = raw_get(xr[i], e);
                if (x2base != nullptr) v += raw_get(x2r[i], e);
                if (a.in_scale != nullptr) {
                    const int ch = c + e;
                    if (ch < a.cin) v = fmaxf(v * a.in_scale[ch] + a.in_shift[ch], 0.0f);
                    // rows that are zero padding must stay zero AFTER the pre-activation (the reference pads
                    // the already-activated tensor): handled by the caller-visible rule below
                }
                hv[e] = to_half_sat(v);
            }
            *reinterpret_cast<half8v*>(xtile + lds_off(row, kc)) = hv;
            *reinterpret_cast<half8v*>(wt + lds_off(row, kc)) = wr[i];
        }
    };

    float4v acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};

    issue_loads(0);
    store_lds(0, 0);
    __syncthreads();

    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstages) issue_loads(s + 1);
        const char* wt = smem + buf * ((CV_TC + CV_TN) * CV_BK * 2);
        const char* xtile = wt + CV_TC * CV_BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8v af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const half8v*>(wt + lds_off(wc * 64 + mi * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                bf[ni] = *reinterpret_cast<const half8v*>(xtile + lds_off(wn * 64 + ni * 16 + frow, kk * 4 + fchunk));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (s + 1 < nstages) store_lds(s + 1, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels co..co+3 (rows) of time step n (column) ----
    const int crow = 4 * (lane >> 4);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + (lane & 15);
        if (n >= a.n_rows) continue;
        int b = 0, t = 0;
        if (a.row_bias != nullptr || a.gate != nullptr) {
            b = n / a.T_out;
            t = n - b * a.T_out;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int co = co0 + wc * 64 + mi * 16 + crow;
            if (co >= a.cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = co + r;
                float x = acc[mi][ni][r];
                if (c < a.cout) {
                    if (a.bias != nullptr) x += a.bias[c];
                    if (a.row_bias != nullptr) x += a.row_bias[(int64_t)b * a.cout + c];
                    x = apply_act(x, a.pre_act);
                    if (a.scale != nullptr) x = x * a.scale[c] + a.shift[c];
                    x = apply_act(x, a.post_act);
                    if (a.gate != nullptr) x *= a.gate[((int64_t)b * a.gate_nseg + t / a.gate_seg_len) * a.cout + c];
                }
                v[r] = x;
            }
            if (a.y_f16) {
                half_t* yp = reinterpret_cast<half_t*>(a.y) + (int64_t)n * a.ldy + co;
                if (co + 3 < a.cout) {
                    half4v hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = to_half_sat(v[r]);
                    *reinterpret_cast<half4v*>(yp) = hv;
                } else {
                    for (int r = 0; r < 4 && co + r < a.cout; ++r) yp[r] = to_half_sat(v[r]);
                }
            } else {
                float* yp = reinterpret_cast<float*>(a.y) + (int64_t)n * a.ldy + co;
                if (co + 3 < a.cout) {
                    *reinterpret_cast<float4v*>(yp) = float4v{v[0], v[1], v[2], v[3]};
                } else {
                    for (int r = 0; r < 4 && co + r < a.cout; ++r) yp[r] = v[r];
                }
            }
        }
    }
}

// fp32 [Cout][Cin][k] -> fp16 [Cout_pad][k][Cin_pad], zero padded
__global__ void pack_conv_weight_kernel(const float* w, int cout, int cin, int k, int cout_pad, int cin_pad,
                                        half_t* out) {
    const int64_t total = (int64_t)cout_pad * k * cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin_pad);
        const int j = (int)((i / cin_pad) % k);
        const int co = (int)(i / ((int64_t)cin_pad * k));
        float v = 0.0f;
        if (co < cout && ci < cin) v = w[((int64_t)co * cin + ci) * k + j];
        out[i] = to_half_sat(v);
    }
}

int conv1d_cin_pad(int cin) { return (int)round_up(cin, CV_BK); }
int conv1d_cout_pad(int cout) { return (int)round_up(cout, 32); }

int conv1d_launch(const MvConv1dDesc& d, hipStream_t stream) {
    MV_REQUIRE(d.x != nullptr && d.w_packed != nullptr && d.y != nullptr, "conv1d: null tensor");
    MV_REQUIRE(d.B > 0 && d.T_in > 0 && d.T_out > 0 && d.cin > 0 && d.cout > 0 && d.k > 0, "conv1d: bad geometry");
    MV_REQUIRE(d.dilation >= 1 && d.stride >= 1 && d.pad >= 0, "conv1d: bad dilation/stride/pad");
    MV_REQUIRE((int64_t)(d.T_out - 1) * d.stride - d.pad + (int64_t)(d.k - 1) * d.dilation < d.T_in + d.pad,
               "conv1d: T_out reaches beyond the padded input");
    if (d.pad_mode == MV_PAD_REFLECT) MV_REQUIRE(d.pad < d.T_in, "conv1d: reflect padding needs pad < T_in");
    MV_REQUIRE(d.x_dtype == MV_DT_F16 || d.x_dtype == MV_DT_F32, "conv1d: x dtype");
    MV_REQUIRE(d.y_dtype == MV_DT_F16 || d.y_dtype == MV_DT_F32, "conv1d: y dtype");
    MV_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "conv1d: in_scale/in_shift go together");
    MV_REQUIRE((d.scale == nullptr) == (d.shift == nullptr), "conv1d: scale/shift go together");
    // vector-access contract of the loader / epilogue
    const int xalign = d.x_dtype == MV_DT_F16 ? 8 : 4;
    MV_REQUIRE(d.ldx % xalign == 0 && (reinterpret_cast<uintptr_t>(d.x) & 15) == 0, "conv1d: x must be 16-byte aligned per row");
    if (d.x2 != nullptr)
        MV_REQUIRE(d.ldx2 % xalign == 0 && (reinterpret_cast<uintptr_t>(d.x2) & 15) == 0, "conv1d: x2 alignment");
    const int yalign_bytes = d.y_dtype == MV_DT_F16 ? 8 : 16;
    MV_REQUIRE(d.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(d.y) & (yalign_bytes - 1)) == 0, "conv1d: y alignment");
    if (d.gate != nullptr) MV_REQUIRE(d.gate_seg_len > 0, "conv1d: gate needs a segment length");
    MV_REQUIRE((int64_t)d.B * d.T_out < ((int64_t)1 << 31) - CV_TN, "conv1d: too many rows for 32-bit indexing");

    ConvArgs a;
    a.x = d.x;
    a.x2 = d.x2;
    a.ldx = d.ldx;
    a.ldx2 = d.ldx2;
    a.in_scale = d.in_scale;
    a.in_shift = d.in_shift;
    a.w = reinterpret_cast<const half_t*>(d.w_packed);
    a.bias = d.bias;
    a.row_bias = d.row_bias;
    a.scale = d.scale;
    a.shift = d.shift;
    a.gate = d.gate;
    a.y = d.y;
    a.ldy = d.ldy;
    a.B = d.B;
    a.T_in = d.T_in;
    a.T_out = d.T_out;
    a.cin = d.cin;
    a.cin_pad = conv1d_cin_pad(d.cin);
    a.cout = d.cout;
    a.cout_pad = conv1d_cout_pad(d.cout);
    a.k = d.k;
    a.dil = d.dilation;
    a.stride = d.stride;
    a.pad = d.pad;
    a.pad_mode = d.pad_mode;
    a.pre_act = d.pre_act;
    a.post_act = d.post_act;
    a.y_f16 = d.y_dtype == MV_DT_F16;
    a.gate_seg_len = d.gate_seg_len > 0 ? d.gate_seg_len : 1;
    a.gate_nseg = (int)ceil_div(d.T_out, a.gate_seg_len);
    a.n_rows = d.B * d.T_out;
    a.n_tiles = (int)ceil_div(a.n_rows, CV_TN);
    a.co_tiles = (int)ceil_div(d.cout, CV_TC);
    const int grid = (int)round_up(a.n_tiles, 8) * a.co_tiles;
    static bool smem_set = false;
    if (!smem_set) {
        if (MV_SET_MAX_SMEM(conv1d_mfma_kernel<half_t>, CV_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM(conv1d_mfma_kernel<float>, CV_LDS_BYTES) != hipSuccess)
            return fail(MV_ERR_HIP, "conv1d: cannot reserve dynamic LDS");
        smem_set = true;
    }
    if (d.x_dtype == MV_DT_F16) {
        MV_LAUNCH(conv1d_mfma_kernel<half_t>, (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    } else {
        MV_LAUNCH(conv1d_mfma_kernel<float>, (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    }
    return check_launch("conv1d_mfma_kernel");
}

}  // namespace mv

extern "C" {

int64_t mv_conv1d_packed_elems(int32_t cout, int32_t cin, int32_t k) {
    return (int64_t)mv::conv1d_cout_pad(cout) * k * mv::conv1d_cin_pad(cin);
}

int mv_conv1d_pack_weight(const float* w, int32_t cout, int32_t cin, int32_t k, void* packed_f16, mv_stream_t stream) {
    MV_REQUIRE(w != nullptr && packed_f16 != nullptr && cout > 0 && cin > 0 && k > 0, "mv_conv1d_pack_weight: bad argument");
    const int64_t total = mv_conv1d_packed_elems(cout, cin, k);
    const int grid = (int)(mv::ceil_div(total, 256) < 2048 ? mv::ceil_div(total, 256) : 2048);
    MV_LAUNCH(mv::pack_conv_weight_kernel, (grid, 1, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), w, cout, cin, k,
              mv::conv1d_cout_pad(cout), mv::conv1d_cin_pad(cin), reinterpret_cast<half_t*>(packed_f16));
    return mv::check_launch("pack_conv_weight_kernel");
}

int mv_conv1d_forward(const MvConv1dDesc* d, mv_stream_t stream) {
    MV_REQUIRE(d != nullptr, "mv_conv1d_forward: null descriptor");
    return mv::conv1d_launch(*d, static_cast<hipStream_t>(stream));
}

}  // extern "C"
