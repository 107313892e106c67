// Small exact-fp32 dense layers on the f32 MFMA (v_mfma_f32_16x16x4_f32) and cosine scoring.
//
// Used for the parts of the path whose inputs are per-utterance vectors, where precision matters more
// than rate and the work is tiny:
//   * SE excitation FCs (mvector/models/ecapa_tdnn.py:81-82), CAM context FCs (campplus.py:97-98),
//   * the hoisted ASP context term W[:, C:3C] . [mean; std] (pooling.py:110-117),
//   * asp_bn folded into fc (ecapa_tdnn.py:278-281), TDNN linear + bn6 (tdnn.py:66-67),
//     CAM++ dense layer (campplus.py:200-216),
//   * cosine scoring (predict.py:169-183, 275-279; trainer.py:454-461).
//
// One workgroup (4 waves) computes one 16 (rows of x) x 16 (outputs) tile; the four waves split K and
// reduce through LDS.  Each lane loads float4 of x and of w along K (16 rows x 64 B per instruction) and
// feeds the four components to four MFMAs, which is a consistent permutation of the K index for both
// operands.
//
// Measured (r10f, tools/bench_linear.py): an XCD-aware workgroup -> tile map for the K = 6144 layers (every XCD owning two row tiles of x, so
// that x passes through one L2 instead of eight) changes nothing -- 24.3 / 24.7 us warm and 28 us behind a cache flush in both forms, and
// [2048, 6144] x [192, 6144] takes 8 x the time of 256 rows: the launch is bound by the workgroup's own chain (three dependent load trips + 96
// fp32 MFMAs per wave, four waves per SIMD), not by operand delivery through the fabric.
#include "kernels.h"

namespace mv {

__device__ __forceinline__ float lin_act(float v, int act) {
    if (act == MV_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MV_ACT_TANH) return tanhf(v);
    if (act == MV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

__device__ __forceinline__ float4v load4_guard(const float* row, int k, int K, bool vec_ok) {
    if (vec_ok && k + 3 < K) return *reinterpret_cast<const float4v*>(row + k);
    float4v v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? row[k + e] : 0.0f;
    return v;
}

// y[b, o] = act( sum_k x[b,k] w[o,k] + bias[o] );  with `cosine` set the dot product is divided by the two
// row norms |x_b| |w_o| (accumulated in the same K loop), i.e. sklearn's cosine_similarity.
template <int NW, int U = 4>
__global__ __launch_bounds__(64 * NW) void linear_f32_kernel(const float* x, int64_t ldx, const float* w, int64_t ldw,
                                                             const float* bias, int act, float* y, int64_t ldy, int B,
                                                             int K, int O, int cosine) {
    __shared__ float red[NW][16][17];
    __shared__ float nrm[2][NW][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    const int o0 = blockIdx.x * 16;
    const int b0 = blockIdx.y * 16;
    const int br = (b0 + i < B) ? b0 + i : B - 1;  // clamp: duplicates are masked at the store
    const int oc = (o0 + i < O) ? o0 + i : O - 1;
    const float* xrow = x + (int64_t)br * ldx;
    const float* wrow = w + (int64_t)oc * ldw;
    const bool xvec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const bool wvec = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    float4v acc = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    float sqx = 0.0f, sqw = 0.0f;
    // wave `wave` takes K blocks of 16 with index == wave (mod NW).  Main loop: U unguarded blocks per trip, their 2 U
    // 16-byte loads in flight together (the chain of dependent load -> MFMA trips is what these small layers wait for: U = 8 for the
    // K = 6144 layers of the ASP head, three trips instead of six); the guarded loop takes the remainder (and everything when the rows
    // are not 16-byte aligned).
    const int kfull = (xvec && wvec) ? (K & ~15) : 0;
    int k0 = MV_UNIFORM(wave) * 16;   // (a scalar: the loops below branch on it)
    for (; k0 + (U - 1) * 16 * NW < kfull; k0 += U * 16 * NW) {
        float4v xa[U], wb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xa[u] = *reinterpret_cast<const float4v*>(xrow + k0 + u * 16 * NW + 4 * g);
            wb[u] = *reinterpret_cast<const float4v*>(wrow + k0 + u * 16 * NW + 4 * g);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][e], wb[u][e], acc, 0, 0, 0);
            if (cosine) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sqx += xa[u][e] * xa[u][e];
                    sqw += wb[u][e] * wb[u][e];
                }
            }
        }
    }
    for (; k0 < K; k0 += 16 * NW) {
        const int k = k0 + 4 * g;
        float4v xa, wb;
        // (the choice between the 16-byte load and the guarded one is made per WAVE: decided per lane, both forms run one after the other under
        //  complementary masks into the same registers, and the second waits -- s_waitcnt vmcnt(0) -- for the first: two round trips per block)
        if (xvec && wvec && k0 + 16 <= K) {
            xa = *MV_GLOBAL_PTR(float4v, xrow + k);
            wb = *MV_GLOBAL_PTR(float4v, wrow + k);
        } else {
            xa = load4_guard(xrow, k, K, false);
            wb = load4_guard(wrow, k, K, false);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], wb[e], acc, 0, 0, 0);
        if (cosine) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sqx += xa[e] * xa[e];
                sqw += wb[e] * wb[e];
            }
        }
    }
    if (cosine) {
        sqx += __shfl_xor(sqx, 16);
        sqx += __shfl_xor(sqx, 32);
        sqw += __shfl_xor(sqw, 16);
        sqw += __shfl_xor(sqw, 32);
        if (g == 0) {
            nrm[0][wave][i] = sqx;
            nrm[1][wave][i] = sqw;
        }
    }
    // lane holds D[row = 4g + r][col = i]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * g + r][i] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int r = tid >> 4, c = tid & 15;
        const int b = b0 + r, o = o0 + c;
        if (b < B && o < O) {
            float v = 0.0f, sx = 0.0f, sw = 0.0f;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                v += red[q][r][c];
                sx += nrm[0][q][r];
                sw += nrm[1][q][c];
            }
            if (cosine) {
                const float nx = sqrtf(sx);
                const float nw = sqrtf(sw);
                v = v / (fmaxf(nx, 1.17549435e-38f) * fmaxf(nw, 1.17549435e-38f));
            }
            if (bias != nullptr) v += bias[o];
            y[(int64_t)b * ldy + o] = lin_act(v, act);
        }
    }
}

// ---- long reductions with many rows (round 5): split K over workgroups ------------------------------------------------------------------
// [256, 6144] x [192, 6144] on the kernel above is 192 workgroups of 16 x 16 outputs, each streaming its 16 + 16 rows of 6144 floats: 151 MB
// through L2 for 0.6 GFLOP, three dependent load trips per wave -- 31 us per launch, twice per EcapaTdnn-1024 step (ASP context bias, final fc).
// Here a workgroup of four waves owns 32 x 32 outputs of ONE K slice of 384: every wave loads two row fragments and two weight fragments per 16
// K values for SIXTEEN MFMAs (2 x 2 register blocking: half the operand bytes per output), all of its loads in one trip; the slices' partial
// sums go to a caller workspace [slices][B][O] and a second launch adds them in slice order (+ bias, activation) -- a fixed order, so a row's
// bits depend on nothing but the row.  (Atomics into y would need no second launch and give a different sum every run.)
constexpr int LSK_SLICE = 384;   // K values per slice: 4 waves x 6 blocks of 16
constexpr int LSK_U = LSK_SLICE / 64;

__global__ __launch_bounds__(256) void linear_f32_splitk_kernel(const float* x, int64_t ldx, const float* w, int64_t ldw, float* part, int B, int K, int O) {
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int kb = blockIdx.z * LSK_SLICE;
    const int ke = kb + LSK_SLICE < K ? kb + LSK_SLICE : K;
    const float* xrow[2];
    const float* wrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int br = b0 + 16 * m + i < B ? b0 + 16 * m + i : B - 1;   // clamp: duplicates are masked at the store
        const int oc = o0 + 16 * m + i < O ? o0 + 16 * m + i : O - 1;
        xrow[m] = x + (int64_t)br * ldx;
        wrow[m] = w + (int64_t)oc * ldw;
    }
    const bool xvec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const bool wvec = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    float4v acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    float4v xa[LSK_U][2], wb[LSK_U][2];
    // wave `wave` takes the blocks of 16 with index == wave (mod 4) of the slice: all loads of the wave in one trip.  Two COPIES of the load loop, chosen
    // once per workgroup: with the guarded form anywhere near them (chosen per lane by `k + 3 < K`, or per wave inside the loop) the 16-byte loads ran
    // behind an s_waitcnt vmcnt(0) each -- both forms write the same registers, under complementary masks or on paths the wait-count insertion
    // cannot tell apart -- and the "one trip" was twelve round trips one after the other (r14o, tools/isa_audit.py).
    if (xvec && wvec && kb + LSK_SLICE <= K) {   // uniform: every slice but a ragged last one
#pragma unroll
        for (int u = 0; u < LSK_U; ++u) {
            const int k = kb + (u * 4 + wave) * 16 + 4 * g;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xa[u][m] = *MV_GLOBAL_PTR(float4v, xrow[m] + k);
                wb[u][m] = *MV_GLOBAL_PTR(float4v, wrow[m] + k);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < LSK_U; ++u) {
            const int k = kb + (u * 4 + wave) * 16 + 4 * g;
            const int kc = k < ke ? k : kb;   // blocks behind the slice's end: any valid (aligned) address, the values are zeroed below
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xa[u][m] = load4_guard(xrow[m], kc, K, xvec);
                wb[u][m] = load4_guard(wrow[m], kc, K, wvec);
                if (k >= ke) xa[u][m] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    }
#pragma unroll
    for (int u = 0; u < LSK_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][m][e], wb[u][n][e], acc[m][n], 0, 0, 0);
    // lane holds D[row = 16 m + 4 g + r][col = 16 n + i]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][16 * m + 4 * g + r][16 * n + i] = acc[m][n][r];
    __syncthreads();
    float* pz = part + (int64_t)blockIdx.z * B * O;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j, r = idx >> 5, c = idx & 31;
        const int b = b0 + r, o = o0 + c;
        if (b < B && o < O) pz[(int64_t)b * O + o] = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
    }
}

__global__ __launch_bounds__(256) void linear_f32_splitk_finish_kernel(const float* part, const float* bias, int act, float* y, int64_t ldy, int B, int O, int slices) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * O) return;
    const int b = (int)(idx / O), o = (int)(idx - (int64_t)b * O);
    float v = 0.0f;
    for (int s = 0; s < slices; ++s) v += part[(int64_t)s * B * O + idx];   // slice order
    if (bias != nullptr) v += bias[o];
    y[(int64_t)b * ldy + o] = lin_act(v, act);
}

// inv[r] = 1 / max(|x_r|, tiny); optionally normalise in place
__global__ __launch_bounds__(256) void row_inv_norm_kernel(float* x, int n, int dim, float* inv, int normalize) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float* p = x + (int64_t)row * dim;
    float s = 0.0f;
    for (int k = lane; k < dim; k += 64) s += p[k] * p[k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float nrm = sqrtf(s);
    const float iv = 1.0f / fmaxf(nrm, 1.17549435e-38f);
    if (inv != nullptr && lane == 0) inv[row] = iv;
    if (normalize)
        for (int k = lane; k < dim; k += 64) p[k] = p[k] / nrm;  // features / np.linalg.norm (predict.py:165-166)
}

// The split-K form is chosen by the LAYER (K, O), never by the number of rows: a row's bits must not depend on the batch it sits in (one utterance
// and 256 take the same form; r14i: with a row threshold the B = 1 / 8 rows of EcapaTdnn-1024 differed from their B = 256 bits).  Workspace:
// [slices][B][O] floats.
size_t linear_f32_splitk_floats(int B, int K, int O) {
    if (K < 2048 || O < 16 || B < 1 || B > 65535 * 32) return 0;
    return (size_t)ceil_div(K, LSK_SLICE) * B * O;
}

int linear_f32_launch(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int act, float* y,
                      int64_t ldy, int B, int K, int O, int cosine, hipStream_t stream, float* splitk_ws, size_t splitk_ws_floats) {
    MV_REQUIRE(x != nullptr && w != nullptr && y != nullptr, "linear_f32: null tensor");
    MV_REQUIRE(B > 0 && K > 0 && O > 0 && ldx >= K && ldw >= K && ldy >= O, "linear_f32: bad geometry");
    {
        const size_t need = cosine ? 0 : linear_f32_splitk_floats(B, K, O);
        if (need > 0 && splitk_ws != nullptr && splitk_ws_floats >= need) {
            const int slices = (int)ceil_div(K, LSK_SLICE);
            MV_LAUNCH(linear_f32_splitk_kernel, ((unsigned)ceil_div(O, 32), (unsigned)ceil_div(B, 32), (unsigned)slices), (256, 1, 1), 0, stream, x, ldx, w, ldw,
                      splitk_ws, B, K, O);
            MV_LAUNCH(linear_f32_splitk_finish_kernel, ((unsigned)ceil_div((int64_t)B * O, 256), 1, 1), (256, 1, 1), 0, stream, splitk_ws, bias, act, y, ldy, B, O, slices);
            return check_launch("linear_f32_splitk_kernel");
        }
    }
    // grid.y holds at most 65535 row tiles: longer inputs (evaluation score matrices with > 1 M trial rows) go in row chunks
    constexpr int kMaxRows = 65535 * 16;
    if (B > kMaxRows) {
        for (int b0 = 0; b0 < B; b0 += kMaxRows) {
            const int rows = B - b0 < kMaxRows ? B - b0 : kMaxRows;
            const int rc = linear_f32_launch(x + (int64_t)b0 * ldx, ldx, w, ldw, bias, act, y + (int64_t)b0 * ldy, ldy, rows, K, O, cosine, stream);
            if (rc != MV_OK) return rc;
        }
        return MV_OK;
    }
    if (K >= 4096) {
        MV_LAUNCH((linear_f32_kernel<16, 8>), ((unsigned)ceil_div(O, 16), (unsigned)ceil_div(B, 16), 1), (1024, 1, 1), 0, stream, x, ldx,
                  w, ldw, bias, act, y, ldy, B, K, O, cosine);
    } else if (K >= 2048) {  // long reductions: 16 waves split K so the dependent load chain per wave stays short
        MV_LAUNCH(linear_f32_kernel<16>, ((unsigned)ceil_div(O, 16), (unsigned)ceil_div(B, 16), 1), (1024, 1, 1), 0, stream, x, ldx,
                  w, ldw, bias, act, y, ldy, B, K, O, cosine);
    } else {
        MV_LAUNCH(linear_f32_kernel<4>, ((unsigned)ceil_div(O, 16), (unsigned)ceil_div(B, 16), 1), (256, 1, 1), 0, stream, x, ldx, w,
                  ldw, bias, act, y, ldy, B, K, O, cosine);
    }
    return check_launch("linear_f32_kernel");
}

}  // namespace mv

extern "C" {

int mv_linear_f32(const float* x, int64_t ldx, const float* w, const float* bias, int32_t act, float* y, int64_t ldy,
                  int32_t B, int32_t K, int32_t O, mv_stream_t stream) {
    return mv::linear_f32_launch(x, ldx, w, K, bias, act, y, ldy, B, K, O, 0, static_cast<hipStream_t>(stream));
}

size_t mv_linear_f32_workspace_floats(int32_t B, int32_t K, int32_t O) { return mv::linear_f32_splitk_floats(B, K, O); }

int mv_linear_f32_ws(const float* x, int64_t ldx, const float* w, const float* bias, int32_t act, float* y, int64_t ldy, int32_t B, int32_t K, int32_t O,
                     float* workspace, size_t workspace_floats, mv_stream_t stream) {
    return mv::linear_f32_launch(x, ldx, w, K, bias, act, y, ldy, B, K, O, 0, static_cast<hipStream_t>(stream), workspace, workspace_floats);
}

int mv_cosine_f32(const float* a, int32_t n, const float* b, int32_t m, int32_t dim, float* scores, mv_stream_t stream) {
    if (n == 0 || m == 0) return MV_OK;
    return mv::linear_f32_launch(a, dim, b, dim, nullptr, MV_ACT_NONE, scores, m, n, dim, m, 1,
                                 static_cast<hipStream_t>(stream));
}

int mv_l2_normalize_f32(float* x, int32_t n, int32_t dim, mv_stream_t stream) {
    MV_REQUIRE(x != nullptr && n > 0 && dim > 0, "mv_l2_normalize_f32: bad argument");
    MV_LAUNCH(mv::row_inv_norm_kernel, ((unsigned)mv::ceil_div(n, 4), 1, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), x, n,
              dim, static_cast<float*>(nullptr), 1);
    return mv::check_launch("row_inv_norm_kernel");
}

}  // extern "C"
