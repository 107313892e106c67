// One CAMDenseTDNNLayer of CAM++ (mvector/models/campplus.py:71-150) as ONE launch, one workgroup per utterance:
//
//     h    = ReLU(BN2(W1 . ReLU(BN1(x))))                1x1 conv over the cin channels written so far (bottleneck 128)
//     ctx  = mean_T(h) + segmean_100(h)                   (last segment divided by its true length: avg_pool1d ceil_mode)
//     m    = sigmoid(Wb . ReLU(Wa . ctx + ba) + bb)       per 100-frame segment, 128 -> 64 -> 32, fp32
//     y    = conv_k3,dil(h) * m                           zero padding; written as 32 new channels behind the cin inputs
//
// Why: as separate launches (1x1 conv with the transform on load, segment means, two tiny FCs, k=3 conv with the gate in its
// epilogue) a layer is five kernels and ~76 us for ~12 us of memory traffic (profiles/r02k: 52 layers = 3.9 ms of the 5.8 ms
// CAM++ step); the bottleneck h makes an HBM round trip and the context needs a grid-wide boundary.  The context couples all
// frames of ONE utterance and nothing else, so an utterance per workgroup keeps h (T2 x 128 fp16 = 38 KB for 3 s) in LDS from
// the 1x1 GEMM to the k=3 conv, evaluates the context FCs in place and touches HBM for exactly: the utterance's input
// channels (once), the weights (L2-resident, shared by all workgroups) and the 32 output channels.
//
// GEMM conventions as in res2.hip: weights are the MFMA A operand (rows = output channels), activations the B operand
// (columns = time steps), so a lane ends up with 4 consecutive output channels of one time step.  8 waves: wave (cw = w & 3,
// th = w >> 2) owns output-channel tiles {2 cw, 2 cw + 1} x time tiles {5 th .. 5 th + 4} of the 1x1 GEMM; W1 streams through
// a 3-slot LDS ring by LDS-DMA, x is loaded to registers one 64-channel stage ahead, passed through BN1 + ReLU in fp32 and
// written to a double-buffered LDS tile.  (Default cache policy on these loads: the block's buffer -- 78 MB for 256 utterances --
// is re-read by every later layer and lives in the 256 MB Infinity Cache; non-temporal loads measured 1.5-2.6 TB/s, r03i.)
#include <type_traits>

#include "kernels.h"

namespace mv {

constexpr int CD_THREADS = 512;
constexpr int CD_TT = 10;                           // time tiles of 16 frames: T2 <= 160 (3.2 s of audio after the stride-2 TDNN)
constexpr int CD_ROWS = CD_TT * 16;
constexpr int CD_BN = 128;                          // bottleneck channels (bn_size * growth_rate)
constexpr int CD_G = 32;                            // growth rate
constexpr int CD_PAD = 2;                           // largest dilation of the k=3 conv
constexpr int CD_XS_BYTES = CD_ROWS * 128;          // one x stage: [160 rows][64 fp16]
constexpr int CD_WS_BYTES = CD_BN * 128;            // one W1 stage: [128 rows][64 fp16]
constexpr int CD_RING = 3;
constexpr int CD_H_BYTES = (CD_ROWS + 2 * CD_PAD) * CD_BN * 2;  // h with zero halo rows on both sides
constexpr int CD_XPF = 4;                           // x stages in flight in registers (global latency >> one stage of 20 MFMAs per wave)
static_assert(CD_XPF == 4, "the stage loop is unrolled by four and its counted waits assume this depth");
constexpr int CD_MAX_SEG = 2;                       // segments of 100 frames within 160 frames

struct CamDenseArgs {
    half_t* x;            // [B, T2, ldx]: channels [0, cin) are read, [cin, cin + 32) are written
    int64_t ldx;
    const half_t* w1;     // packed [128][1][cin_pad64]
    const float *bn1_s, *bn1_t;   // [cin]
    const float *bn2_s, *bn2_t;   // [128]
    const half_t* wl;     // packed [32][3][128]
    const float *wa, *ba, *wb, *bb;  // [64][128], [64], [32][64], [32]
    int T2, cin, cin_pad, dil, seg_len;
};

__global__ __launch_bounds__(CD_THREADS) void cam_dense_layer_kernel(CamDenseArgs a) {
    MV_DYN_SMEM(smem);
    char* xs = smem;                                           // 2 x CD_XS_BYTES (later: reduction scratch)
    char* ws = xs + 2 * CD_XS_BYTES;                           // CD_RING x CD_WS_BYTES
    char* hbuf = ws + CD_RING * CD_WS_BYTES;                   // CD_H_BYTES, row r = t + CD_PAD
    float* fsm = reinterpret_cast<float*>(hbuf + CD_H_BYTES);  // ctx [2][128] | g1 [2][64] | gate [2][32] | BN1 scale, shift [cin_pad] each
    float* ctx = fsm;
    float* g1 = ctx + CD_MAX_SEG * CD_BN;
    float* gate = g1 + CD_MAX_SEG * 64;
    float* lbn_s = gate + CD_MAX_SEG * CD_G;
    float* lbn_t = lbn_s + a.cin_pad;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x;
    const int T2 = a.T2;
    half_t* xb = a.x + (int64_t)b * T2 * a.ldx;
    const int nst = a.cin_pad / 64;

    // zero the bottleneck buffer: halo rows and the rows behind T2 must read as zero padding
    for (int i = tid; i < CD_H_BYTES / 16; i += CD_THREADS) reinterpret_cast<float4v*>(hbuf)[i] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    // BN1 parameters into LDS: read per stage next to x values that were requested four stages earlier -- a global load there
    // would wait behind every younger x prefetch (vector-memory results return in order)
    for (int i = tid; i < a.cin_pad; i += CD_THREADS) {
        lbn_s[i] = i < a.cin ? a.bn1_s[i] : 0.0f;
        lbn_t[i] = i < a.cin ? a.bn1_t[i] : 0.0f;
    }
    __syncthreads();

    // ---- parameters of the later phases are requested NOW: they are the oldest vector-memory operations of the wave, so they
    // are long complete when phases A-epilogue / B / C consume them and never sit in front of a counted wait.  (Requested where
    // they are used, each of them costs a full L2 / HBM round trip in a kernel that has ~20 us of work per launch.)
    const int cw = wave & 3, th = wave >> 2;
    float4v e_bn2s[2], e_bn2t[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        e_bn2s[mi] = *reinterpret_cast<const float4v*>(a.bn2_s + (cw * 2 + mi) * 16 + 4 * fg);
        e_bn2t[mi] = *reinterpret_cast<const float4v*>(a.bn2_t + (cw * 2 + mi) * 16 + 4 * fg);
    }
    // context FCs, cooperative: FC1 row j = tid >> 3, channels 16 * (tid & 7) .. + 15; FC2 row co = tid >> 4, inputs 4 * (tid & 15) .. + 3
    float4v e_wa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e_wa[u] = *reinterpret_cast<const float4v*>(a.wa + (tid >> 3) * CD_BN + (tid & 7) * 16 + 4 * u);
    const float4v e_wb = *reinterpret_cast<const float4v*>(a.wb + (tid >> 4) * 64 + (tid & 15) * 4);
    const float e_ba = a.ba[tid >> 3], e_bb = a.bb[tid >> 4];
    half8v e_wl[3][4];  // k=3 conv weights of this wave's channel tile: A fragments of the 12 K steps
    {
        const half_t* wrow = a.wl + (int64_t)((wave & 1) * 16 + fr) * 3 * CD_BN + 8 * fg;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e_wl[tap][kk] = *reinterpret_cast<const half8v*>(wrow + tap * CD_BN + kk * 32);
    }

    // ---- phase A: h = ReLU(BN2(W1 . ReLU(BN1(x)))) ----
    const int lrow = lane >> 3, kc = (lane & 7) ^ lrow;
    auto issue_w = [&](int s) {  // W1 stage s: 128 rows x 128 bytes = 16 transfers of 1 KiB, two per wave
        char* dst = ws + (s % CD_RING) * CD_WS_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tr = wave * 2 + u;
            const int co = tr * 8 + lrow;
            glds16(a.w1 + (int64_t)co * a.cin_pad + s * 64 + kc * 8, dst + tr * 1024);
        }
    };
    // x staging: thread = (chunk of 8 channels, row), three rows per thread and stage
    const int xchunk = tid & 7, xrow0 = tid >> 3;
    half8v xr[CD_XPF][3];
    // (every wave issues exactly three loads per stage -- rows / channels beyond the data are clamped here and zeroed in
    // store_x -- so the counted waits of the stage loop can rely on the number of operations in flight)
    auto load_x = [&](int s, half8v (&r)[3]) {
        int c = s * 64 + xchunk * 8;
        c = c < a.cin ? c : 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            int row = xrow0 + 64 * p;
            row = row < T2 ? row : T2 - 1;
            r[p] = *reinterpret_cast<const half8v*>(xb + (int64_t)row * a.ldx + c);
        }
    };
    auto store_x = [&](int s, const half8v (&r)[3]) {  // BN1 + ReLU in fp32, then into the stage tile (rows >= T2 and channels >= cin stay zero)
        const int c = s * 64 + xchunk * 8;
        const bool live = c < a.cin;
        const float4v s0 = *reinterpret_cast<const float4v*>(lbn_s + c), s1 = *reinterpret_cast<const float4v*>(lbn_s + c + 4);
        const float4v t0 = *reinterpret_cast<const float4v*>(lbn_t + c), t1 = *reinterpret_cast<const float4v*>(lbn_t + c + 4);
        char* dst = xs + (s & 1) * CD_XS_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int row = xrow0 + 64 * p;
            if (row < CD_ROWS) {
                half8v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (half_t)fmaxf((float)r[p][e] * s0[e] + t0[e], 0.0f);
                    o[4 + e] = (half_t)fmaxf((float)r[p][4 + e] * s1[e] + t1[e], 0.0f);
                }
                if (!(row < T2 && live)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)0.0f;
                }
                *reinterpret_cast<half8v*>(dst + row * 128 + ((xchunk ^ (row & 7)) << 4)) = o;
            }
        }
    };

    issue_w(0);
    if (nst > 1) issue_w(1);
    // x: CD_XPF stages in flight in registers (register set = stage % CD_XPF; the stage loop is unrolled by CD_XPF so the sets
    // are named at compile time)
#pragma unroll
    for (int u = 0; u < CD_XPF; ++u)
        if (u < nst) load_x(u, xr[u]);
    store_x(0, xr[0]);
    float4v acc[2][5];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    auto stage = [&](int s, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;  // s % CD_XPF
        // W1 stage s has landed and x stage s has been written by every thread:
        // issued after W1 stage s (which went out in stage s - 2): the x loads of stages s - 2 + CD_XPF and s - 1 + CD_XPF
        // (3 each) and W1 stage s + 1 (2 transfers), as far as those stages exist.  Stages 0 and 1: the prologue's wait for
        // x stage 0 already covered both W1 stages.
        if (s >= 2) {
            if (s + CD_XPF - 1 < nst) {
                wait_vm<8>();
            } else if (s + CD_XPF - 2 < nst) {
                wait_vm<5>();
            } else if (s + 1 < nst) {
                wait_vm<2>();
            } else {
                wait_vm<0>();
            }
        }
        lds_barrier();
        if (s + 2 < nst) issue_w(s + 2);
        if (s + CD_XPF < nst) load_x(s + CD_XPF, xr[u]);  // set u was consumed when stage s was written (end of stage s - 1)
        const char* wt = ws + (s % CD_RING) * CD_WS_BYTES;
        const char* xt = xs + (s & 1) * CD_XS_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8v af[2], bf[5];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = (cw * 2 + mi) * 16 + fr;
                af[mi] = *reinterpret_cast<const half8v*>(wt + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) {
                const int row = (th * 5 + ni) * 16 + fr;
                bf[ni] = *reinterpret_cast<const half8v*>(xt + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 5; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (s + 1 < nst) store_x(s + 1, xr[(u + 1) % CD_XPF]);  // the other x tile: its readers passed this stage's barrier
    };
    for (int s0 = 0; s0 < nst; s0 += CD_XPF) {
        stage(s0, std::integral_constant<int, 0>{});
        if (s0 + 1 < nst) stage(s0 + 1, std::integral_constant<int, 1>{});
        if (s0 + 2 < nst) stage(s0 + 2, std::integral_constant<int, 2>{});
        if (s0 + 3 < nst) stage(s0 + 3, std::integral_constant<int, 3>{});
    }
    // epilogue A: BN2 + ReLU -> hbuf (fp16, swizzled 16-byte chunks: chunk ^= row & 15), frames >= T2 stay zero
    auto h_off = [&](int row, int chunk) { return row * (CD_BN * 2) + ((chunk ^ (row & 15)) << 4); };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int co = (cw * 2 + mi) * 16 + 4 * fg;
        const float4v sc = e_bn2s[mi], sh = e_bn2t[mi];
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) {
            const int t = (th * 5 + ni) * 16 + fr;
            half4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (half_t)fmed3(fmaxf(acc[mi][ni][r] * sc[r] + sh[r], 0.0f), 0.0f, 65504.0f);
            if (t < T2) *reinterpret_cast<half4v*>(hbuf + h_off(t + CD_PAD, co >> 3) + (co & 7) * 2) = hv;
        }
    }
    __syncthreads();

    // ---- phase B: context gate per 100-frame segment ----
    const int nseg = (T2 + a.seg_len - 1) / a.seg_len;  // <= CD_MAX_SEG (checked by the launcher)
    {
        // partial sums: thread = (8-channel chunk, 32 row phases); scratch [32][2][128] floats in the (now idle) x tiles
        float* part = reinterpret_cast<float*>(xs);
        const int cg = tid & 15, rp = tid >> 4;
        float sum[CD_MAX_SEG][8];
#pragma unroll
        for (int sg = 0; sg < CD_MAX_SEG; ++sg)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[sg][e] = 0.0f;
        for (int t = rp; t < T2; t += 32) {
            const half8v v = *reinterpret_cast<const half8v*>(hbuf + h_off(t + CD_PAD, cg));
            if (t < a.seg_len) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[0][e] += (float)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[1][e] += (float)v[e];
            }
        }
#pragma unroll
        for (int sg = 0; sg < CD_MAX_SEG; ++sg)
#pragma unroll
            for (int e = 0; e < 8; ++e) part[(rp * CD_MAX_SEG + sg) * CD_BN + cg * 8 + e] = sum[sg][e];
        __syncthreads();
        if (tid < CD_MAX_SEG * CD_BN) {
            const int sg = tid / CD_BN, c = tid - sg * CD_BN;
            float v = 0.0f, other = 0.0f;
            for (int p = 0; p < 32; ++p) {
                v += part[(p * CD_MAX_SEG + sg) * CD_BN + c];
                other += part[(p * CD_MAX_SEG + (1 - sg)) * CD_BN + c];
            }
            const int t0 = sg * a.seg_len;
            const int len = (t0 + a.seg_len < T2 ? t0 + a.seg_len : T2) - t0;
            ctx[sg * CD_BN + c] = len > 0 ? (v + other) / (float)T2 + v / (float)len : 0.0f;
        }
        __syncthreads();
        {   // g1 = ReLU(Wa ctx + ba): 8 threads per output row, both segments, partial dot products summed over 8 lanes (DPP)
            const int j = tid >> 3, part8 = tid & 7;
#pragma unroll
            for (int sg = 0; sg < CD_MAX_SEG; ++sg) {
                const float* cx = ctx + sg * CD_BN + part8 * 16;
                float v = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4v c4 = *reinterpret_cast<const float4v*>(cx + 4 * u);
                    v = fmaf(e_wa[u][0], c4[0], v);
                    v = fmaf(e_wa[u][1], c4[1], v);
                    v = fmaf(e_wa[u][2], c4[2], v);
                    v = fmaf(e_wa[u][3], c4[3], v);
                }
                v += dpp_mov<DPP_QUAD_XOR1>(0.0f, v);
                v += dpp_mov<DPP_QUAD_XOR2>(0.0f, v);
                v += dpp_mov<DPP_ROW_HALF_MIRROR>(0.0f, v);  // lanes l and 7 - l of each 8: the two quads hold equal sums after the quad steps
                if (part8 == 0) g1[sg * 64 + j] = fmaxf(v + e_ba, 0.0f);
            }
        }
        __syncthreads();
        {   // gate = sigmoid(Wb g1 + bb): 16 threads per output row (one DPP row), both segments
            const int co = tid >> 4, part16 = tid & 15;
#pragma unroll
            for (int sg = 0; sg < CD_MAX_SEG; ++sg) {
                const float4v g4 = *reinterpret_cast<const float4v*>(g1 + sg * 64 + part16 * 4);
                float v = e_wb[0] * g4[0];
                v = fmaf(e_wb[1], g4[1], v);
                v = fmaf(e_wb[2], g4[2], v);
                v = fmaf(e_wb[3], g4[3], v);
                v = row16_sum(v);
                if (part16 == 0) gate[sg * CD_G + co] = 1.0f / (1.0f + expf(-(v + e_bb)));
            }
        }
        __syncthreads();
    }

    // ---- phase C: y = conv_k3(h) * gate -> channels [cin, cin + 32) of x ----
    {
        const int ct = wave & 1, tg = wave >> 1;  // channel tile, time tiles tg, tg + 4, tg + 8
        float4v yc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) yc[j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half8v af = e_wl[tap][kk];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int tile = tg + 4 * j;
                    if (tile < CD_TT) {  // uniform per wave
                        const int row = tile * 16 + fr + (tap - 1) * a.dil + CD_PAD;
                        const half8v bfr = *reinterpret_cast<const half8v*>(hbuf + h_off(row, kk * 4 + fg));
                        yc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr, yc[j], 0, 0, 0);
                    }
                }
            }
        }
        const int co = ct * 16 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tile = tg + 4 * j;
            const int t = tile * 16 + fr;
            if (tile < CD_TT && t < T2) {
                const float* gt = gate + (t / a.seg_len) * CD_G + co;
                half4v hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (half_t)fmed3(yc[j][r] * gt[r], -65504.0f, 65504.0f);
                *reinterpret_cast<half4v*>(xb + (int64_t)t * a.ldx + a.cin + co) = hv;
            }
        }
    }
}

constexpr int CD_MAX_CIN = 2048;  // BN1 parameters of the layer live in LDS
constexpr size_t CD_LDS_BYTES = 2 * CD_XS_BYTES + CD_RING * CD_WS_BYTES + CD_H_BYTES + (CD_MAX_SEG * (CD_BN + 64 + CD_G) + 2 * CD_MAX_CIN) * sizeof(float);

bool cam_dense_layer_supported(int T2, int cin, int bottleneck, int growth, int dil, int seg_len) {
    return bottleneck == CD_BN && growth == CD_G && T2 >= 1 && T2 <= CD_ROWS && cin % 32 == 0 && cin >= 32 && cin <= CD_MAX_CIN - 64 && dil >= 1 &&
           dil <= CD_PAD &&
           seg_len > 0 && (T2 + seg_len - 1) / seg_len <= CD_MAX_SEG;
}

int cam_dense_layer_launch(half_t* x, int64_t ldx, int B, int T2, int cin, const half_t* w1, const float* bn1_s, const float* bn1_t,
                           const float* bn2_s, const float* bn2_t, const half_t* wl, const float* wa, const float* ba, const float* wb,
                           const float* bb, int dil, int seg_len, hipStream_t stream) {
    MV_REQUIRE(cam_dense_layer_supported(T2, cin, CD_BN, CD_G, dil, seg_len), "cam_dense_layer: unsupported geometry");
    MV_REQUIRE(ldx >= cin + CD_G && (ldx % 8) == 0, "cam_dense_layer: the row must hold the inputs and 32 new channels (16-byte aligned chunks)");
    static bool smem_set = false;
    if (!smem_set) {
        if (MV_SET_MAX_SMEM(cam_dense_layer_kernel, CD_LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "cam_dense_layer: cannot reserve LDS");
        smem_set = true;
    }
    CamDenseArgs a;
    a.x = x; a.ldx = ldx; a.w1 = w1; a.bn1_s = bn1_s; a.bn1_t = bn1_t; a.bn2_s = bn2_s; a.bn2_t = bn2_t; a.wl = wl;
    a.wa = wa; a.ba = ba; a.wb = wb; a.bb = bb;
    a.T2 = T2; a.cin = cin; a.cin_pad = conv1d_cin_pad(cin); a.dil = dil; a.seg_len = seg_len;
    MV_LAUNCH(cam_dense_layer_kernel, ((unsigned)B, 1, 1), (CD_THREADS, 1, 1), CD_LDS_BYTES, stream, a);
    return check_launch("cam_dense_layer_kernel");
}

}  // namespace mv
