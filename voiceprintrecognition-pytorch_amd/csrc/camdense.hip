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
// th = w >> 2) owns output-channel tiles {2 cw, 2 cw + 1} x time tiles {5 th .. 5 th + 4} of the 1x1 GEMM.  Both operand streams
// are LDS-DMA rings (global_load_lds, issued from inline assembly -- glds16_untracked): W1 in 3 slots, the raw x stage
// ([160 rows][64 channels] fp16) in 4 slots; a landed x stage is passed through BN1 + ReLU IN PLACE (LDS -> fp32 -> LDS) one
// stage before the MFMAs read it, so two x stages and two weight stages are in flight under every stage's MFMAs with one counted
// s_waitcnt and one barrier per stage.  (Round 2, first form: x went through registers, "four stages in flight" -- but the compiler,
// seeing register loads next to builtin LDS-DMA on the same counter, put s_waitcnt vmcnt(0) in front of their first use and of the
// fragment reads, draining everything in every stage.  Measured after the change, r04l: the same 25 us per layer on average, 16 us
// for the 2-stage layers, 36 us for the 16-stage ones = ~14 us of fixed cost -- launch, parameter and first-stage latency, the
// serial context phase, store drain -- plus ~1.35 us per stage, which is what the in-place transform (~120 VALU per wave) and 20
// MFMAs cost two waves per SIMD.  The stream is no longer the limit; chaining the layers of a block in one launch is what
// would remove the fixed part.  Measured, r05t: the layers of a block in ONE launch -- layer loop inside the kernel, s_waitcnt vmcnt(0) +
// barrier between layers, arguments as an array in the kernel argument segment -- is correct and 3 % SLOWER end to end (101.3 k -> 98.4 k
// utt/s): between two launches the store drain of one workgroup overlaps with the start of the next kernel's workgroup on the same CU,
// inside one workgroup it is a wait.)  Round 3 timeline (profiles/r07b_cam_dense_inkernel_timeline.log, mean over the 52 layers,
// 24.4 us): BN1 tables 1.0 us, parameters + first x stage 5.6 us, stage loop 11.7 us (1.3 us per stage), h 1.2, context 2.6,
// k = 3 conv + stores 2.3.  Moving the parameter requests behind the last stage's transfers (r07c) only moves the wait behind the
// loop (4 us: every workgroup of the launch asks the same L2 lines for the same 90 KB at the same moment) -- the fixed part of a
// layer is memory latency of a chip-wide synchronised start, which only a prefetch across the layer boundary (persistent
// per-block kernel that requests layer l + 1's parameters and first stages under layer l's context phase) removes.
// Default cache policy on these loads: the block's buffer --
// 78 MB for 256 utterances -- is re-read by every later layer and lives in the 256 MB Infinity Cache; non-temporal loads measured
// 1.5-2.6 TB/s, r03i.
#include <type_traits>

#include "kernels.h"

namespace mv {

constexpr int CD_THREADS = 512;
constexpr int CD_TT = 10;                           // time tiles of 16 frames: T2 <= 160 (3.2 s of audio after the stride-2 TDNN)
constexpr int CD_ROWS = CD_TT * 16;
constexpr int CD_BN = 128;                          // bottleneck channels (bn_size * growth_rate)
constexpr int CD_G = 32;                            // growth rate
constexpr int CD_PAD = 2;                           // largest dilation of the k=3 conv
constexpr int CD_XS_BYTES = CD_ROWS * 128;          // one x stage: [160 rows][64 fp16] = 20 transfers of 1 KiB
constexpr int CD_WS_BYTES = CD_BN * 128;            // one W1 stage: [128 rows][64 fp16] = 16 transfers
constexpr int CD_RING = 3;                          // W1 stages: s (read), s + 1, s + 2
constexpr int CD_XRING = 4;                         // x stages: s (read), s + 1 (BN1 + ReLU in place), s + 2, s + 3 (in flight)
constexpr int CD_H_BYTES = (CD_ROWS + 2 * CD_PAD) * CD_BN * 2;  // h with zero halo rows on both sides; reuses the x ring after phase A
constexpr int CD_PART_BYTES = 32 * 2 * CD_BN * 4;   // phase B partial sums, behind h in the x ring
static_assert(CD_H_BYTES + CD_PART_BYTES <= CD_XRING * CD_XS_BYTES, "h and the phase B scratch live in the idle x ring");
constexpr int CD_MAX_SEG = 2;                       // segments of 100 frames within 160 frames
constexpr int CD_MAX_CIN = 2048;                    // BN1 parameters of the layer live in LDS

// 256 zero bytes: source of the padding transfers
__device__ __attribute__((aligned(256))) const unsigned char g_cd_zero_page[256] = {0};

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
    char* xs = smem;                                           // CD_XRING x CD_XS_BYTES; after phase A: h, then the phase B scratch
    char* ws = xs + CD_XRING * CD_XS_BYTES;                    // CD_RING x CD_WS_BYTES
    char* hbuf = xs;                                           // CD_H_BYTES, row r = t + CD_PAD
    float* fsm = reinterpret_cast<float*>(ws + CD_RING * CD_WS_BYTES);  // ctx [2][128] | g1 [2][64] | gate [2][32] | BN1 scale, shift [cin_pad] each
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

    // BN1 parameters into LDS: the in-place transform of every stage reads them next to transfers in flight
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
    float4v e_wb = *reinterpret_cast<const float4v*>(a.wb + (tid >> 4) * 64 + (tid & 15) * 4);
    float e_ba = a.ba[tid >> 3], e_bb = a.bb[tid >> 4];
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
    const unsigned xs_addr = lds_addr(xs), ws_addr = lds_addr(ws);
    const unsigned dump_addr = lds_addr(reinterpret_cast<char*>(gate + CD_MAX_SEG * CD_G + 2 * CD_MAX_CIN));  // 1 KiB behind the BN1 tables: where the padding transfers land
    const half_t* zero = reinterpret_cast<const half_t*>(g_cd_zero_page);
    const int wave_u = MV_UNIFORM(wave);
    // Every wave issues exactly 3 x transfers and 2 W1 transfers per stage -- stages beyond the last one and the x transfers 20..23
    // read a constant page into the dump KiB -- so the waits below can be counted.
    auto issue_x = [&](int s) {  // x stage s: 160 rows x 128 bytes; transfer tr covers rows 8 tr .. 8 tr + 7 (rows >= T2 re-read row T2 - 1)
        const bool real = s < nst;
        const unsigned dst = xs_addr + (unsigned)((s & (CD_XRING - 1)) * CD_XS_BYTES);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tr = wave_u + 8 * u;
            int row = tr * 8 + lrow;
            row = row < T2 ? row : T2 - 1;
            const bool live = real && tr < CD_ROWS / 8;  // uniform
            glds16_untracked(live ? xb + (int64_t)row * a.ldx + s * 64 + kc * 8 : zero, live ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    auto issue_w = [&](int s) {  // W1 stage s: 128 rows x 128 bytes = 16 transfers, two per wave
        const bool real = s < nst;
        const unsigned dst = ws_addr + (unsigned)((s % CD_RING) * CD_WS_BYTES);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tr = wave_u * 2 + u;
            const int co = tr * 8 + lrow;
            glds16_untracked(real ? a.w1 + (int64_t)co * a.cin_pad + s * 64 + kc * 8 : zero, real ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    // BN1 + ReLU in fp32, in place: thread = (16-byte chunk, row), three rows per thread and stage; rows >= T2 and channels >= cin
    // become zero (whatever the transfer brought: clamped rows, the not yet written output channels of this very layer)
    const int xchunk = tid & 7, xrow0 = tid >> 3;
    auto transform = [&](int s) {
        const int c = s * 64 + xchunk * 8;
        const bool live = c < a.cin;
        const float4v s0 = *reinterpret_cast<const float4v*>(lbn_s + c), s1 = *reinterpret_cast<const float4v*>(lbn_s + c + 4);
        const float4v t0 = *reinterpret_cast<const float4v*>(lbn_t + c), t1 = *reinterpret_cast<const float4v*>(lbn_t + c + 4);
        char* tile = xs + (s & (CD_XRING - 1)) * CD_XS_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int row = xrow0 + 64 * p;
            if (row < CD_ROWS) {
                half8v* cell = reinterpret_cast<half8v*>(tile + row * 128 + ((xchunk ^ (row & 7)) << 4));
                const half8v r = *cell;
                half8v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (half_t)fmaxf((float)r[e] * s0[e] + t0[e], 0.0f);
                    o[4 + e] = (half_t)fmaxf((float)r[4 + e] * s1[e] + t1[e], 0.0f);
                }
                if (!(row < T2 && live)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)0.0f;
                }
                *cell = o;
            }
        }
    };

    // request order = stages -3, -2, -1 of the rule "stage s requests x(s+3), then W1(s+2)": x0 | x1 W0 | x2 W1
    issue_x(0);
    issue_x(1);
    issue_w(0);
    issue_x(2);
    issue_w(1);
    // The parameter loads above are the only vector loads the compiler tracks: touch what they deliver, so that its wait for them
    // sits here and not inside the stage loop (where a conservative vmcnt(0) would drain the rings)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        MV_OPAQUE(e_bn2s[mi]);
        MV_OPAQUE(e_bn2t[mi]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) MV_OPAQUE(e_wa[u]);
    MV_OPAQUE(e_wb);
    MV_OPAQUE(e_ba);
    MV_OPAQUE(e_bb);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) MV_OPAQUE(e_wl[tap][kk]);
    wait_vm<10>();  // x0 has landed (younger: x1 W0 x2 W1 = 3 + 2 + 3 + 2)
    lds_barrier();
    transform(0);
    float4v acc[2][5];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
        // W1(s) and x(s+1) have landed: requested in stage s - 2, younger are x(s+2) and W1(s+1) = 3 + 2 transfers
        wait_vm<5>();
        lds_barrier();  // ... in every wave; x(s) is transformed; every wave is done with x(s-1) and W1(s-1), whose slots are requested now
        issue_x(s + 3);
        issue_w(s + 2);
        const char* wt = ws + (s % CD_RING) * CD_WS_BYTES;
        const char* xt = xs + (s & (CD_XRING - 1)) * CD_XS_BYTES;
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
        if (s + 1 < nst) transform(s + 1);  // uniform
    }
    wait_vm<0>();   // only padding transfers are left: nothing may still be landing when the ring is reused
    lds_barrier();  // every wave is done with the x ring: it becomes h (rows [CD_PAD, CD_PAD + T2)) and the phase B scratch
    // halo rows and the rows behind T2 of h read as zero padding
    {
        constexpr int ROWB = CD_BN * 2;
        const int tail0 = (T2 + CD_PAD) * ROWB, total = CD_H_BYTES;
        for (int i = tid * 16; i < CD_PAD * ROWB; i += CD_THREADS * 16) *reinterpret_cast<float4v*>(hbuf + i) = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = tail0 + tid * 16; i < total; i += CD_THREADS * 16) *reinterpret_cast<float4v*>(hbuf + i) = float4v{0.0f, 0.0f, 0.0f, 0.0f};
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
        float* part = reinterpret_cast<float*>(xs + CD_H_BYTES);
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
        // channel tile ct, time tiles tg, tg + 4, tg + 8; waves of time groups 2 / 3 own two tiles: their third accumulator repeats tile 9 and is never
        // stored (twelve spare MFMAs instead of a uniform branch -- and a drained LDS queue -- per K step, camblock.hip)
        const int ct = wave & 1, tg = wave >> 1;
        const bool third = tg + 8 < CD_TT;
        int trow[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) trow[j] = (j < 2 || third ? tg + 4 * j : CD_TT - 1) * 16 + fr;
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
                    const int row = trow[j] + (tap - 1) * a.dil + CD_PAD;
                    const half8v bfr = *reinterpret_cast<const half8v*>(hbuf + h_off(row, kk * 4 + fg));
                    yc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr, yc[j], 0, 0, 0);
                }
            }
        }
        const int co = ct * 16 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int t = trow[j];
            if ((j < 2 || third) && t < T2) {
                const float* gt = gate + (t / a.seg_len) * CD_G + co;
                half4v hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (half_t)fmed3(yc[j][r] * gt[r], -65504.0f, 65504.0f);
                *reinterpret_cast<half4v*>(xb + (int64_t)t * a.ldx + a.cin + co) = hv;
            }
        }
    }
}

constexpr size_t CD_LDS_BYTES = CD_XRING * CD_XS_BYTES + CD_RING * CD_WS_BYTES + (CD_MAX_SEG * (CD_BN + 64 + CD_G) + 2 * CD_MAX_CIN) * sizeof(float) + 1024;

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
    static DeviceOnce smem_set;   // (per device: the attribute belongs to the current device's code object)
    int smem_set_slot;
    if (device_once_pending(smem_set, &smem_set_slot)) {
        if (MV_SET_MAX_SMEM(cam_dense_layer_kernel, CD_LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "cam_dense_layer: cannot reserve LDS");
        device_once_done(smem_set, smem_set_slot);
    }
    CamDenseArgs a;
    a.x = x; a.ldx = ldx; a.w1 = w1; a.bn1_s = bn1_s; a.bn1_t = bn1_t; a.bn2_s = bn2_s; a.bn2_t = bn2_t; a.wl = wl;
    a.wa = wa; a.ba = ba; a.wb = wb; a.bb = bb;
    a.T2 = T2; a.cin = cin; a.cin_pad = conv1d_cin_pad(cin); a.dil = dil; a.seg_len = seg_len;
    MV_LAUNCH(cam_dense_layer_kernel, ((unsigned)B, 1, 1), (CD_THREADS, 1, 1), CD_LDS_BYTES, stream, a);
    return check_launch("cam_dense_layer_kernel");
}


// ------------------------------------------------------------------------------------------------------------------------------
// Long utterances (T2 > 160 strided frames = more than 3.2 s of audio): the layer as TWO launches over chunks of 160 frames instead of the
// five launches of the unfused path (r10r: 163 k instead of 368 k audio-seconds per second beyond 3.2 s).  The context couples all frames of an
// utterance, so one grid-wide boundary is needed; everything else stays fused:
//   launch A, workgroup = (chunk, utterance): phase A of cam_dense_layer_kernel on the chunk's rows (both operand streams on LDS-DMA rings,
//            BN1 + ReLU in place, BN2 + ReLU epilogue), h rows to a global fp16 buffer [B, T2, 128], time sums of h per (chunk, 100-frame
//            segment inside the chunk: at most three) to a small fp32 buffer -- no atomics, deterministic;
//   launch B, workgroup = (chunk, utterance): context = total / T2 + segment sum / segment length from those partial sums, the two FCs and
//            the sigmoid for the chunk's (at most three) segments, h rows of the chunk + dil halo rows from the global buffer into LDS
//            (zero outside [0, T2): the k = 3 conv's zero padding), k = 3 conv, gate, 32 new channels of x.
constexpr int CL_MAX_SEG = 3;  // 100-frame segments a 160-frame chunk can touch

struct CamLongArgs {
    half_t* x;            // [B, T2, ldx]
    int64_t ldx;
    const half_t* w1;
    const float *bn1_s, *bn1_t, *bn2_s, *bn2_t;
    const half_t* wl;
    const float *wa, *ba, *wb, *bb;
    half_t* hws;          // [B, T2, 128] fp16
    float* hpart;         // [B, nchunks, CL_MAX_SEG, 128] fp32
    int T2, cin, cin_pad, dil, seg_len, nchunks;
    int chunk_rows;       // rows per chunk: T2 spread evenly over the chunks, a multiple of 16, <= 160
};

__global__ __launch_bounds__(CD_THREADS) void cam_dense_long_gemm_kernel(CamLongArgs a) {
    MV_DYN_SMEM(smem);
    char* xs = smem;
    char* ws = xs + CD_XRING * CD_XS_BYTES;
    char* hbuf = xs;                                            // [160 rows][128] fp16 after the stage loop (40 KiB) ...
    float* part = reinterpret_cast<float*>(xs + CD_ROWS * CD_BN * 2);  // ... then [32][3][128] fp32 partial sums (48 KiB): x ring + W ring are one idle block
    static_assert(CD_ROWS * CD_BN * 2 + 32 * CL_MAX_SEG * CD_BN * 4 <= CD_XRING * CD_XS_BYTES + CD_RING * CD_WS_BYTES, "h and the partial sums live in the idle rings");
    float* fsm = reinterpret_cast<float*>(ws + CD_RING * CD_WS_BYTES);
    float* lbn_s = fsm + CD_MAX_SEG * (CD_BN + 64 + CD_G);
    float* lbn_t = lbn_s + a.cin_pad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int r0 = chunk * a.chunk_rows;
    const int Tn = a.T2 - r0 < a.chunk_rows ? a.T2 - r0 : a.chunk_rows;   // rows of this chunk (>= 1)
    const half_t* xb = a.x + ((int64_t)b * a.T2 + r0) * a.ldx;
    const int nst = a.cin_pad / 64;
    for (int i = tid; i < a.cin_pad; i += CD_THREADS) {
        lbn_s[i] = i < a.cin ? a.bn1_s[i] : 0.0f;
        lbn_t[i] = i < a.cin ? a.bn1_t[i] : 0.0f;
    }
    __syncthreads();
    const int cw = wave & 3, th = wave >> 2;
    float4v e_bn2s[2], e_bn2t[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        e_bn2s[mi] = *reinterpret_cast<const float4v*>(a.bn2_s + (cw * 2 + mi) * 16 + 4 * fg);
        e_bn2t[mi] = *reinterpret_cast<const float4v*>(a.bn2_t + (cw * 2 + mi) * 16 + 4 * fg);
    }
    const int lrow = lane >> 3, kc = (lane & 7) ^ lrow;
    const unsigned xs_addr = lds_addr(xs), ws_addr = lds_addr(ws);
    const unsigned dump_addr = lds_addr(reinterpret_cast<char*>(fsm + CD_MAX_SEG * (CD_BN + 64 + CD_G) + 2 * CD_MAX_CIN));
    const half_t* zero = reinterpret_cast<const half_t*>(g_cd_zero_page);
    const int wave_u = MV_UNIFORM(wave);
    auto issue_x = [&](int s) {
        const bool real = s < nst;
        const unsigned dst = xs_addr + (unsigned)((s & (CD_XRING - 1)) * CD_XS_BYTES);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tr = wave_u + 8 * u;
            int row = tr * 8 + lrow;
            row = row < Tn ? row : Tn - 1;
            const bool live = real && tr < CD_ROWS / 8;  // uniform
            glds16_untracked(live ? xb + (int64_t)row * a.ldx + s * 64 + kc * 8 : zero, live ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    auto issue_w = [&](int s) {
        const bool real = s < nst;
        const unsigned dst = ws_addr + (unsigned)((s % CD_RING) * CD_WS_BYTES);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tr = wave_u * 2 + u;
            const int co = tr * 8 + lrow;
            glds16_untracked(real ? a.w1 + (int64_t)co * a.cin_pad + s * 64 + kc * 8 : zero, real ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    const int xchunk = tid & 7, xrow0 = tid >> 3;
    auto transform = [&](int s) {
        const int c = s * 64 + xchunk * 8;
        const bool live = c < a.cin;
        const float4v s0 = *reinterpret_cast<const float4v*>(lbn_s + c), s1 = *reinterpret_cast<const float4v*>(lbn_s + c + 4);
        const float4v t0 = *reinterpret_cast<const float4v*>(lbn_t + c), t1 = *reinterpret_cast<const float4v*>(lbn_t + c + 4);
        char* tile = xs + (s & (CD_XRING - 1)) * CD_XS_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int row = xrow0 + 64 * p;
            if (row < CD_ROWS) {
                half8v* cell = reinterpret_cast<half8v*>(tile + row * 128 + ((xchunk ^ (row & 7)) << 4));
                const half8v r = *cell;
                half8v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (half_t)fmaxf((float)r[e] * s0[e] + t0[e], 0.0f);
                    o[4 + e] = (half_t)fmaxf((float)r[4 + e] * s1[e] + t1[e], 0.0f);
                }
                if (!(row < Tn && live)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)0.0f;
                }
                *cell = o;
            }
        }
    };
    issue_x(0);
    issue_x(1);
    issue_w(0);
    issue_x(2);
    issue_w(1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        MV_OPAQUE(e_bn2s[mi]);
        MV_OPAQUE(e_bn2t[mi]);
    }
    wait_vm<10>();
    lds_barrier();
    transform(0);
    float4v acc[2][5];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
        wait_vm<5>();
        lds_barrier();
        // one interleaved schedule per stage, as in cam_dense_block_kernel (camblock.hip, r10j): the five transfer requests of x(s+3) / W1(s+2)
        // and the in-place BN1 + ReLU of x(s+1) (fp32 FMA -> fp16, packed max) between the MFMA groups on x(s) / W1(s); the transform of the
        // stage behind the last one works on a dead slot
        const char* wt = ws + (s % CD_RING) * CD_WS_BYTES;
        const char* xt = xs + (s & (CD_XRING - 1)) * CD_XS_BYTES;
        const bool xreal = s + 3 < nst, wreal = s + 2 < nst;
        const unsigned xdst = xs_addr + (unsigned)(((s + 3) & (CD_XRING - 1)) * CD_XS_BYTES);
        const unsigned wdst = ws_addr + (unsigned)(((s + 2) % CD_RING) * CD_WS_BYTES);
        auto dma_x = [&](int u) {
            const int tr = wave_u + 8 * u;
            int row = tr * 8 + lrow;
            row = row < Tn ? row : Tn - 1;
            const bool live = xreal && tr < CD_ROWS / 8;  // uniform
            glds16_untracked(live ? xb + (int64_t)row * a.ldx + (s + 3) * 64 + kc * 8 : zero, live ? xdst + (unsigned)(tr * 1024) : dump_addr);
        };
        auto dma_w = [&](int u) {
            const int tr = wave_u * 2 + u;
            const int co = tr * 8 + lrow;
            glds16_untracked(wreal ? a.w1 + (int64_t)co * a.cin_pad + (s + 2) * 64 + kc * 8 : zero, wreal ? wdst + (unsigned)(tr * 1024) : dump_addr);
        };
        const int tc = (s + 1) * 64 + xchunk * 8;
        const bool tlive = tc < a.cin;
        const float4v ts0 = *reinterpret_cast<const float4v*>(lbn_s + tc), ts1 = *reinterpret_cast<const float4v*>(lbn_s + tc + 4);
        const float4v tt0 = *reinterpret_cast<const float4v*>(lbn_t + tc), tt1 = *reinterpret_cast<const float4v*>(lbn_t + tc + 4);
        char* ttile = xs + ((s + 1) & (CD_XRING - 1)) * CD_XS_BYTES;
        auto cell_ptr = [&](int p) {
            const int row = xrow0 + 64 * p;
            return reinterpret_cast<half8v*>(ttile + row * 128 + ((xchunk ^ (row & 7)) << 4));
        };
        auto cell_math = [&](int p, const half8v& r) {
            const int row = xrow0 + 64 * p;
            half8v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)__builtin_fmaf((float)r[e], ts0[e], tt0[e]);
                o[4 + e] = (half_t)__builtin_fmaf((float)r[4 + e], ts1[e], tt1[e]);
            }
            const half8v z8 = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
            o = __builtin_elementwise_max(o, z8);
            if (!(row < Tn && tlive)) o = z8;
            return o;
        };
        const bool cell2 = xrow0 + 128 < CD_ROWS;
        half8v af[2], bf[5], r0c, r1c, r2c = half8v{}, o0, o1;
        auto load_frags = [&](int kk) {
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
        };
        auto mm = [&](int ni) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        };
        load_frags(0);
        r0c = *cell_ptr(0);
        dma_x(0);
        mm(0);
        mm(1);
        o0 = cell_math(0, r0c);
        dma_x(1);
        mm(2);
        mm(3);
        *cell_ptr(0) = o0;
        r1c = *cell_ptr(1);
        dma_x(2);
        mm(4);
        load_frags(1);
        o1 = cell_math(1, r1c);
        dma_w(0);
        mm(0);
        mm(1);
        *cell_ptr(1) = o1;
        if (cell2) r2c = *cell_ptr(2);
        dma_w(1);
        mm(2);
        mm(3);
        mm(4);
        if (cell2) *cell_ptr(2) = cell_math(2, r2c);
    }
    wait_vm<0>();
    lds_barrier();
    // epilogue: BN2 + ReLU -> hbuf rows [0, Tn) (swizzled 16-byte chunks: chunk ^= row & 15)
    auto h_off = [&](int row, int chunk16) { return row * (CD_BN * 2) + ((chunk16 ^ (row & 15)) << 4); };
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
            if (t < Tn) *reinterpret_cast<half4v*>(hbuf + h_off(t, co >> 3) + (co & 7) * 2) = hv;
        }
    }
    __syncthreads();
    // the chunk's rows of h to the global buffer (16-byte pieces, un-swizzled), and its time sums per segment
    {
        half_t* hdst = a.hws + ((int64_t)b * a.T2 + r0) * CD_BN;
        for (int i = tid; i < Tn * 16; i += CD_THREADS) {
            const int t = i >> 4, c16 = i & 15;
            *reinterpret_cast<half8v*>(hdst + (int64_t)t * CD_BN + c16 * 8) = *reinterpret_cast<const half8v*>(hbuf + h_off(t, c16));
        }
        const int first_seg = r0 / a.seg_len;
        const int cg = tid & 15, rp = tid >> 4;
        float sum[CL_MAX_SEG][8];
#pragma unroll
        for (int sg = 0; sg < CL_MAX_SEG; ++sg)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[sg][e] = 0.0f;
        for (int t = rp; t < Tn; t += 32) {
            const half8v v = *reinterpret_cast<const half8v*>(hbuf + h_off(t, cg));
            const int sg = (r0 + t) / a.seg_len - first_seg;
#pragma unroll
            for (int q = 0; q < CL_MAX_SEG; ++q)
                if (sg == q) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[q][e] += (float)v[e];
                }
        }
#pragma unroll
        for (int sg = 0; sg < CL_MAX_SEG; ++sg)
#pragma unroll
            for (int e = 0; e < 8; ++e) part[(rp * CL_MAX_SEG + sg) * CD_BN + cg * 8 + e] = sum[sg][e];
        __syncthreads();
        if (tid < CL_MAX_SEG * CD_BN) {
            float v = 0.0f;
            for (int p = 0; p < 32; ++p) v += part[p * CL_MAX_SEG * CD_BN + tid];   // tid = sg * 128 + c
            a.hpart[((int64_t)b * a.nchunks + chunk) * CL_MAX_SEG * CD_BN + tid] = v;
        }
    }
}

__global__ __launch_bounds__(CD_THREADS) void cam_dense_long_conv_kernel(CamLongArgs a) {
    __shared__ __attribute__((aligned(16))) char hbuf[(CD_ROWS + 2 * CD_PAD) * CD_BN * 2];
    __shared__ float ctx[CL_MAX_SEG * CD_BN], g1[CL_MAX_SEG * 64], gate[CL_MAX_SEG * CD_G];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int T2 = a.T2;
    const int r0 = chunk * a.chunk_rows;
    const int Tn = T2 - r0 < a.chunk_rows ? T2 - r0 : a.chunk_rows;
    const int first_seg = r0 / a.seg_len;
    auto h_off = [&](int row, int chunk16) { return row * (CD_BN * 2) + ((chunk16 ^ (row & 15)) << 4); };
    // parameters first (their latency runs under the staging of h)
    float4v e_wa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e_wa[u] = *reinterpret_cast<const float4v*>(a.wa + (tid >> 3) * CD_BN + (tid & 7) * 16 + 4 * u);
    const float4v e_wb = *reinterpret_cast<const float4v*>(a.wb + (tid >> 4) * 64 + (tid & 15) * 4);
    const float e_ba = a.ba[tid >> 3], e_bb = a.bb[tid >> 4];
    half8v e_wl[3][4];
    {
        const half_t* wrow = a.wl + (int64_t)((wave & 1) * 16 + fr) * 3 * CD_BN + 8 * fg;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e_wl[tap][kk] = *reinterpret_cast<const half8v*>(wrow + tap * CD_BN + kk * 32);
    }
    // h rows [r0 - PAD, r0 + 160 + PAD) into LDS (row index + PAD), zeros outside [0, T2)
    {
        const half_t* hsrc = a.hws + (int64_t)b * T2 * CD_BN;
        for (int i = tid; i < (CD_ROWS + 2 * CD_PAD) * 16; i += CD_THREADS) {
            const int row = i >> 4, c16 = i & 15;
            const int t = r0 + row - CD_PAD;
            half8v v = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
            if (t >= 0 && t < T2 && row - CD_PAD < Tn + CD_PAD) v = *reinterpret_cast<const half8v*>(hsrc + (int64_t)t * CD_BN + c16 * 8);
            *reinterpret_cast<half8v*>(hbuf + h_off(row, c16)) = v;
        }
    }
    // context of the chunk's segments: total over all chunks / T2 + segment sum / segment length
    if (tid < CL_MAX_SEG * CD_BN) {
        const int sg = tid / CD_BN, c = tid - sg * CD_BN;
        const int seg = first_seg + sg;
        const float* hp = a.hpart + (int64_t)b * a.nchunks * CL_MAX_SEG * CD_BN + c;
        float total = 0.0f, mine = 0.0f;
        for (int cc = 0; cc < a.nchunks; ++cc) {
            const int fs = cc * a.chunk_rows / a.seg_len;
#pragma unroll
            for (int q = 0; q < CL_MAX_SEG; ++q) {
                const float v = hp[(cc * CL_MAX_SEG + q) * CD_BN];
                total += v;
                if (fs + q == seg) mine += v;
            }
        }
        const int t0 = seg * a.seg_len;
        const int len = (t0 + a.seg_len < T2 ? t0 + a.seg_len : T2) - t0;
        ctx[sg * CD_BN + c] = len > 0 ? total / (float)T2 + mine / (float)len : 0.0f;
    }
    __syncthreads();
    {
        const int j = tid >> 3, part8 = tid & 7;
#pragma unroll
        for (int sg = 0; sg < CL_MAX_SEG; ++sg) {
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
            v += dpp_mov<DPP_ROW_HALF_MIRROR>(0.0f, v);
            if (part8 == 0) g1[sg * 64 + j] = fmaxf(v + e_ba, 0.0f);
        }
    }
    __syncthreads();
    {
        const int co = tid >> 4, part16 = tid & 15;
#pragma unroll
        for (int sg = 0; sg < CL_MAX_SEG; ++sg) {
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
    {
        half_t* xb = a.x + ((int64_t)b * T2 + r0) * a.ldx;
        // channel tile ct, time tiles tg, tg + 4, tg + 8; waves of time groups 2 / 3 own two tiles: their third accumulator repeats tile 9 and is never
        // stored (twelve spare MFMAs instead of a uniform branch -- and a drained LDS queue -- per K step, camblock.hip)
        const int ct = wave & 1, tg = wave >> 1;
        const bool third = tg + 8 < CD_TT;
        int trow[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) trow[j] = (j < 2 || third ? tg + 4 * j : CD_TT - 1) * 16 + fr;
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
                    const int row = trow[j] + (tap - 1) * a.dil + CD_PAD;
                    const half8v bfr = *reinterpret_cast<const half8v*>(hbuf + h_off(row, kk * 4 + fg));
                    yc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr, yc[j], 0, 0, 0);
                }
            }
        }
        const int co = ct * 16 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int t = trow[j];
            if ((j < 2 || third) && t < Tn) {
                const float* gt = gate + ((r0 + t) / a.seg_len - first_seg) * CD_G + co;
                half4v hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (half_t)fmed3(yc[j][r] * gt[r], -65504.0f, 65504.0f);
                *reinterpret_cast<half4v*>(xb + (int64_t)t * a.ldx + a.cin + co) = hv;
            }
        }
    }
}

bool cam_dense_long_supported(int T2, int cin, int bottleneck, int growth, int dil, int seg_len) {
    // (a 160-frame chunk touches at most three segments when a segment is at least 80 frames long)
    return bottleneck == CD_BN && growth == CD_G && T2 > CD_ROWS && cin % 32 == 0 && cin >= 32 && cin <= CD_MAX_CIN - 64 && dil >= 1 && dil <= CD_PAD &&
           seg_len >= 80;
}

int64_t cam_dense_long_part_floats(int B, int T2) { return (int64_t)B * (ceil_div(T2, CD_ROWS) + 4) * CL_MAX_SEG * CD_BN; }  // (+ 4: the launcher may take up to four more chunks)

int cam_dense_long_launch(half_t* x, int64_t ldx, int B, int T2, int cin, const half_t* w1, const float* bn1_s, const float* bn1_t,
                          const float* bn2_s, const float* bn2_t, const half_t* wl, const float* wa, const float* ba, const float* wb,
                          const float* bb, int dil, int seg_len, half_t* hws, float* hpart, hipStream_t stream) {
    MV_REQUIRE(cam_dense_long_supported(T2, cin, CD_BN, CD_G, dil, seg_len), "cam_dense_long: unsupported geometry");
    MV_REQUIRE(ldx >= cin + CD_G && (ldx % 8) == 0, "cam_dense_long: the row must hold the inputs and 32 new channels (16-byte aligned chunks)");
    MV_REQUIRE(hws != nullptr && hpart != nullptr && (reinterpret_cast<uintptr_t>(hws) & 15) == 0, "cam_dense_long: workspace");
    MV_REQUIRE(B <= 65535, "cam_dense_long: batch too large for one launch (grid.y)");
    static DeviceOnce smem_set;   // (per device: the attribute belongs to the current device's code object)
    int smem_set_slot;
    if (device_once_pending(smem_set, &smem_set_slot)) {
        if (MV_SET_MAX_SMEM(cam_dense_long_gemm_kernel, CD_LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "cam_dense_long: cannot reserve LDS");
        device_once_done(smem_set, smem_set_slot);
    }
    CamLongArgs a;
    a.x = x; a.ldx = ldx; a.w1 = w1; a.bn1_s = bn1_s; a.bn1_t = bn1_t; a.bn2_s = bn2_s; a.bn2_t = bn2_t; a.wl = wl;
    a.wa = wa; a.ba = ba; a.wb = wb; a.bb = bb; a.hws = hws; a.hpart = hpart;
    a.T2 = T2; a.cin = cin; a.cin_pad = conv1d_cin_pad(cin); a.dil = dil; a.seg_len = seg_len;
    // Even chunks (165 frames: 96 + 69, not 160 + 5), and as many of them as make the rounds of workgroups cheapest: one workgroup per CU at a time,
    // a workgroup's time ~ 0.45 + 0.55 * rows / 160 of a full one (r10s: the per-stage cost is mostly fixed), so 76 utterances x 4 chunks of 125 rows
    // (304 workgroups: a second round for 48 of them) lose to 6 chunks of 96 (456: two rounds of cheaper workgroups).
    {
        const int cus = device_cu_count();
        const int nmin = (int)ceil_div(T2, CD_ROWS);
        double best = 0.0;
        a.nchunks = nmin;
        a.chunk_rows = (int)round_up(ceil_div(T2, nmin), 16);
        for (int n = nmin; n <= nmin + 4; ++n) {
            const int rows = (int)round_up(ceil_div(T2, n), 16);
            const int chunks = (int)ceil_div(T2, rows);
            const double cost = (double)ceil_div((int64_t)B * chunks, cus > 0 ? cus : 256) * (0.45 + 0.55 * rows / CD_ROWS);
            if (n == nmin || cost < best * 0.97) {   // (3 % margin: stay with fewer workgroups on a tie)
                best = cost;
                a.nchunks = chunks;
                a.chunk_rows = rows;
            }
        }
    }
    MV_LAUNCH(cam_dense_long_gemm_kernel, ((unsigned)a.nchunks, (unsigned)B, 1), (CD_THREADS, 1, 1), CD_LDS_BYTES, stream, a);
    int rc = check_launch("cam_dense_long_gemm_kernel");
    if (rc != MV_OK) return rc;
    MV_LAUNCH(cam_dense_long_conv_kernel, ((unsigned)a.nchunks, (unsigned)B, 1), (CD_THREADS, 1, 1), 0, stream, a);
    return check_launch("cam_dense_long_conv_kernel");
}

}  // namespace mv
