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

#include <type_traits>

namespace mv {

constexpr int CV_TC = 128;  // output channels per tile
constexpr int CV_TN = 128;  // time steps per tile
constexpr int CV_BK = 64;   // K elements per stage
constexpr int CV_THREADS = 256;
constexpr int CV_LDS_BYTES = 2 * (CV_TC + CV_TN) * CV_BK * 2;  // 64 KiB
constexpr int CV_LDS_BYTES_WIDE = 2 * (CV_TC + 160) * CV_BK * 2;  // 128 x 160 tile: 72 KiB
constexpr int CV_SMALL_NS = 4;                                     // K stages of the 64 x 64 kernel's ring (three in flight)
constexpr int CV_LDS_BYTES_SMALL = CV_SMALL_NS * (64 + 64) * CV_BK * 2;   // 64 x 64 tile (small problems): 64 KiB
constexpr int CV_TAIL_NS = 4;                                      // K stages of the 128 x 128 kernel's ring when it runs the ring GEMM's tail (conv1d_launch)
constexpr int CV_LDS_BYTES_TAIL = CV_TAIL_NS * (CV_TC + CV_TN) * CV_BK * 2;   // 128 KiB: one workgroup per CU
constexpr int CV_LDS_BYTES_BIG = 256 * (256 * 2 + 8);          // 256^2 tile: 2 x 64 KiB stages, 130 KiB staged epilogue

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
    const half_t* add_src;  // optional second output: sum_dst = y + add_src
    half_t* sum_dst;
    int64_t ld_add, ld_sum;
    int B, T_in, T_out, cin, cin_pad, cout, cout_pad, k, dil, stride, pad, pad_mode;
    int pre_act, post_act, y_f16, gate_seg_len, gate_nseg;
    int n_rows, n_tiles, co_tiles;
    float* stat_sum;  // optional partial time sums of the output (persistent kernel): [ceil(n_rows / 64)][2][cout]
    float* stat_sq;   // optional partial sums of squares (about the BatchNorm shift), same layout
    // one-shot 128 x 160 kernel: tiles never straddle utterances (tile = (utterance, 160-frame block)), and the workgroups of channel
    // tile 0 leave the time sums / sums of squares of their x tiles in in_sum / in_sq: [B * tiles_per_utt][cin]
    int per_utt, tiles_per_utt;
    float* in_sum;
    float* in_sq;
    unsigned long long* clock_probe;  // optional (ring kernel): [workgroup][4] shader-clock / reference ticks at entry and exit
    // The ring walk's last PARTIAL round as quarter tiles (conv1d_launch, "tail"): the ring kernel stops at virtual tile id ring_vb_end (0: walks all),
    // and a launch of the 128 x 128 kernel takes the tail_nv virtual ids from tail_vb0 on -- workgroup b computes quarter b / tail_nv of the
    // 256 x 256 tile (tail_n_tiles x tail_co_tiles geometry) with virtual id tail_vb0 + b % tail_nv.  tail_nv == 0: the kernel's own tile walk.
    int ring_vb_end = 0;
    int tail_vb0 = 0, tail_nv = 0, tail_n_tiles = 0, tail_co_tiles = 0, tail_split = 2;   // tail_split: sub-tiles per side of a 256 x 256 tile (2: 128 x 128, 4: 64 x 64)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MV_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MV_ACT_TANH) return tanhf(v);
    if (act == MV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][64] fp16 tile
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// 256 zero bytes: source of every padded / out-of-range 16-byte chunk, so the loaders never branch
__device__ __attribute__((aligned(256))) const unsigned char g_zero_page[256] = {0};

// 256 ones: stands in for an absent BatchNorm scale in the persistent kernel's parameter slots
__device__ __attribute__((aligned(256))) const float g_one_page[256] = {
#define MV_ONE16 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f
    MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16,
    MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16, MV_ONE16
#undef MV_ONE16
};

__device__ __forceinline__ half_t to_half_sat(float v) {
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);  // saturate instead of producing inf
    return (half_t)v;
}

__device__ __forceinline__ float clamp3(float v, float lo, float hi) { return fmed3(v, lo, hi); }

// (max_raw = max as exactly one v_max_f32, glds16 = one 16-byte global -> LDS transfer per lane, wait_vm<N>: arch/gfx950.h)
__device__ __forceinline__ void wait_all_loads() { wait_vm<0>(); }

struct RowMap {  // where the rows (time steps) of this lane live
    int b, t;    // b < 0: row beyond the tensor
};

// input time index of output step t for tap `tap`; returns -1 for zero padding
__device__ __forceinline__ int input_time(const ConvArgs& a, int t, int tap) {
    int tin = t * a.stride - a.pad + tap * a.dil;
    if (tin < 0 || tin >= a.T_in) {
        if (a.pad_mode == MV_PAD_REFLECT)
            tin = tin < 0 ? -tin : 2 * (a.T_in - 1) - tin;
        else
            tin = -1;
    }
    return tin;
}

// ---- shared MFMA stage and epilogue ---------------------------------------------------------------------------
// Wave tile = MI x NI MFMA tiles of 16 x 16; wc / wn = position of the wave inside the workgroup tile.
template <int MI, int NI>
__device__ __forceinline__ void mma_half_stage(const char* wt, const char* xtile, int wc, int wn, int lane, int kk, float4v (&acc)[MI][NI]) {
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    half8v af[MI], bf[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const half8v*>(wt + lds_off(wc * (MI * 16) + mi * 16 + frow, kk * 4 + fchunk));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const half8v*>(xtile + lds_off(wn * (NI * 16) + ni * 16 + frow, kk * 4 + fchunk));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
}

template <int MI, int NI>
__device__ __forceinline__ void mma_stage(const char* wt, const char* xtile, int wc, int wn, int lane, float4v (&acc)[MI][NI]) {
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        half8v af[MI], bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            af[mi] = *reinterpret_cast<const half8v*>(wt + lds_off(wc * (MI * 16) + mi * 16 + frow, kk * 4 + fchunk));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            bf[ni] = *reinterpret_cast<const half8v*>(xtile + lds_off(wn * (NI * 16) + ni * 16 + frow, kk * 4 + fchunk));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
}

// ---- hand-scheduled K stage of the 256 x 256 tile (wave tile 128 x 64: 8 x 4 MFMA tiles, 2 K halves) ---------------
// The compiler serialises "2 ds_read, s_waitcnt lgkmcnt(0), 8 MFMA" and so exposes an LDS round trip in front of every
// group of MFMAs.  Here the stage is eight steps (K half, pair of channel tiles) built from the arch header's inline-assembly
// pieces (mfma8_step = counted wait + 8 MFMAs, lds_read2 / lds_read4 = fragment requests without a wait): the two weight
// fragments of step i+1 are requested BEFORE the counted wait and the 8 MFMAs of step i, so the round trip runs under
// the matrix pipe.  Rules kept by construction: a fragment register is only re-targeted by a read that is issued after
// the last MFMA that sources it; LDS returns in order, so lgkmcnt(2) = "everything but the two newest reads".
// Fragment addresses: row * 128 + ((chunk ^ (row & 7)) << 4) with row = tile base + frow -- the swizzle term depends on
// the lane and the K half only, the channel / time tile is an immediate multiple of 2048.

// wt / xtile: LDS byte addresses of the stage's weight and activation tiles.  between(i), i = 0..7, runs between the
// fragment requests and the (counted wait +) MFMAs of step i: the caller issues one of the next stage's eight global->LDS transfers there,
// so the ~80 cycles each transfer spends entering the texture path pass while the matrix pipe works off queued MFMAs
// instead of in front of them (a K stage otherwise starts with ~640 cycles of transfer issue and an idle matrix pipe).
template <class F>
__device__ __forceinline__ void mma_stage_8x4(unsigned wt, unsigned xtile, int wc, int wn, int lane, float4v (&acc)[8][4], F&& between) {
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned arow = wt + (wc * 128 + frow) * 128, brow = xtile + (wn * 64 + frow) * 128;
    const unsigned sw0 = (unsigned)((fchunk ^ (frow & 7)) << 4), sw1 = (unsigned)(((4 + fchunk) ^ (frow & 7)) << 4);
    const unsigned a0 = arow + sw0, a1 = arow + sw1, b0 = brow + sw0, b1 = brow + sw1;
    half8v af[2][2], bf[4];
    // K half 0
    lds_read4(bf, b0);
    lds_read2<0, 2048>(af[0][0], af[0][1], a0);
    lds_read2<4096, 6144>(af[1][0], af[1][1], a0);
    between(0);
    mfma8_step<2>(acc[0], acc[1], af[0][0], af[0][1], bf);
    lds_read2<8192, 10240>(af[0][0], af[0][1], a0);
    between(1);
    mfma8_step<2>(acc[2], acc[3], af[1][0], af[1][1], bf);
    lds_read2<12288, 14336>(af[1][0], af[1][1], a0);
    between(2);
    mfma8_step<2>(acc[4], acc[5], af[0][0], af[0][1], bf);
    lds_read2<0, 2048>(af[0][0], af[0][1], a1);  // first pair of K half 1: its registers were last sourced by the step above
    between(3);
    mfma8_step<2>(acc[6], acc[7], af[1][0], af[1][1], bf);
    // K half 1: the activation fragments are single-buffered, so they are re-targeted only now
    lds_read4(bf, b1);
    lds_read2<4096, 6144>(af[1][0], af[1][1], a1);
    between(4);
    mfma8_step<2>(acc[0], acc[1], af[0][0], af[0][1], bf);
    lds_read2<8192, 10240>(af[0][0], af[0][1], a1);
    between(5);
    mfma8_step<2>(acc[2], acc[3], af[1][0], af[1][1], bf);
    lds_read2<12288, 14336>(af[1][0], af[1][1], a1);
    between(6);
    mfma8_step<2>(acc[4], acc[5], af[0][0], af[0][1], bf);
    between(7);
    mfma8_step<0>(acc[6], acc[7], af[1][0], af[1][1], bf);
    mfma_hazard_pad();
}

// The same stage with its LAST group of eight MFMAs carried across the stage barrier (round 6; ring kernel).  In the form above a stage ends with
// its step-7 MFMAs and the next one begins -- behind the counted wait and the barrier -- with six fragment reads whose LDS round trip nothing
// covers: the matrix pipe idles from the barrier to the first fragments (~300 of a stage's ~2950 cycles in the r05 timeline, on top of the barrier
// skew itself).  Here a stage that is followed by another stage of the same tile (HOLD) leaves step 7 undone -- its four B and two A fragments
// are in registers, complete since the barrier's lgkmcnt(0) -- and the next stage (PENDING) issues those eight MFMAs right behind its own first
// six fragment requests, so the pipe works through them while the requests travel.  That needs the activation fragments double-buffered by
// K half (bf0 / bf1: +16 registers; the weight fragments already alternate) and the fragment registers to live in the caller's scope.
// Accumulation order per accumulator is unchanged (K halves in order, stages in order): bit-identical results.
// A tile's first stage has nothing pending, its last stage holds nothing back (the epilogue reads complete accumulators).
template <bool PENDING, bool HOLD, class F>
__device__ __forceinline__ void mma_stage_8x4_carry(unsigned wt, unsigned xtile, int wc, int wn, int lane, float4v (&acc)[8][4], half8v (&af)[2][2],
                                                    half8v (&bf0)[4], half8v (&bf1)[4], F&& between) {
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned arow = wt + (wc * 128 + frow) * 128, brow = xtile + (wn * 64 + frow) * 128;
    const unsigned sw0 = (unsigned)((fchunk ^ (frow & 7)) << 4), sw1 = (unsigned)(((4 + fchunk) ^ (frow & 7)) << 4);
    const unsigned a0 = arow + sw0, a1 = arow + sw1, b0 = brow + sw0, b1 = brow + sw1;
    // K half 0
    lds_read4(bf0, b0);
    lds_read2<0, 2048>(af[0][0], af[0][1], a0);
    if constexpr (PENDING) mfma8_step<6>(acc[6], acc[7], af[1][0], af[1][1], bf1);   // step 7 of the previous stage (no wait: six reads may be out)
    lds_read2<4096, 6144>(af[1][0], af[1][1], a0);
    between(0);
    mfma8_step<2>(acc[0], acc[1], af[0][0], af[0][1], bf0);
    lds_read2<8192, 10240>(af[0][0], af[0][1], a0);
    between(1);
    mfma8_step<2>(acc[2], acc[3], af[1][0], af[1][1], bf0);
    lds_read2<12288, 14336>(af[1][0], af[1][1], a0);
    between(2);
    mfma8_step<2>(acc[4], acc[5], af[0][0], af[0][1], bf0);
    lds_read2<0, 2048>(af[0][0], af[0][1], a1);
    lds_read4(bf1, b1);   // K half 1's activation fragments one step early: bf1 is free since the carried group was issued (the single-buffered form had to wait for step 3)
    between(3);
    mfma8_step<6>(acc[6], acc[7], af[1][0], af[1][1], bf0);
    // K half 1
    lds_read2<4096, 6144>(af[1][0], af[1][1], a1);
    between(4);
    mfma8_step<2>(acc[0], acc[1], af[0][0], af[0][1], bf1);
    lds_read2<8192, 10240>(af[0][0], af[0][1], a1);
    between(5);
    mfma8_step<2>(acc[2], acc[3], af[1][0], af[1][1], bf1);
    lds_read2<12288, 14336>(af[1][0], af[1][1], a1);
    between(6);
    mfma8_step<2>(acc[4], acc[5], af[0][0], af[0][1], bf1);
    between(7);
    if constexpr (!HOLD) {
        mfma8_step<0>(acc[6], acc[7], af[1][0], af[1][1], bf1);
        mfma_hazard_pad();
    }
}

// lane holds channels co..co+3 (rows) of time step n (column); cout is a multiple of 4, so a lane's four channels are
// all valid or all invalid and every per-channel parameter is one float4 load (uniform branches only).
__device__ __forceinline__ float4v act4(float4v v, int act) {
    if (act == MV_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
    } else if (act == MV_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
    } else if (act == MV_ACT_SIGMOID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
    }
    return v;
}

// n_end: first row behind the tile's valid rows (the tensor's row count, or the end of the tile's utterance)
// Accumulators start from the bias (a lane holds channels co .. co + 3 of tile mi for every time tile): EVERY conv kernel of this file
// does, the persistent ones included, so that acc = bias + sum over K in one order whatever tile shape the launcher picks -- a row's
// result does not depend on the batch it sits in.
template <int MI, int NI>
__device__ __forceinline__ void conv_acc_init(const ConvArgs& a, int co0, int wc, int lane, float4v (&acc)[MI][NI]) {
    const int crow = 4 * (lane >> 4);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int co = co0 + wc * (MI * 16) + mi * 16 + crow;
        const float4v b4 = (a.bias != nullptr && co < a.cout) ? *reinterpret_cast<const float4v*>(a.bias + co) : float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = b4;
    }
}

template <int MI, int NI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, int n0, int n_end, int co0, int wc, int wn, int lane,
                                              float4v (&acc)[MI][NI]) {
    const int crow = 4 * (lane >> 4);
    int nn[NI], nb[NI], nt[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        nn[ni] = n0 + wn * (NI * 16) + ni * 16 + (lane & 15);
        nb[ni] = 0;
        nt[ni] = 0;
        if (a.row_bias != nullptr || a.gate != nullptr) {
            nb[ni] = nn[ni] / a.T_out;
            nt[ni] = nn[ni] - nb[ni] * a.T_out;
        }
    }
    const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    const float4v one4 = float4v{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int co = co0 + wc * (MI * 16) + mi * 16 + crow;
        if (co >= a.cout) continue;
        const float4v scale4 = a.scale != nullptr ? *reinterpret_cast<const float4v*>(a.scale + co) : one4;
        const float4v shift4 = a.scale != nullptr ? *reinterpret_cast<const float4v*>(a.shift + co) : zero4;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = nn[ni];
            if (n >= n_end) continue;
            float4v v = acc[mi][ni];   // (the bias is already in the accumulators: conv_acc_init)
            if (a.row_bias != nullptr) v += *reinterpret_cast<const float4v*>(a.row_bias + (int64_t)nb[ni] * a.cout + co);
            v = act4(v, a.pre_act);
            v = v * scale4 + shift4;
            v = act4(v, a.post_act);
            if (a.gate != nullptr)
                v *= *reinterpret_cast<const float4v*>(a.gate + ((int64_t)nb[ni] * a.gate_nseg + nt[ni] / a.gate_seg_len) * a.cout + co);
            half4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = to_half_sat(v[r]);
            if (a.y_f16) {
                *reinterpret_cast<half4v*>(reinterpret_cast<half_t*>(a.y) + (int64_t)n * a.ldy + co) = hv;
            } else {
                *reinterpret_cast<float4v*>(reinterpret_cast<float*>(a.y) + (int64_t)n * a.ldy + co) = v;
            }
            if (a.sum_dst != nullptr) {
                // second output: y + add_src (the next Res2Net step's input x_{j+1} + y_j, ecapa_tdnn.py:47)
                const half4v sv = *reinterpret_cast<const half4v*>(a.add_src + (int64_t)n * a.ld_add + co);
                half4v ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = to_half_sat((float)hv[r] + (float)sv[r]);
                *reinterpret_cast<half4v*>(a.sum_dst + (int64_t)n * a.ld_sum + co) = ov;
            }
        }
    }
}

// Epilogue for fp16 outputs without a second output: the finished tile is staged in LDS (free once the K loop has passed
// its last barrier) as [TN rows][TC channels] with an 8-byte row pad, then written with 16-byte stores that cover whole
// contiguous row segments (TC*2 bytes).  The direct form stores 8 bytes per lane into 16 different rows per instruction,
// which the PMC run showed as 1.7x write amplification at the memory side (WRITE_SIZE 790 MB for a 468 MB tensor).
template <int MI, int NI, int TC, int TN, int NTHREADS>
__device__ __forceinline__ void conv_epilogue_staged(const ConvArgs& a, char* smem, int n0, int n_end, int co0, int wc, int wn, int lane,
                                                     int tid, float4v (&acc)[MI][NI]) {
    constexpr int ROWB = TC * 2 + 8;
    const int crow = 4 * (lane >> 4);
    int nl[NI], nb[NI], nt[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        nl[ni] = wn * (NI * 16) + ni * 16 + (lane & 15);
        const int n = n0 + nl[ni];
        nb[ni] = 0;
        nt[ni] = 0;
        if ((a.row_bias != nullptr || a.gate != nullptr) && n < n_end) {
            nb[ni] = n / a.T_out;
            nt[ni] = n - nb[ni] * a.T_out;
        }
    }
    const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    const float4v one4 = float4v{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int col = wc * (MI * 16) + mi * 16 + crow;
        const int co = co0 + col;
        const bool cok = co < a.cout;
        const float4v scale4 = (cok && a.scale != nullptr) ? *reinterpret_cast<const float4v*>(a.scale + co) : one4;
        const float4v shift4 = (cok && a.scale != nullptr) ? *reinterpret_cast<const float4v*>(a.shift + co) : zero4;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const bool ok = cok && n0 + nl[ni] < n_end;
            float4v v = acc[mi][ni];   // (the bias is already in the accumulators: conv_acc_init)
            if (a.row_bias != nullptr && ok) v += *reinterpret_cast<const float4v*>(a.row_bias + (int64_t)nb[ni] * a.cout + co);
            v = act4(v, a.pre_act);
            v = v * scale4 + shift4;
            v = act4(v, a.post_act);
            if (a.gate != nullptr && ok)
                v *= *reinterpret_cast<const float4v*>(a.gate + ((int64_t)nb[ni] * a.gate_nseg + nt[ni] / a.gate_seg_len) * a.cout + co);
            half4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = to_half_sat(v[r]);
            *reinterpret_cast<half4v*>(smem + nl[ni] * ROWB + col * 2) = hv;
        }
    }
    __syncthreads();
    // 16-byte chunks: CPRW per row; consecutive threads walk a row, then the next row
    constexpr int CPRW = TC / 8;
    half_t* y = reinterpret_cast<half_t*>(a.y);
    const bool vec_ok = (a.ldy % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    for (int i = tid; i < TN * CPRW; i += NTHREADS) {
        const int row = i / CPRW, ch = i - row * CPRW;
        const int n = n0 + row;
        const int co = co0 + ch * 8;
        if (n >= n_end || co >= a.cout) continue;
        const char* src = smem + row * ROWB + ch * 16;
        half_t* dst = y + (int64_t)n * a.ldy + co;
        if (vec_ok && co + 8 <= a.cout) {
            // LDS rows are 8-byte aligned only: read as two 8-byte halves
            const half4v lo = *reinterpret_cast<const half4v*>(src);
            const half4v hi = *reinterpret_cast<const half4v*>(src + 8);
            half8v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = lo[e];
                o[e + 4] = hi[e];
            }
            *reinterpret_cast<half8v*>(dst) = o;
        } else {
            *reinterpret_cast<half4v*>(dst) = *reinterpret_cast<const half4v*>(src);
            if (co + 8 <= a.cout) *reinterpret_cast<half4v*>(dst + 4) = *reinterpret_cast<const half4v*>(src + 8);
        }
    }
}

__host__ __device__ __forceinline__ bool tile_of_index_geom(int n_tiles, int co_tiles, int bid, int& n_tile, int& co_tile) {
    // XCD-aware super-tiles over a DENSE id space: ids 0 .. n_tiles * co_tiles - 1 are all tiles (round 6: the walk used to run over
    // round_up(n_tiles, 8) * co_tiles virtual ids with holes where an XCD owned one n-tile fewer -- 48 of 3648 on the MFA layer of the headline batch,
    // which left 32 persistent workgroups a tile short and 48 instead of 16 tiles in the last partial round).
    // Workgroup ids are dealt round-robin to the 8 XCDs (id mod 8), each with a private L2.  XCD x owns the n-tiles x, x+8, ... of the `nfull` complete
    // rows of eight; inside an XCD the blocks walk groups of <= 8 co-tiles: for each group, for each owned n-tile, for each co-tile of the group.  The
    // ~64 workgroups resident on an XCD therefore cover ~8 n-tiles x 8 co-tiles and stream K in near lockstep, so every activation slice and every
    // weight slice fetched into that L2 is reused ~8 times before it is evicted.  The n_tiles % 8 leftover n-tiles come last, co-tile by co-tile.
    const int nfull = n_tiles >> 3, rest = n_tiles & 7;
    const int main_ids = nfull * 8 * co_tiles;
    if (bid >= main_ids) {
        const int j = bid - main_ids;
        if (j >= rest * co_tiles) return false;
        co_tile = j / rest;
        n_tile = 8 * nfull + j - co_tile * rest;
        return true;
    }
    const int xcd = bid & 7;
    const int seq = bid >> 3;
    const int group = seq / (nfull * 8);          // full groups come first
    const int base = group * 8;
    const int gw = co_tiles - base < 8 ? co_tiles - base : 8;
    const int idx = seq - nfull * base;
    const int n_local = idx / gw;
    co_tile = base + idx - n_local * gw;
    n_tile = xcd + 8 * n_local;
    return true;
}

__device__ __forceinline__ bool tile_of_index(const ConvArgs& a, int bid, int& n_tile, int& co_tile) {
    return tile_of_index_geom(a.n_tiles, a.co_tiles, bid, n_tile, co_tile);
}

__device__ __forceinline__ bool tile_of_block(const ConvArgs& a, int& n_tile, int& co_tile) {
    if (a.tail_nv > 0) {   // (uniform) a sub-tile of a 256 x 256 tile of the ring walk's last partial round; tail_vb0 % 8 == 0, so
                           // workgroup sits on the XCD the ring walk gives that tile (id mod 8)
        const int sub = (int)blockIdx.x / a.tail_nv, v = (int)blockIdx.x - sub * a.tail_nv;
        int nt = 0, ct = 0;
        if (!tile_of_index_geom(a.tail_n_tiles, a.tail_co_tiles, a.tail_vb0 + v, nt, ct)) return false;
        n_tile = a.tail_split * nt + (sub & (a.tail_split - 1));
        co_tile = a.tail_split * ct + sub / a.tail_split;
        return n_tile < a.n_tiles;
    }
    return tile_of_index(a, blockIdx.x, n_tile, co_tile);
}

// Time sums of one landed x stage ([TN rows][64 channels] in LDS; rows beyond the utterance came from the zero page and add nothing)
// for the fused input statistics of a 1x1 layer: wave w takes the 16-byte chunks 2w and 2w + 1 (16 channels), its two 32-lane halves
// one chunk each, lane g of a half the rows g, g + 32, ...; sums and sums of squares in fp32 (v_pk_add / v_pk_fma), reduced over the
// 32 lanes (DPP row sums + one cross-row exchange) and written by one lane per half: 2 x 32 bytes per (tile, chunk).
struct InStats {
    float2v s1[4], s2[4];
};

template <int TN>
__device__ __forceinline__ void input_stats_accumulate(InStats& st, const char* xtile, int wave, int lane) {
    const int chunk = 2 * wave + (lane >> 5), g = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) st.s1[q] = st.s2[q] = float2v{0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < TN / 32; ++p) {
        const int row = g + 32 * p;
        const half8v v = *reinterpret_cast<const half8v*>(xtile + lds_off(row, chunk));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2v f = float2v{(float)v[2 * q], (float)v[2 * q + 1]};
            st.s1[q] += f;
            st.s2[q] = __builtin_elementwise_fma(f, f, st.s2[q]);
        }
    }
}

__device__ __forceinline__ void input_stats_reduce_store(const ConvArgs& a, const InStats& st, int c0, int n_tile, int wave, int lane) {
    const int chunk = 2 * wave + (lane >> 5), g = lane & 31;
    // row sums in every lane of a 16-lane row, then the odd rows add the even row in front of them (row_bcast:15: lane 15 of the
    // previous row): lanes 16..31 / 48..63 hold the sums of their 32-lane half -- all DPP, no LDS round trip
    float r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // (MV_OPAQUE: keeps the vectoriser from pairing the sixteen reductions into v_pk_add_f32, which cannot carry a DPP operand --
        // every step would become v_mov 0, v_mov_b32_dpp and half a packed add instead of one v_add_f32_dpp)
        float t1 = st.s1[e >> 1][e & 1], t2 = st.s2[e >> 1][e & 1];
        MV_OPAQUE(t1);
        MV_OPAQUE(t2);
        t1 = row16_sum(t1);
        t2 = row16_sum(t2);
        t1 += dpp_mov<DPP_ROW_BCAST15>(0.0f, t1);
        t2 += dpp_mov<DPP_ROW_BCAST15>(0.0f, t2);
        MV_OPAQUE(t1);
        MV_OPAQUE(t2);
        r1[e] = t1;
        r2[e] = t2;
    }
    const int c = c0 + chunk * 8;
    if (g == 16 && c < a.cin) {  // cin % 8 == 0: a chunk is inside or outside
        float* ps = a.in_sum + (int64_t)n_tile * a.cin + c;
        float* pq = a.in_sq + (int64_t)n_tile * a.cin + c;
        *reinterpret_cast<float4v*>(ps) = float4v{r1[0], r1[1], r1[2], r1[3]};
        *reinterpret_cast<float4v*>(ps + 4) = float4v{r1[4], r1[5], r1[6], r1[7]};
        *reinterpret_cast<float4v*>(pq) = float4v{r2[0], r2[1], r2[2], r2[3]};
        *reinterpret_cast<float4v*>(pq + 4) = float4v{r2[4], r2[5], r2[6], r2[7]};
    }
}

// The fused form's route for the same partial rows (round 6): the reduced sums of a stage go to a small LDS buffer (groups of four stages, two buffers) and a
// finished group leaves as ONE 16-byte store per thread at the top of a stage, in FRONT of that stage's transfers.  Stored straight from the reduction -- four
// stores per wave in the middle of every stage, two lanes each -- they were the youngest operations in front of the stage's closing s_waitcnt vmcnt(0): every one
// of the 48 stages waited for a store round trip (timing probes r15bh: 147-152 us with the stores, 136 without, 112 without statistics).  Same values, same
// addresses: the finish kernel and the stand-alone statistics kernel see no difference.
constexpr int CV_IN_STATS_GROUP = 4;                                             // stages per flush
constexpr int CV_IN_STATS_BUF_BYTES = 2 * 2 * CV_IN_STATS_GROUP * CV_BK * 4;     // [buffer][sum | sq][stage of the group][64 channels] fp32 = 4 KiB
__device__ __forceinline__ void input_stats_reduce_to_lds(const InStats& st, float* sbuf, int s, int wave, int lane) {
    const int chunk = 2 * wave + (lane >> 5), g = lane & 31;
    float r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // (the reduction of input_stats_reduce_store)
        float t1 = st.s1[e >> 1][e & 1], t2 = st.s2[e >> 1][e & 1];
        MV_OPAQUE(t1);
        MV_OPAQUE(t2);
        t1 = row16_sum(t1);
        t2 = row16_sum(t2);
        t1 += dpp_mov<DPP_ROW_BCAST15>(0.0f, t1);
        t2 += dpp_mov<DPP_ROW_BCAST15>(0.0f, t2);
        MV_OPAQUE(t1);
        MV_OPAQUE(t2);
        r1[e] = t1;
        r2[e] = t2;
    }
    if (g == 16) {
        float* b = sbuf + ((s / CV_IN_STATS_GROUP) & 1) * (2 * CV_IN_STATS_GROUP * CV_BK) + (s % CV_IN_STATS_GROUP) * CV_BK + chunk * 8;
        *reinterpret_cast<float4v*>(b) = float4v{r1[0], r1[1], r1[2], r1[3]};
        *reinterpret_cast<float4v*>(b + 4) = float4v{r1[4], r1[5], r1[6], r1[7]};
        *reinterpret_cast<float4v*>(b + CV_IN_STATS_GROUP * CV_BK) = float4v{r2[0], r2[1], r2[2], r2[3]};
        *reinterpret_cast<float4v*>(b + CV_IN_STATS_GROUP * CV_BK + 4) = float4v{r2[4], r2[5], r2[6], r2[7]};
    }
}
// the group whose last stage is s_last (a barrier lies between its last reduction and this call): threads 0 .. 127, 16 bytes each
__device__ __forceinline__ void input_stats_flush(const ConvArgs& a, const float* sbuf, int s_last, int n_tile, int tid) {
    if (tid >= 2 * CV_IN_STATS_GROUP * CV_BK / 4) return;
    const int grp = s_last / CV_IN_STATS_GROUP;
    const int arr = tid / (CV_IN_STATS_GROUP * CV_BK / 4), idx = tid - arr * (CV_IN_STATS_GROUP * CV_BK / 4);   // idx: float4 inside [stage][64 channels]
    const int sl = idx / (CV_BK / 4), c4 = (idx - sl * (CV_BK / 4)) * 4;
    const int s = grp * CV_IN_STATS_GROUP + sl, c = s * CV_BK + c4;
    if (s > s_last || c >= a.cin) return;   // (cin % 8 == 0: a group of four channels is inside or outside)
    const float4v v = *reinterpret_cast<const float4v*>(sbuf + (grp & 1) * (2 * CV_IN_STATS_GROUP * CV_BK) + arr * (CV_IN_STATS_GROUP * CV_BK) + sl * CV_BK + c4);
    float* dst = (arr ? a.in_sq : a.in_sum) + (int64_t)n_tile * a.cin + c;
    *reinterpret_cast<float4v*>(dst) = v;
}

// ---- fast path: fp16 input, no input transform: global -> LDS directly (global_load_lds), no register staging ----
// Workgroup tile = (WC*MI*16) output channels x (WN*NI*16) time steps, WC x WN waves.  Two instances are built:
//   <2,2,4,4>  128 x 128, 4 waves, 64 KiB LDS (2 workgroups per CU)  -- narrow layers
//   <2,2,2,2>  64 x 64, 4 waves, 32 KiB LDS                           -- small problems (a few utterances): four times the workgroups
//   <2,4,8,4>  256 x 256, 8 waves, 128 KiB LDS (1 workgroup per CU)  -- wide layers: each wave owns 128 x 64, i.e. 12
//              fragment reads per 32 MFMAs instead of 8 per 16, and half the global->LDS bytes per FLOP; the 128^2
//              kernel is LDS-bandwidth bound (reads + DMA writes ~1200 LDS cycles vs 1024 MFMA cycles per K stage).
// INSTATS: the fused input statistics as a compile-time variant, so that their VALU work sits in the K loop's basic block next to
// the MFMAs (the scheduler interleaves them; behind a run-time branch they would run after the matrix pipe's 40 issues)
// NS > 2 (the 64 x 64 instantiation, round 5): a ring of NS stages with NS - 1 in flight, untracked transfers, counted waits and one LDS-only barrier per
// stage -- with one stage of prefetch behind `s_waitcnt vmcnt(0)` a stage of these small tiles (16 MFMAs per wave) costs one memory round trip: one 3 s
// utterance's dense layers 16-48 serial round trips each (r12h: 9 convs = 149 of 459 us).  Same accumulation order: the same bits.
template <int WC, int WN, int MI, int NI, bool INSTATS = false, int NS = 2>
__global__ __launch_bounds__(64 * WC * WN) void conv1d_glds_kernel(ConvArgs a) {
    constexpr int TC = WC * MI * 16, TN = WN * NI * 16, NW = WC * WN;
    constexpr int NTW = TC / 8 / NW, NTX = TN / 8 / NW;  // 1 KiB transfers per wave per stage
    constexpr int STAGE_BYTES = (TC + TN) * CV_BK * 2;
    MV_DYN_SMEM(smem);
    int n_tile, co_tile;
    if (!tile_of_block(a, n_tile, co_tile)) return;  // whole workgroup leaves before any barrier
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS destinations of the transfers are SGPR math
    const int wc = wave / WN, wn = wave % WN;
    // rows of the tile: n0 .. n0 + TN - 1 of the flat [B * T_out] row space, valid below n_end.  per_utt: tile = (utterance, block of
    // TN frames), so no tile straddles two utterances (the fused input statistics are per utterance)
    int n0 = n_tile * TN, n_end = a.n_rows;
    if (a.per_utt) {
        const int ub = n_tile / a.tiles_per_utt;
        n0 = ub * a.T_out + (n_tile - ub * a.tiles_per_utt) * TN;
        n_end = (ub + 1) * a.T_out;
    }
    const int co0 = co_tile * TC;

    // Wave w issues NTX (NTW) transfers of 8 rows x 128 B for the activation (weight) tile: transfer i covers rows
    // (w*NT+i)*8 .. +8.  Lane l lands at LDS position (row = l>>3, slot = l&7) and fetches source chunk slot ^ (row & 7).
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ (lrow & 7);  // (row & 7) == lrow because transfers start at multiples of 8
    RowMap rm[NTX];
    const half_t* wsrc[NTW];
#pragma unroll
    for (int i = 0; i < NTX; ++i) {
        const int n = n0 + (wave * NTX + i) * 8 + lrow;
        if (n < n_end) {
            rm[i].b = n / a.T_out;
            rm[i].t = n - rm[i].b * a.T_out;
        } else {
            rm[i].b = -1;
            rm[i].t = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int co = co0 + (wave * NTW + i) * 8 + lrow;
        wsrc[i] = co < a.cout_pad ? a.w + (int64_t)co * a.k * a.cin_pad + kc * 8 : nullptr;
    }
    const half_t* xbase = reinterpret_cast<const half_t*>(a.x);
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page);
    const int kstages_per_tap = a.cin_pad / CV_BK;
    const int nstages = a.k * kstages_per_tap;

    // Row pointers of this lane's activation transfers for the current tap (zero page for padded / missing rows), rebuilt
    // when the tap changes: a stage then costs a select and a 64-bit add per transfer instead of the full index arithmetic.
    const half_t* xrow[NTX];
    bool xok[NTX];
    int cur_tap = -1;
    const bool full_k = (a.cin % CV_BK) == 0;  // no partial channel block at the end of a tap
    auto issue = [&](int s, int buf) {
        char* wt = smem + buf * STAGE_BYTES;
        char* xtile = wt + TC * CV_BK * 2;
        const int tap = s / kstages_per_tap;
        const int c0 = (s - tap * kstages_per_tap) * CV_BK;
        if (tap != cur_tap) {  // uniform
            cur_tap = tap;
#pragma unroll
            for (int i = 0; i < NTX; ++i) {
                const int tin = input_time(a, rm[i].t, tap);
                xok[i] = rm[i].b >= 0 && tin >= 0;
                xrow[i] = xok[i] ? xbase + ((int64_t)rm[i].b * a.T_in + tin) * a.ldx + kc * 8 : zero;
            }
        }
        const bool ch_ok = full_k || c0 + kc * 8 < a.cin;
#pragma unroll
        for (int i = 0; i < NTX; ++i) glds16((xok[i] && ch_ok) ? xrow[i] + c0 : zero, xtile + (wave * NTX + i) * 1024);
        const int64_t woff = (int64_t)tap * a.cin_pad + c0;
#pragma unroll
        for (int i = 0; i < NTW; ++i) glds16(wsrc[i] != nullptr ? wsrc[i] + woff : zero, wt + (wave * NTW + i) * 1024);
    };

    float4v acc[MI][NI];
    conv_acc_init<MI, NI>(a, co0, wc, lane, acc);

    if constexpr (NS > 2) {
        // (The fused input statistics as such a ring -- the 128 x 160 tile with three stages in flight, 144 KiB, ONE workgroup per CU -- measured 197-212 us
        //  against 150-160 us for the double buffer with two workgroups per CU on the ASP hidden layer, r15b: four waves per CU cannot overlap a stage's
        //  MFMA / statistics chain with another workgroup's; parked as tools/variants/conv1d_in_stats_ring.patch.txt.)
        static_assert(!INSTATS && NS <= 4, "the ring form has no fused statistics; its waits are written out for up to three stages in flight");
        constexpr int TPS = NTX + NTW;   // transfers per wave and stage: always all of them (zero page for what is missing), so the waits can be counted
        const unsigned smem_addr = lds_addr(smem);
        auto issue_ring = [&](int s) {
            const unsigned wt = smem_addr + (unsigned)((s % NS) * STAGE_BYTES);
            const unsigned xtile = wt + TC * CV_BK * 2;
            const int tap = s / kstages_per_tap;
            const int c0 = (s - tap * kstages_per_tap) * CV_BK;
            if (tap != cur_tap) {  // uniform
                cur_tap = tap;
#pragma unroll
                for (int i = 0; i < NTX; ++i) {
                    const int tin = input_time(a, rm[i].t, tap);
                    xok[i] = rm[i].b >= 0 && tin >= 0;
                    xrow[i] = xok[i] ? xbase + ((int64_t)rm[i].b * a.T_in + tin) * a.ldx + kc * 8 : zero;
                }
            }
            const bool ch_ok = full_k || c0 + kc * 8 < a.cin;
#pragma unroll
            for (int i = 0; i < NTX; ++i) glds16_untracked((xok[i] && ch_ok) ? xrow[i] + c0 : zero, xtile + (unsigned)((wave * NTX + i) * 1024));
            const int64_t woff = (int64_t)tap * a.cin_pad + c0;
#pragma unroll
            for (int i = 0; i < NTW; ++i) glds16_untracked(wsrc[i] != nullptr ? wsrc[i] + woff : zero, wt + (unsigned)((wave * NTW + i) * 1024));
        };
        for (int s0 = 0; s0 < NS - 1 && s0 < nstages; ++s0) issue_ring(s0);
        for (int s = 0; s < nstages; ++s) {
            const int last = s + NS - 2 < nstages - 1 ? s + NS - 2 : nstages - 1;   // the last stage requested so far
            const int younger = last - s;                                          // stages requested behind stage s (uniform)
            if (younger >= 2) {
                wait_vm<2 * TPS>();
            } else if (younger == 1) {
                wait_vm<TPS>();
            } else {
                wait_vm<0>();
            }
            lds_barrier();   // stage s has landed in every wave; every wave is done with stage s - 1, whose slot is requested now
            if (s + NS - 1 < nstages) issue_ring(s + NS - 1);
            const char* wt = smem + (s % NS) * STAGE_BYTES;
            mma_stage<MI, NI>(wt, wt + TC * CV_BK * 2, wc, wn, lane, acc);
        }
        __syncthreads();   // (nothing in flight: the last stage waited for everything) the staged epilogue reuses the ring
        if (a.y_f16 && a.sum_dst == nullptr) {
            conv_epilogue_staged<MI, NI, TC, TN, 64 * NW>(a, smem, n0, n_end, co0, wc, wn, lane, tid, acc);
        } else {
            conv_epilogue<MI, NI>(a, n0, n_end, co0, wc, wn, lane, acc);
        }
        return;
    }
    issue(0, 0);
    wait_all_loads();
    __syncthreads();
    InStats st;  // (INSTATS) time sums of the stage before the current one, reduced under the current stage's MFMAs
    [[maybe_unused]] float* sbuf = reinterpret_cast<float*>(smem + NS * STAGE_BYTES);   // (INSTATS) reduced sums of up to two groups of stages
    for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        if constexpr (INSTATS) {   // stage s - 2 was reduced during stage s - 1 (a barrier ago): when it closed a group, the group leaves now (uniform)
            if (s >= 2 && (s - 1) % CV_IN_STATS_GROUP == 0) input_stats_flush(a, sbuf, s - 2, n_tile, tid);
        }
        if (s + 1 < nstages) issue(s + 1, buf ^ 1);  // lands while this stage computes
        const char* wt = smem + buf * STAGE_BYTES;
        if constexpr (INSTATS) {
            // the statistics of the stage in two halves, each behind one K half's MFMAs: the matrix pipe works off its 20 issues while
            // the vector unit runs the ~100 VALU operations written after them (a single channel tile here: cout <= 128)
            static_assert(NW == 4 && TN % 32 == 0, "input statistics: 256 threads, whole groups of 32 rows");
            mma_half_stage<MI, NI>(wt, wt + TC * CV_BK * 2, wc, wn, lane, 0, acc);
            if (s > 0) input_stats_reduce_to_lds(st, sbuf, s - 1, wave, lane);  // the previous stage's sums (uniform)
            mma_half_stage<MI, NI>(wt, wt + TC * CV_BK * 2, wc, wn, lane, 1, acc);
            input_stats_accumulate<TN>(st, wt + TC * CV_BK * 2, wave, lane);
        } else {
            mma_stage<MI, NI>(wt, wt + TC * CV_BK * 2, wc, wn, lane, acc);
        }
        wait_all_loads();
        __syncthreads();
    }
    if constexpr (INSTATS) {
        // the last stage's sums, then what the buffer still holds: the group of stage nstages - 2 when that stage closed one (its flush would have come at the
        // top of a stage nstages), and the group of the last stage
        input_stats_reduce_to_lds(st, sbuf, nstages - 1, wave, lane);
        __syncthreads();
        if (nstages >= 2 && (nstages - 1) % CV_IN_STATS_GROUP == 0) input_stats_flush(a, sbuf, nstages - 2, n_tile, tid);
        input_stats_flush(a, sbuf, nstages - 1, n_tile, tid);
    }
    if (a.y_f16 && a.sum_dst == nullptr) {
        conv_epilogue_staged<MI, NI, TC, TN, 64 * NW>(a, smem, n0, n_end, co0, wc, wn, lane, tid, acc);
    } else {
        conv_epilogue<MI, NI>(a, n0, n_end, co0, wc, wn, lane, acc);
    }
}

// ---- the fused input statistics WITHOUT the GEMM (small batches) --------------------------------------------------------------------------
// With a handful of utterances the INSTATS kernel above is two workgroups per utterance walking 48 serial K stages (one 3 s utterance through
// the ASP hidden layer: 84-93 us).  Small batches therefore split the work: the conv runs on 64 x 64 tiles (no statistics), and this kernel
// produces the partial rows -- one workgroup per (utterance tile of CV_IN_STATS_TN frames, K stage of 64 channels), the SAME LDS image of the
// x tile and the SAME input_stats_accumulate / input_stats_reduce_store, so the partial rows carry the bits of the fused form (a row's
// embedding does not depend on the batch it sits in).  1x1, stride 1, no padding (the fused form's own requirements).
constexpr int CV_IN_STATS_TN_K = 160;
__global__ __launch_bounds__(CV_THREADS) void conv1d_in_stats_kernel(ConvArgs a) {
    constexpr int TN = CV_IN_STATS_TN_K, NW = CV_THREADS / 64, NTX = TN / 8 / NW;
    MV_DYN_SMEM(smem);
    const int n_tile = blockIdx.x, s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ub = n_tile / a.tiles_per_utt;
    const int n0 = ub * a.T_out + (n_tile - ub * a.tiles_per_utt) * TN, n_end = (ub + 1) * a.T_out;
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ (lrow & 7);
    const int c0 = s * CV_BK;
    const half_t* xbase = reinterpret_cast<const half_t*>(a.x);
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page);
    const bool ch_ok = c0 + kc * 8 < a.cin;
#pragma unroll
    for (int i = 0; i < NTX; ++i) {
        const int n = n0 + (wave * NTX + i) * 8 + lrow;
        const half_t* src = zero;
        if (n < n_end && ch_ok) {
            const int b = n / a.T_out, t = n - b * a.T_out;
            src = xbase + ((int64_t)b * a.T_in + t) * a.ldx + kc * 8 + c0;
        }
        glds16(src, smem + (wave * NTX + i) * 1024);
    }
    wait_all_loads();
    __syncthreads();
    InStats st;
    input_stats_accumulate<TN>(st, smem, wave, lane);
    input_stats_reduce_store(a, st, c0, n_tile, wave, lane);
}

// ---- persistent 256 x 256 kernel: one workgroup per CU walks a list of tiles ------------------------------------
// The 256^2 kernel above runs one workgroup per CU (128 KiB of LDS), so nothing overlaps a tile's prologue (first
// stage in flight, nothing to compute) and epilogue (128 KiB of stores, all CUs at once, memory pipes idle during the
// K loops): ~20 us per tile, a third of the time of a K = 1024 layer.  Here the workgroup stays resident and walks the
// tiles  blockIdx.x + i * gridDim.x  of the same XCD-aware order:
//   * the first K stage of the next tile is requested while the last stage of the current one computes;
//   * the epilogue of tile i runs inside stage 0 of tile i+1 (after that stage's loads have been requested, before its
//     MFMAs overwrite the accumulators), so its stores drain under the next tile's K loop;
//   * bias / scale / shift of a tile arrive by LDS-DMA with its first stage (three rotating 3 KiB slots), so the
//     epilogue issues no vector loads that would have to queue behind the stage loads;
//   * the output leaves straight from the accumulators as 16-byte stores (v_permlane16_swap pairs up adjacent channel
//     tiles so that a lane owns 8 consecutive channels); no LDS staging, no barrier in the epilogue.
constexpr int CVP_DMA_STEPS = 4;   // over how many of a stage's 8 MFMA steps its 8 transfers are spread (8 measured slower: r02d)
constexpr int CVP_STAGE_BYTES = 65536;
constexpr int CVP_PARAM_OFF = 2 * CVP_STAGE_BYTES;
constexpr int CVP_PARAM_SLOT = 3 * 1024;
constexpr int CVP_LDS_BYTES = CVP_PARAM_OFF + 3 * CVP_PARAM_SLOT;  // 140 288 B of the 160 KiB

// Epilogue of the persistent kernel, straight from the accumulators to 16-byte stores -- no LDS staging, no fences.
// A lane holds channels 4q..4q+3 (q = lane >> 4) of one time step r (= lane & 15) for each of its 8 channel tiles.  For a
// pair of adjacent tiles (A, B) two v_permlane16_swap per register pair hand every even 16-lane row the second half of
// tile A's channels and every odd row the first half of tile B's, so each lane ends up with 8 CONSECUTIVE channels:
//   rows q even: A: 4q .. 4q+7           rows q odd: B: 4(q-1) .. 4(q-1)+7
// and one store instruction writes, per time step, the 64 contiguous bytes of channels A*16 .. A*16+31.
// STATS (0 none, 1 sums, 2 sums + sums of squares): the per-utterance time statistics of the output -- the SE squeeze
// (ecapa_tdnn.py:79) and the ASP global mean / std (pooling.py:104-109) -- are taken from the accumulators instead of a
// second pass over the stored tensor.  Moments are about the BatchNorm shift t[c] (u = y - t; for conv -> ReLU -> BN that
// is relu(.) * scale: a dead channel gives exact zeros).  A wave's 64 time steps are reduced with DPP row sums and written
// as one partial row per (64-row block, slot): slot 0 = steps of the utterance the block starts in, slot 1 = steps of the
// next utterance when its first frame lies inside the block (T_out >= 64, so at most one boundary).  stats_finish_kernel
// adds up the 5-6 partial rows of an utterance in a fixed order (no atomics: results are run-to-run identical).
// Where the epilogue's per-channel parameters (bias / BatchNorm scale / shift of the wave's 128 channels) come from: the lane asks
// for the four channels  mi * 16 + 4 * (lane >> 4) ..+3  of channel tile mi (0..7).
struct LdsParams {  // double-buffer kernel: a 3 KiB slot of LDS filled by LDS-DMA with the tile's first stage
    const char* par;
    int wc, q;
    __device__ __forceinline__ float4v at(int array, int mi) const {
        return *reinterpret_cast<const float4v*>(par + array * 1024 + (wc * 128 + mi * 16 + 4 * q) * 4);
    }
    __device__ __forceinline__ float4v bias4(int mi) const { return at(0, mi); }
    __device__ __forceinline__ float4v scale4(int mi) const { return at(1, mi); }
    __device__ __forceinline__ float4v shift4(int mi) const { return at(2, mi); }
};
struct LaneParams {  // ring kernel (no LDS left): lane L of the wave holds channel  half * 64 + L  of the wave's 128 in six registers;
    float b[2], s[2], t[2];  // a lane fetches what it needs from the holder lanes through the LDS crossbar (ds_bpermute)
    int qaddr;               // 16 * (lane >> 4): byte index of holder lane 4 * (lane >> 4)
    __device__ __forceinline__ float4v get(const float (&h)[2], int mi) const {
        const float src = h[mi >> 2];
        float4v r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = lane_gather(src, qaddr + (mi & 3) * 64 + 4 * j);
        return r;
    }
    __device__ __forceinline__ float4v bias4(int mi) const { return get(b, mi); }
    __device__ __forceinline__ float4v scale4(int mi) const { return get(s, mi); }
    __device__ __forceinline__ float4v shift4(int mi) const { return get(t, mi); }
};

// HWSAT (ring kernel; STATS == 0, no post-activation -- the launcher sends everything else to the double-buffer kernel): the fp16 saturation is the
// wave's MODE.FP16_OVFL (arch/gfx950.h: fp16_saturation_on / half_hwsat), not a v_med3 per value: 2.5 instead of 3.5 vector issue slots per value.
template <int STATS, bool HWSAT = false, class P>
__device__ __forceinline__ void persistent_epilogue(const ConvArgs& a, const P& par, int n0, int co0, int wc, int wn, int lane,
                                                    float4v (&acc)[8][4]) {
    static_assert(!HWSAT || STATS == 0, "the hardware-saturated form is the plain epilogue");
    const int r = lane & 15, q = lane >> 4;
    half_t* y = reinterpret_cast<half_t*>(a.y);
    // host admits none / ReLU only: max(v, -inf) is the identity
    const float lo_pre = a.pre_act == MV_ACT_RELU ? 0.0f : -INFINITY, lo_post = a.post_act == MV_ACT_RELU ? 0.0f : -65504.0f;
    const int ch_lane = (q & 1) * 16 + 4 * (q & ~1);  // first channel of this lane inside the 32-channel pair
    // statistics: where the next utterance starts, relative to this wave's first time step
    const int first = n0 + wn * 64;
    const int split = STATS ? (first / a.T_out + 1) * a.T_out - first : 0;  // rows >= split belong to the next utterance
    if constexpr (STATS == 0) {
        // Two halves of 4 channel tiles: their scale / shift stay in registers across the four 16-step column blocks (the
        // per-block broadcast reads of the parameters used to cost more LDS cycles than the whole K stage), the bias is
        // already in the accumulators (see the init in stage 0), and the two 64-byte halves of a 128-byte output line are
        // still written by consecutive stores.
        // The row check of a store is per WAVE where it can be (round 6): all of the tensor's row tiles but the last hold valid rows only, and a wave whose 64
        // time steps are all inside the tensor writes its sixteen stores as straight-line code -- behind a per-lane condition every store sat in its own
        // skipped block (v_cmp, s_and_saveexec, s_cbranch_execz, s_or per store, and nothing scheduled across the block boundaries).
        const bool wave_full = MV_UNIFORM(n0 + wn * 64 + 64 <= a.n_rows);
        auto halves = [&](auto FULL) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4v sc[4], sh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sc[u] = par.scale4(4 * h + u);
                sh[u] = par.shift4(4 * h + u);
            }
            auto finish = [&](const float4v& c, int u) {
                // pre-activation as one v_max each, BatchNorm affine as two v_pk_fma_f32, post-activation + fp16 saturation
                // as one v_med3 each
                float2v v0 = {max_raw(c[0], lo_pre), max_raw(c[1], lo_pre)};
                float2v v1 = {max_raw(c[2], lo_pre), max_raw(c[3], lo_pre)};
                v0 = v0 * float2v{sc[u][0], sc[u][1]} + float2v{sh[u][0], sh[u][1]};
                v1 = v1 * float2v{sc[u][2], sc[u][3]} + float2v{sh[u][2], sh[u][3]};
                half4v hv;
                if constexpr (HWSAT) {
                    hv[0] = half_hwsat(v0[0]);
                    hv[1] = half_hwsat(v0[1]);
                    hv[2] = half_hwsat(v1[0]);
                    hv[3] = half_hwsat(v1[1]);
                } else {
                    hv[0] = (half_t)clamp3(v0[0], lo_post, 65504.0f);
                    hv[1] = (half_t)clamp3(v0[1], lo_post, 65504.0f);
                    hv[2] = (half_t)clamp3(v1[0], lo_post, 65504.0f);
                    hv[3] = (half_t)clamp3(v1[1], lo_post, 65504.0f);
                }
                return hv;
            };
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + r;
                half_t* yrow = y + (int64_t)n * a.ldy + co0 + wc * 128 + ch_lane;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int p = 2 * h + pp;
                    const half4v ha = finish(acc[2 * p][ni], 2 * pp), hb = finish(acc[2 * p + 1][ni], 2 * pp + 1);
                    unsigned xa[2], xb[2];
                    __builtin_memcpy(xa, &ha, 8);
                    __builtin_memcpy(xb, &hb, 8);
                    row_swap_odd_even(xa[0], xb[0]);
                    row_swap_odd_even(xa[1], xb[1]);
                    const unsigned o[4] = {xa[0], xa[1], xb[0], xb[1]};
                    half8v ov;
                    __builtin_memcpy(&ov, o, 16);
                    if (decltype(FULL)::value || n < a.n_rows) *reinterpret_cast<half8v*>(yrow + p * 32) = ov;
                }
            }
        }
        };
        // (ring kernel only: in the double-buffer kernel -- LDS parameters, 242 registers -- the second copy of the epilogue made the register allocator spill,
        // 256 VGPRs + 48 bytes of scratch, and block 0 of the backbone went from 88-92 to 101-104 us: r15bk)
        if (HWSAT && wave_full) {
            halves(std::true_type{});
        } else {
            halves(std::false_type{});
        }
        return;
    }
    const bool straddles = STATS && split < 64;
    float4v keep_sum[2], keep_sq[2];  // [slot]: this lane's share of the wave's partial row (channel tile r, channels 4q..4q+3)
    keep_sum[0] = keep_sum[1] = keep_sq[0] = keep_sq[1] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float4v shift4[2], scale4[2];
        float4v s_sum[2][2], s_sq[2][2];  // [slot][tile of the pair]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            shift4[u] = par.shift4(2 * p + u);
            scale4[u] = par.scale4(2 * p + u);
            s_sum[0][u] = s_sum[1][u] = s_sq[0][u] = s_sq[1][u] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + r;
            half4v hv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4v v = acc[2 * p + u][ni];  // bias included (accumulator init)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo_pre);
                v = v * scale4[u];  // u = y - shift
                if (STATS) {
                    float4v w = v;
                    if (a.post_act == MV_ACT_RELU) {  // the moments are those of the stored value
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = fmaxf(v[e] + shift4[u][e], 0.0f) - shift4[u][e];
                    }
                    if (straddles) {
                        const bool late = ni * 16 + r >= split;
                        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
                        const float4v w0 = late ? zero4 : w, w1 = late ? w : zero4;
                        s_sum[0][u] += w0;
                        s_sum[1][u] += w1;
                        if (STATS == 2) {
                            s_sq[0][u] += w0 * w0;
                            s_sq[1][u] += w1 * w1;
                        }
                    } else {
                        s_sum[0][u] += w;
                        if (STATS == 2) s_sq[0][u] += w * w;
                    }
                }
                v += shift4[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[u][e] = (half_t)clamp3(v[e], lo_post, 65504.0f);  // post-activation + fp16 saturation: one v_med3
            }
            unsigned xa[2], xb[2];
            __builtin_memcpy(xa, &hv[0], 8);
            __builtin_memcpy(xb, &hv[1], 8);
            row_swap_odd_even(xa[0], xb[0]);
            row_swap_odd_even(xa[1], xb[1]);
            const unsigned o[4] = {xa[0], xa[1], xb[0], xb[1]};
            half8v ov;
            __builtin_memcpy(&ov, o, 16);
            if (n < a.n_rows) *reinterpret_cast<half8v*>(y + (int64_t)n * a.ldy + co0 + wc * 128 + ch_lane + p * 32) = ov;
        }
        if (STATS) {
            // DPP row sums over the 16 time steps of a tile row group: every lane of a row then holds the sums of its 4
            // channels; lane r keeps those of channel tile r, so the wave's 128 channels leave in ONE 512-byte store per
            // (slot, moment) at the end instead of one 4-lane store per tile (store issue is what the epilogue waits on)
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                if (slot == 1 && !straddles) break;  // uniform
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float4v t = s_sum[slot][u], t2 = s_sq[slot][u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = row16_sum(t[e]);
                        if (STATS == 2) t2[e] = row16_sum(t2[e]);
                    }
                    const bool mine = r == 2 * p + u;
                    keep_sum[slot] = mine ? t : keep_sum[slot];
                    if (STATS == 2) keep_sq[slot] = mine ? t2 : keep_sq[slot];
                }
            }
        }
    }
    if (STATS) {
        const int64_t prow = ((int64_t)(first >> 6) * 2) * a.cout + co0 + wc * 128 + r * 16 + 4 * q;
        if (r < 8) {
            *reinterpret_cast<float4v*>(a.stat_sum + prow) = keep_sum[0];
            if (STATS == 2) *reinterpret_cast<float4v*>(a.stat_sq + prow) = keep_sq[0];
            if (straddles) {
                *reinterpret_cast<float4v*>(a.stat_sum + prow + a.cout) = keep_sum[1];
                if (STATS == 2) *reinterpret_cast<float4v*>(a.stat_sq + prow + a.cout) = keep_sq[1];
            }
        }
    }
}

// per-utterance mean (and std) from the partial rows written by the STATS epilogue
__global__ void stats_finish_kernel(const float* psum, const float* psq, const float* shift, int B, int T, int C, float* mean,
                                    float* stdv, int64_t ld_out, float clamp_eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= C || b >= B) return;
    const int64_t r0 = (int64_t)b * T, r1 = r0 + T - 1;
    float s = 0.0f, q2 = 0.0f;
    for (int64_t k = r0 >> 6; k <= (r1 >> 6); ++k) {
        const int64_t owner = (k << 6) / T;             // utterance of the block's first time step
        const int slot = owner == b ? 0 : 1;            // otherwise the block starts in utterance b - 1 and b begins inside it
        s += psum[(k * 2 + slot) * C + c];
        if (psq != nullptr) q2 += psq[(k * 2 + slot) * C + c];
    }
    const float t = shift != nullptr ? shift[c] : 0.0f;
    const float m = s / (float)T;
    mean[(int64_t)b * ld_out + c] = t + m;
    if (stdv != nullptr) {
        float var = fmaxf(q2 / (float)T - m * m, 0.0f);
        if (clamp_eps > 0.0f) var = fmaxf(var, clamp_eps);
        stdv[(int64_t)b * ld_out + c] = sqrtf(var);
    }
}

int conv_stats_finish_launch(const float* psum, const float* psq, const float* shift, int B, int T, int C, float* mean, float* stdv,
                             int64_t ld_out, float clamp_eps, hipStream_t stream) {
    MV_REQUIRE(psum != nullptr && mean != nullptr && B > 0 && T >= 64 && C > 0, "conv_stats_finish: bad argument");
    MV_LAUNCH(stats_finish_kernel, ((unsigned)ceil_div(C, 256), (unsigned)B, 1), (256, 1, 1), 0, stream, psum, psq, shift, B, T, C, mean,
              stdv, ld_out, clamp_eps);
    return check_launch("stats_finish_kernel");
}

// per-utterance mean / std of a conv INPUT from the partial rows of input_stats_stage: [B * tiles_per_utt][C]
__global__ void in_stats_finish_kernel(const float* psum, const float* psq, int B, int T, int C, int tiles_per_utt, float* mean, float* stdv,
                                       int64_t ld_out, float clamp_eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= C || b >= B) return;
    // The tile partials are fp32 sums; their combination and E[x^2] - mean^2 run in double: for a channel with |mean| >> std the
    // difference cancels the leading bits, and every fp32 rounding of s / T, m * m, q / T in front of it would be amplified by
    // (mean / std)^2 on top of the partials' own rounding (the reference's two-pass form, pooling.py, has no such term)
    double s = 0.0, q2 = 0.0;
    for (int k = 0; k < tiles_per_utt; ++k) {
        s += (double)psum[((int64_t)b * tiles_per_utt + k) * C + c];
        q2 += (double)psq[((int64_t)b * tiles_per_utt + k) * C + c];
    }
    const double m = s / (double)T;
    mean[(int64_t)b * ld_out + c] = (float)m;
    if (stdv != nullptr) {
        float var = (float)fmax(q2 / (double)T - m * m, 0.0);
        if (clamp_eps > 0.0f) var = fmaxf(var, clamp_eps);
        stdv[(int64_t)b * ld_out + c] = sqrtf(var);
    }
}

constexpr int CV_IN_STATS_TN = 160;  // the tile of the kernel that takes them
static_assert(CV_IN_STATS_TN == CV_IN_STATS_TN_K, "the stand-alone statistics kernel mirrors the fused form's tile");
int64_t conv_in_stats_elems(int B, int T, int cin) { return (int64_t)B * ceil_div(T, CV_IN_STATS_TN) * cin; }

int conv_in_stats_finish_launch(const float* psum, const float* psq, int B, int T, int C, float* mean, float* stdv, int64_t ld_out,
                                float clamp_eps, hipStream_t stream) {
    MV_REQUIRE(psum != nullptr && psq != nullptr && mean != nullptr && B > 0 && T > 0 && C > 0, "conv_in_stats_finish: bad argument");
    MV_LAUNCH(in_stats_finish_kernel, ((unsigned)ceil_div(C, 256), (unsigned)B, 1), (256, 1, 1), 0, stream, psum, psq, B, T, C,
              (int)ceil_div(T, CV_IN_STATS_TN), mean, stdv, ld_out, clamp_eps);
    return check_launch("in_stats_finish_kernel");
}


template <bool SIMPLE, int STATS>
__global__ __launch_bounds__(512) void conv1d_glds_persistent_kernel(ConvArgs a) {
    constexpr int WN = 4, MI = 8, NI = 4, TC = 256, TN = 256, NTW = 4, NTX = 4;
    MV_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS destinations of the transfers are SGPR math
    const int wc = wave / WN, wn = wave % WN;
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ (lrow & 7);
    const int total = a.n_tiles * a.co_tiles;  // workgroup ids of tile_of_index
    const half_t* xbase = reinterpret_cast<const half_t*>(a.x);
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page);
    const int kstages_per_tap = a.cin_pad / CV_BK;
    const int nstages = a.k * kstages_per_tap;

    // loader state: the tile whose stages are being requested
    RowMap rm[NTX];
    const half_t* wsrc[NTW];
    int l_vb = blockIdx.x, l_n0 = 0, l_co0 = 0, l_ps = 0;
    int l_tap = -1;  // tap the activation row pointers below belong to

    auto advance = [&](int vb) {  // first valid tile at or after vb (stride gridDim.x); sets the loader state
        int n_tile = 0, co_tile = 0;
        while (vb < total && !tile_of_index(a, vb, n_tile, co_tile)) vb += gridDim.x;
        l_vb = vb;
        if (vb >= total) return false;
        l_n0 = n_tile * TN;
        l_co0 = co_tile * TC;
        l_tap = -1;  // new rows: the row pointers are rebuilt by the next issue()
#pragma unroll
        for (int i = 0; i < NTX; ++i) {
            const int n = l_n0 + (wave * NTX + i) * 8 + lrow;
            if (n < a.n_rows) {
                rm[i].b = n / a.T_out;
                rm[i].t = n - rm[i].b * a.T_out;
            } else {
                rm[i].b = -1;
                rm[i].t = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int co = l_co0 + (wave * NTW + i) * 8 + lrow;  // cout % 256 == 0: always inside the packed weights
            wsrc[i] = a.w + (int64_t)co * a.k * a.cin_pad + kc * 8;
        }
        return true;
    };

    // Row pointers of this lane's activation transfers for the loader's current tap (zero page for padded / missing
    // rows): a stage then costs one select and one 64-bit add per transfer instead of the full index arithmetic.
    const half_t* xrow[NTX];
    bool xok[NTX];
    const bool full_k = (a.cin % CV_BK) == 0;  // no partial channel block at the end of a tap
    // prepare(s, buf): source pointers of this lane's eight transfers of stage s (4 activation + 4 weight row groups);
    // dma(i) issues transfer i.  Stage s+1 is prepared at the top of stage s and its transfers are issued one per MFMA step.
    const half_t* dsrc[NTX + NTW];
    int dbuf = 0;
    auto prepare = [&](int s, int buf) {
        dbuf = buf;
        const int tap = SIMPLE ? 0 : s / kstages_per_tap;
        const int c0 = (s - tap * kstages_per_tap) * CV_BK;
        if (tap != l_tap) {  // uniform: once per tile for 1x1 convolutions
            l_tap = tap;
#pragma unroll
            for (int i = 0; i < NTX; ++i) {
                const int tin = input_time(a, rm[i].t, tap);
                xok[i] = rm[i].b >= 0 && tin >= 0;
                xrow[i] = xok[i] ? xbase + ((int64_t)rm[i].b * a.T_in + tin) * a.ldx + kc * 8 : zero;
            }
        }
        if (SIMPLE || full_k) {
#pragma unroll
            for (int i = 0; i < NTX; ++i) dsrc[i] = xrow[i] + (xok[i] ? c0 : 0);
        } else {
            const bool ch_ok = c0 + kc * 8 < a.cin;
#pragma unroll
            for (int i = 0; i < NTX; ++i) {   // (a select between INTEGERS: as a select between pointers it became a two-entry table in scratch memory -- loaded,
                //  with an s_waitcnt vmcnt(0), at every tap: 25 scratch instructions in the kernel, tools/isa_audit.py)
                const uintptr_t px = reinterpret_cast<uintptr_t>(xrow[i] + c0), pz = reinterpret_cast<uintptr_t>(zero);
                dsrc[i] = reinterpret_cast<const half_t*>((xok[i] && ch_ok) ? px : pz);
            }
        }
        const int64_t woff = SIMPLE ? (int64_t)c0 : (int64_t)tap * a.cin_pad + c0;
#pragma unroll
        for (int i = 0; i < NTW; ++i) dsrc[NTX + i] = wsrc[i] + woff;
    };
    auto dma = [&](int i) {  // i < NTX: activation rows, else weight rows
        char* wt = smem + dbuf * CVP_STAGE_BYTES;
        char* dst = i < NTX ? wt + TC * CV_BK * 2 + (wave * NTX + i) * 1024 : wt + (wave * NTW + (i - NTX)) * 1024;
        glds16(dsrc[i], dst);
    };
    auto issue = [&](int s, int buf) {
        prepare(s, buf);
#pragma unroll
        for (int i = 0; i < NTX + NTW; ++i) dma(i);
    };

    // waves 0..2 fetch bias / scale / shift of the loader's tile (256 floats = 1 KiB each); absent arrays are replaced
    // by constant pages, so the epilogue is branch-free
    const float* pparam = a.bias != nullptr ? a.bias : reinterpret_cast<const float*>(g_zero_page);
    bool pconst = a.bias == nullptr;
    if (wave == 1) {
        pparam = a.scale != nullptr ? a.scale : g_one_page;
        pconst = a.scale == nullptr;
    } else if (wave == 2) {
        pparam = a.shift != nullptr ? a.shift : reinterpret_cast<const float*>(g_zero_page);
        pconst = a.shift == nullptr;
    }
    auto issue_params = [&]() {
        if (wave < 3) {
            // constant pages: g_zero_page is 256 B (every lane reads its first 16 B), g_one_page a full 1 KiB
            const float* src = pconst ? (wave == 1 ? pparam + lane * 4 : pparam) : pparam + l_co0 + lane * 4;
            glds16(src, smem + CVP_PARAM_OFF + l_ps * CVP_PARAM_SLOT + wave * 1024);
        }
    };

    if (!advance(l_vb)) return;  // whole workgroup leaves before any barrier
    int buf = 0;
    issue(0, 0);
    issue_params();

    float4v acc[MI][NI];
    bool pending = false, more = true;
    int e_n0 = 0, e_co0 = 0, e_ps = 0;
    const unsigned smem_base = lds_addr(smem);  // LDS byte address
    // One K stage.  FIRST / LAST are compile-time for the peeled copies, so the steady-state body (neither) is: wait,
    // barrier, request stage s+1, 64 MFMAs -- no tile bookkeeping, no branches.
    auto stage = [&](int s, auto first, auto last) {
        bool feed = true;  // a stage (of this tile or the first of the next one) is requested during this stage
        // bookkeeping of the stage to request (for the last stage: the next tile's row / weight pointers, two integer
        // divisions per row) runs while this stage's transfers are still landing
        if (!decltype(last)::value) {
            prepare(s + 1, buf ^ 1);
        } else {
            more = advance(l_vb + gridDim.x);
            feed = more;
            if (more) prepare(0, buf ^ 1);
        }
        wait_all_loads();
        __syncthreads();  // stage s has landed in `buf`; every wave is done with the other buffer
        if (decltype(last)::value && more) {
            l_ps = l_ps == 2 ? 0 : l_ps + 1;
            issue_params();
        }
        if (decltype(first)::value) {
            if (pending) persistent_epilogue<STATS>(a, LdsParams{smem + CVP_PARAM_OFF + e_ps * CVP_PARAM_SLOT, wc, lane >> 4}, e_n0, e_co0, wc, wn, lane, acc);
            // accumulators start from the bias of their 4 channels (parameter slot of the tile being computed)
            const LdsParams par{smem + CVP_PARAM_OFF + l_ps * CVP_PARAM_SLOT, wc, lane >> 4};
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float4v b4 = par.bias4(mi);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = b4;
            }
        }
        const unsigned wt = smem_base + buf * CVP_STAGE_BYTES;
        // CVP_DMA_STEPS = over how many of the 8 MFMA steps the 8 transfers are spread (8: one per step; 4: two per step
        // in the first half, so the youngest transfer has half a stage more to land)
        if (decltype(last)::value) {
            mma_stage_8x4(wt, wt + TC * CV_BK * 2, wc, wn, lane, acc, [&](int i) {
                if (feed && i < CVP_DMA_STEPS)
                    for (int u = 0; u < 8 / CVP_DMA_STEPS; ++u) dma(i * (8 / CVP_DMA_STEPS) + u);
            });
        } else {
            mma_stage_8x4(wt, wt + TC * CV_BK * 2, wc, wn, lane, acc, [&](int i) {
                if (i < CVP_DMA_STEPS) {
#pragma unroll
                    for (int u = 0; u < 8 / CVP_DMA_STEPS; ++u) dma(i * (8 / CVP_DMA_STEPS) + u);
                }
            });
        }
        buf ^= 1;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    while (more) {
        const int c_n0 = l_n0, c_co0 = l_co0, c_ps = l_ps;
        stage(0, yes{}, no{});  // nstages >= 2 (the launcher keeps single-stage problems on the one-shot kernel)
        for (int s = 1; s + 1 < nstages; ++s) stage(s, no{}, no{});
        stage(nstages - 1, no{}, yes{});
        e_n0 = c_n0;
        e_co0 = c_co0;
        e_ps = c_ps;
        pending = true;
    }
    persistent_epilogue<STATS>(a, LdsParams{smem + CVP_PARAM_OFF + e_ps * CVP_PARAM_SLOT, wc, lane >> 4}, e_n0, e_co0, wc, wn, lane, acc);
}


// Measured dead ends of the persistent kernel (r01s..r01v logs under profiles/): touching the lines of stage s+3 with one
// 4-byte load each to pull them into L2 early (slower: the touches sit in the same in-order vmcnt queue; repeated in round 2
// with 4-byte LDS-DMA touches into a scratch row, all transfers untracked and COUNTED stage waits that leave the touches in
// flight: still 7-12 % slower at every distance 1..4, profiles/r04h_* -- the global -> LDS path is bound by request
// throughput, not by the latency of the single stage in flight, and the touches double its requests); a late start, by a hashed
// fraction of a tile time, for the workgroups that walk one tile fewer than the longest walk, so that the 32 MiB output bursts of
// the lockstep tile boundaries spread out (neutral to 3 % slower, profiles/r04j_*); a 4-wave layout
// with 128 x 128 wave tiles and the accumulators in AccVGPRs (a third less LDS traffic, but one wave per SIMD: 7 % slower on
// K = 3072, 24 % on K = 1024); a counted vmcnt wait that lets the epilogue's stores drain across the next stage (neutral);
// starting the eight XCDs ~1 us apart to break up the 32 MiB store burst of the lockstep epilogues (slower); requesting
// stage 1 of a tile ahead of the previous tile's epilogue stores + waiting with vmcnt(16) (neutral, r02d); streaming store
// policy bits (faster per layer, slower end to end, see conv_store_policy).  Also neutral, but kept because they remove
// work: the next tile's bookkeeping ahead of the wait, epilogue parameters in registers, bias in the accumulator init,
// single-instruction max (r02d: the tile boundary is bounded by the store drain, not by the epilogue's instruction count).  The in-kernel
// timeline (MV_PROBE=3, tools/trace_conv.py) shows per K stage ~400 cycles barrier skew, ~1650 cycles MFMA issue per wave
// (two waves share a SIMD's matrix pipe: 2048 busy cycles) and 1000-2000 cycles until the next stage has landed.
//
// Round 2, bytes in flight: see conv1d_ring_persistent_kernel below (the dense 1x1 layers run there by default; this kernel keeps the taps,
// strides, padding modes and the fused statistics).
//
// Measured dead ends (kept out of the build, logs under profiles/): a 256x128 tile with 2 x 32-wide stages (335 TF,
// r01h), a 256x256 tile with a 4-slot ring of 32-wide stages and counted vmcnt (500 / 775 TF, r01i -- no better than the
// double buffer), padded leading dimensions (r01k, no effect).  Probes with the K loop reduced to its loads or to its
// LDS reads + MFMAs (MV_PROBE, r01j) show the 256x256 kernel is bound by the global->LDS path: loads alone take 83 % of
// the full time (~8.4 TB/s of L2->LDS traffic chip-wide), LDS reads + MFMAs alone reach 1530 TF.

// ---- persistent 256 x 256 kernel, asymmetric LDS ring (round 2) ---------------------------------------------------------------
// The K loop's transfers ALONE scale with the bytes a CU has in flight on the global -> LDS path (tools/gemm_load_depth_probe.hip,
// profiles/r05a): one 64 KiB stage in flight 82 us on a K = 1024 layer / 1080 us on the K = 3072 layer (15.2 / 10.4 TB/s into LDS), 96 KiB
// 71 / 809 us, 128 KiB 55 / 785 us (saturated); 32-wide stages (64-byte rows) stay at 14 / 11 TB/s however many are in flight -- the path
// counts requests, so halving the row segment halves the rate (why the 4-slot ring of 32-wide stages of round 1 gained nothing).
// A third 64 KiB stage does not fit the 160 KiB of LDS, a third HALF stage does: the weight half of a stage (L2 / MALL resident, short
// latency) is requested one stage ahead into a ring of two 32 KiB slots, the activation half (HBM on first touch) two stages ahead into a
// ring of three -- 5 x 32 KiB = all of the CU's LDS, 96 KiB in flight during every stage.  The vector-memory counter retires in order, so
// a stage requests its weight transfers BEFORE its activation transfers and the wait in front of the next stage is vmcnt(4): everything
// has landed except the four youngest transfers of the wave, the activation rows of stage s + 2.  All transfers are issued from inline
// assembly and every wait is counted by hand ("the compiler's wait counts").  With no LDS left, bias / scale / shift live in
// six registers per lane (LaneParams) and reach the lanes that need them through the LDS crossbar; a workgroup's channel tile changes at
// most once per launch in the XCD-aware walk (the walk stride is a multiple of the group width), so the holder registers are reloaded --
// with a full drain -- only then.
// History (profiles/r05b, r05l-r05o): the first form kept the generic loader (row maps, 64-bit pointers per transfer, taps) and was 3-7 %
// SLOWER than the double buffer although its transfers were never waited for: the in-kernel timeline showed ~800 cycles of loader
// bookkeeping per stage in front of the barrier (moved under the MFMAs it simply made that phase 450 cycles longer: a block of vector
// arithmetic between two MFMA steps does not overlap with them).  The form below is restricted to what the backbone's layers are --
// 1x1 convolutions over a dense row range -- where a transfer's address is (uniform tensor base + 128 bytes per stage) + a per-lane
// 32-bit row offset that is constant for the tile: the loader has no vector arithmetic inside a tile.  K = 3072: 1290 -> 1255 us in the
// micro-benchmark, K = 1024 unchanged (its tiles are dominated by the tile boundary); in the model, where the activations are cold,
// 3.66 -> 3.56 ms per step (conv class 968 -> 1008 TFLOP/s, r05o).
constexpr int CVR_SLOT_BYTES = 32768;
constexpr int CVR_X_OFF = 2 * CVR_SLOT_BYTES;
constexpr int CVR_LDS_BYTES = 5 * CVR_SLOT_BYTES;  // 163 840 B

// 1x1 convolutions over a dense row range (k = 1, stride 1, no padding, T_in == T_out, cin % 64 == 0: every TDNNBlock of the backbone
// but the first): input row = output row, so a transfer's address is  tensor + row offset (per lane, 32 bits, fixed for the tile) +
// 128 bytes per stage (scalar) -- the loader has no vector arithmetic inside a tile at all.
__global__ __launch_bounds__(512) void conv1d_ring_persistent_kernel(ConvArgs a) {
    constexpr int MI = 8, NI = 4, TC = 256, TN = 256, NTW = 4, NTX = 4;
    MV_DYN_SMEM(smem);
    fp16_saturation_on();   // the epilogue converts with half_hwsat
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = MV_UNIFORM(tid >> 6);  // scalar: LDS destinations of the transfers are SGPR math
    const int wc = wave / 4, wn = wave % 4;
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ (lrow & 7);
    // tile ids of tile_of_index -- all of them, or the whole rounds only (the launcher hands the last partial round to sub-tiles)
    const int total = a.ring_vb_end > 0 ? a.ring_vb_end : a.n_tiles * a.co_tiles;
    const int step = (int)gridDim.x;
    unsigned long long clk0 = 0, ref0 = 0;   // (scalar registers) MvConv1dDesc.clock_probe
    if (a.clock_probe != nullptr) {
        clk0 = shader_clock();
        ref0 = ref_clock_100mhz();
    }
    const char* xb = reinterpret_cast<const char*>(a.x);
    const char* wb = reinterpret_cast<const char*>(a.w);
    const unsigned xrow_bytes = (unsigned)a.ldx * 2u, wrow_bytes = (unsigned)a.cin_pad * 2u;
    const int nstages = a.cin_pad / CV_BK;
    const unsigned smem_base = lds_addr(smem);

    // three walks over the same tile list: the tile being computed, the weight stream (one stage ahead), the activation stream (two)
    auto first_tile = [&](int vb, int& n_tile, int& co_tile) {  // first valid tile at or after vb; returns total when the walk is over
        while (vb < total && !tile_of_index(a, vb, n_tile, co_tile)) vb += step;
        return vb < total ? vb : total;
    };
    // -- weight stream
    unsigned woff[NTW];
    int w_vb = 0, w_stage = 0, w_slot = 0;
    bool w_more = false;
    auto w_open = [&](int vb) {
        int n_tile = 0, co_tile = 0;
        w_vb = first_tile(vb, n_tile, co_tile);
        w_more = w_vb < total;
        w_stage = 0;
#pragma unroll
        for (int i = 0; i < NTW; ++i)  // cout % 256 == 0: always inside the packed weights
            woff[i] = (unsigned)(co_tile * TC + (wave * NTW + i) * 8 + lrow) * wrow_bytes + (unsigned)kc * 16u;
    };
    auto dma_w = [&](int i) { glds16_untracked_so(wb + w_stage * (CV_BK * 2), woff[i], smem_base + w_slot * CVR_SLOT_BYTES + (wave * NTW + i) * 1024); };
    auto w_advance = [&]() {
        w_slot ^= 1;
        if (++w_stage == nstages) w_open(w_vb + step);
    };
    // -- activation stream (rows beyond the tensor re-read its last row: their results are never stored)
    unsigned xoff[NTX];
    int x_vb = 0, x_stage = 0, x_slot = 0;
    bool x_more = false;
    auto x_open = [&](int vb) {
        int n_tile = 0, co_tile = 0;
        x_vb = first_tile(vb, n_tile, co_tile);
        x_more = x_vb < total;
        x_stage = 0;
#pragma unroll
        for (int i = 0; i < NTX; ++i) {
            int n = n_tile * TN + (wave * NTX + i) * 8 + lrow;
            n = n < a.n_rows ? n : a.n_rows - 1;
            xoff[i] = (unsigned)n * xrow_bytes + (unsigned)kc * 16u;
        }
    };
    auto dma_x = [&](int i) {
        glds16_untracked_so(xb + x_stage * (CV_BK * 2), xoff[i], smem_base + CVR_X_OFF + x_slot * CVR_SLOT_BYTES + (wave * NTX + i) * 1024);
    };
    auto x_advance = [&]() {
        x_slot = x_slot == 2 ? 0 : x_slot + 1;
        if (++x_stage == nstages) x_open(x_vb + step);
    };

    // -- compute walk
    int c_vb = 0, c_n0 = 0, c_co0 = 0;
    {
        int n_tile = 0, co_tile = 0;
        c_vb = first_tile(blockIdx.x, n_tile, co_tile);
        if (c_vb >= total) return;  // whole workgroup leaves before any barrier
        c_n0 = n_tile * TN;
        c_co0 = co_tile * TC;
    }
    // epilogue parameters of the wave's 128 channels
    LaneParams hp;
    hp.qaddr = 16 * (lane >> 4);
    int held_co0 = -1;
    auto load_params = [&](int co0) {  // plain loads + an immediate use: the compiler's wait for them (a full drain) sits right here
        const int c = co0 + wc * 128 + lane;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            hp.b[h] = a.bias != nullptr ? a.bias[c + 64 * h] : 0.0f;
            hp.s[h] = a.scale != nullptr ? a.scale[c + 64 * h] : 1.0f;
            hp.t[h] = a.shift != nullptr ? a.shift[c + 64 * h] : 0.0f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            MV_OPAQUE(hp.b[h]);
            MV_OPAQUE(hp.s[h]);
            MV_OPAQUE(hp.t[h]);
        }
        held_co0 = co0;
    };
    load_params(c_co0);

    // prologue: weights of stage 0, activations of stage 0, activations of stage 1 -- in this order (in-order retirement)
    w_open(c_vb);
    x_open(c_vb);
#pragma unroll
    for (int i = 0; i < NTW; ++i) dma_w(i);
    w_advance();
#pragma unroll
    for (int i = 0; i < NTX; ++i) dma_x(i);
    x_advance();
    bool x_ahead = x_more;  // the wave's four youngest transfers belong to a LATER stage than the one about to be computed
    if (x_more) {
#pragma unroll
        for (int i = 0; i < NTX; ++i) dma_x(i);
        x_advance();
    }

    float4v acc[MI][NI];
    bool pending = false;
    int e_n0 = 0, e_co0 = 0;
    int cw_slot = 0, cx_slot = 0;  // slots of the stage being computed
    // One K stage: counted wait, barrier, [first stage of a tile: epilogue of the previous tile, accumulator init], 64 MFMAs with the
    // weight transfers of stage s + 1 under steps 0-1, the activation transfers of stage s + 2 under steps 2-3 and the streams' advance
    // (at the end of a tile: the next tile's offsets) under steps 4-5.
    // af / bf0 / bf1: fragment registers of the K stage.  EVERY stage leaves its step-7 MFMAs undone (mma_stage_8x4_carry): a tile's later stages issue
    // the previous stage's behind their own first fragment requests; a tile's first stage issues the previous TILE's last group right behind the barrier,
    // in front of that tile's epilogue (two stage forms, as before the carry; four -- with a complete last stage -- made the register allocator spill).
    half8v af[2][2], bf0[4], bf1[4];
    auto stage = [&](auto first) {
        const bool do_w = w_more, do_x = x_more;
        if (x_ahead) {
            wait_vm<NTX>();
        } else {
            wait_vm<0>();
        }
        lds_barrier();  // this stage has landed for every wave; every wave is done with the slots requested below
        if (decltype(first)::value) {
            if (pending) {
                mfma8_step<0>(acc[6], acc[7], af[1][0], af[1][1], bf1);   // the previous tile's last MFMA group (operands in registers since the barrier)
                mfma_hazard_pad();
                persistent_epilogue<0, true>(a, hp, e_n0, e_co0, wc, wn, lane, acc);
            }
            if (c_co0 != held_co0) load_params(c_co0);  // uniform, at most once per launch on the shipped shapes
            // accumulators start from the bias of their 4 channels
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float4v b4 = hp.bias4(mi);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = b4;
            }
        }
        const unsigned wt = smem_base + cw_slot * CVR_SLOT_BYTES, xt = smem_base + CVR_X_OFF + cx_slot * CVR_SLOT_BYTES;
        mma_stage_8x4_carry<!decltype(first)::value, true>(wt, xt, wc, wn, lane, acc, af, bf0, bf1, [&](int i) {
            if (i < 2) {
                if (do_w) {
                    dma_w(2 * i);
                    dma_w(2 * i + 1);
                }
            } else if (i < 4) {
                if (do_x) {
                    dma_x(2 * (i - 2));
                    dma_x(2 * (i - 2) + 1);
                }
            } else if (i == 4) {
                if (do_w) w_advance();
            } else if (i == 5) {
                if (do_x) x_advance();
            }
        });
        x_ahead = do_x;
        cw_slot ^= 1;
        cx_slot = cx_slot == 2 ? 0 : cx_slot + 1;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    while (c_vb < total) {
        stage(yes{});
        for (int s = 1; s < nstages; ++s) stage(no{});
        e_n0 = c_n0;
        e_co0 = c_co0;
        pending = true;
        int n_tile = 0, co_tile = 0;
        c_vb = first_tile(c_vb + step, n_tile, co_tile);
        c_n0 = n_tile * TN;
        c_co0 = co_tile * TC;
    }
    lds_wait<0>(af[1][0], af[1][1]);   // the last stage's step-7 fragments
    mfma8_step<0>(acc[6], acc[7], af[1][0], af[1][1], bf1);
    mfma_hazard_pad();
    persistent_epilogue<0, true>(a, hp, e_n0, e_co0, wc, wn, lane, acc);
    if (a.clock_probe != nullptr && tid == 0) {
        unsigned long long* p = a.clock_probe + 4 * (size_t)blockIdx.x;
        p[0] = clk0;
        p[1] = shader_clock();
        p[2] = ref0;
        p[3] = ref_clock_100mhz();
    }
}

// Wave priorities (r10k): s_setprio 1 around every group of eight MFMAs, and the opposite (priority outside the groups), as probe builds of
// this file against the default in one call: 76.5 / 77.0 k (default), 76.5 / 76.7 k, 76.6 / 76.7 k utt/s -- no effect; the two waves of a SIMD
// do not compete for issue slots in a way an arbitration hint changes.
//
// Also measured on the ring kernel (r05r): the first stage of a tile requesting its transfers BEFORE the epilogue's stores, with the next
// stage's wait counted past the 16 store instructions (vmcnt(20)), so that the boundary's store burst drains under two stages instead of in
// front of stage 1: neutral (K = 1024 190.7-193.8 us vs 191-192), as the same idea had been on the double-buffer kernel -- the ~8 k cycles of
// a tile boundary are the chip absorbing 32 MiB of output at once, not an ordering artefact of the counter.
//
// Measured on top of the ring kernel (r05p, parked in tools/variants/conv1d_pingpong_persistent.hip.txt): ping-pong between the two waves of
// a SIMD -- the workgroup's halves half a K stage out of phase, a stage as four barrier-separated segments in which one half requests the
// twelve fragments of a K half while the other issues that half's 32 MFMAs from registers with the pipe to itself.  Correct (emulator + GPU
// tests) and 12 % SLOWER (K = 3072: 1252 -> 1400-1420 us, K = 1024: 190 -> 220 us): a segment is ~512 matrix-pipe cycles + ~280 of barrier
// and restart, and there are four of them per stage instead of one.

// ---- general path: fp32 or transformed input (second input added, BatchNorm+ReLU on load) through registers -------
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
// (pointer selects between a tensor and the zero page lose the address space: MV_GLOBAL_PTR states it, or the loads become FLAT loads)
__device__ __forceinline__ void load_raw(RawChunk<half_t>& r, const half_t* p) { r.v = *MV_GLOBAL_PTR(half8v, p); }
__device__ __forceinline__ void load_raw(RawChunk<float>& r, const float* p) {
    r.lo = *MV_GLOBAL_PTR(float4v, p);
    r.hi = *MV_GLOBAL_PTR(float4v, p + 4);
}
__device__ __forceinline__ float raw_get(const RawChunk<half_t>& r, int e) { return (float)r.v[e]; }
__device__ __forceinline__ float raw_get(const RawChunk<float>& r, int e) { return e < 4 ? r.lo[e] : r.hi[e - 4]; }

template <typename InT, bool HAS_X2, bool IN_AFFINE>
__global__ __launch_bounds__(CV_THREADS) void conv1d_mfma_kernel(ConvArgs a) {
    MV_DYN_SMEM(smem);
    int n_tile, co_tile;
    if (!tile_of_block(a, n_tile, co_tile)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1, wn = wave & 1;
    const int n0 = n_tile * CV_TN;
    const int co0 = co_tile * CV_TC;

    // loader mapping: thread owns 16-byte chunk kc of rows lrow + 32*i
    const int kc = tid & 7;
    const int lrow = tid >> 3;
    RowMap rm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + lrow + 32 * i;
        if (n < a.n_rows) {
            rm[i].b = n / a.T_out;
            rm[i].t = n - rm[i].b * a.T_out;
        } else {
            rm[i].b = -1;
            rm[i].t = 0;
        }
    }
    const InT* xbase = reinterpret_cast<const InT*>(a.x);
    const InT* x2base = reinterpret_cast<const InT*>(a.x2);
    const InT* zero = reinterpret_cast<const InT*>(g_zero_page);
    const half_t* zero_h = reinterpret_cast<const half_t*>(g_zero_page);
    const int kstages_per_tap = a.cin_pad / CV_BK;
    const int nstages = a.k * kstages_per_tap;

    RawChunk<InT> xr[4], x2r[4];
    half8v wr[4];

    auto issue_loads = [&](int s) {  // branch-free: padded / out-of-range chunks read the zero page
        const int tap = s / kstages_per_tap;
        const int c = (s - tap * kstages_per_tap) * CV_BK + kc * 8;
        const bool ch_ok = c < a.cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tin = input_time(a, rm[i].t, tap);
            const bool ok = rm[i].b >= 0 && tin >= 0 && ch_ok;
            const int64_t row = (int64_t)rm[i].b * a.T_in + tin;
            load_raw(xr[i], ok ? xbase + row * a.ldx + c : zero);
            if (HAS_X2) load_raw(x2r[i], ok ? x2base + row * a.ldx2 + c : zero);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + lrow + 32 * i;
            wr[i] = *MV_GLOBAL_PTR(half8v, co < a.cout_pad ? a.w + ((int64_t)co * a.k + tap) * a.cin_pad + c : zero_h);
        }
    };

    auto store_lds = [&](int s, int buf) {
        char* wt = smem + buf * ((CV_TC + CV_TN) * CV_BK * 2);
        char* xtile = wt + CV_TC * CV_BK * 2;
        const int tap = s / kstages_per_tap;
        const int c = (s - tap * kstages_per_tap) * CV_BK + kc * 8;
        float isc[8], ish[8];
        if (IN_AFFINE) {
            // channels beyond cin meet zero weights, any finite value will do
            const int cc = c + 8 <= a.cin ? c : 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                isc[e] = a.in_scale[cc + e];
                ish[e] = a.in_shift[cc + e];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lrow + 32 * i;
            half8v hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = raw_get(xr[i], e);
                if (HAS_X2) v += raw_get(x2r[i], e);
                if (IN_AFFINE) v = fmaxf(__builtin_fmaf(v, isc[e], ish[e]), 0.0f);   // (ONE rounding, spelled out: the device contracts `v * s + t` anyway, the host build of the emulator does not)
                hv[e] = to_half_sat(v);
            }
            *reinterpret_cast<half8v*>(xtile + lds_off(row, kc)) = hv;
            *reinterpret_cast<half8v*>(wt + lds_off(row, kc)) = wr[i];
        }
    };

    float4v acc[4][4];
    conv_acc_init<4, 4>(a, co0, wc, lane, acc);

    issue_loads(0);
    store_lds(0, 0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstages) issue_loads(s + 1);
        const char* wt = smem + buf * ((CV_TC + CV_TN) * CV_BK * 2);
        mma_stage<4, 4>(wt, wt + CV_TC * CV_BK * 2, wc, wn, lane, acc);
        if (s + 1 < nstages) store_lds(s + 1, buf ^ 1);
        __syncthreads();
    }
    conv_epilogue<4, 4>(a, n0, a.n_rows, co0, wc, wn, lane, acc);
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

// Streaming ("sc1 nt") output stores for the persistent kernels were measured and dropped: all workgroups reach their epilogues together, so a
// layer's output leaves as bursts of 32 MiB (in isolation the stores cost 27 us of a 187 us K = 1024 layer); with the streaming bits a
// K = 1024 layer alone runs 187 -> 173 us, but the layers that consume the output then miss in L2 / MALL: end to end 68.2 k -> 65.7 k
// utterances/s (r02d), and per role with the ring kernel 76.1 k -> 74.0-75.4 k (r05s).
static int cu_count() {
    return device_cu_count();   // (cached per device)
}

// Resident workgroups of the persistent kernel: one per CU (a multiple of 8 keeps  id mod 8 == XCD  across the walk).
// MvConv1dDesc.persist_blocks_hint overrides it (tests walk several tiles per workgroup on small problems; never the bits of a result).
static int persistent_blocks(const MvConv1dDesc& d) {
    const int v = d.persist_blocks_hint > 0 ? d.persist_blocks_hint : cu_count();
    return (int)round_up(v, 8);
}

int conv1d_launch(const MvConv1dDesc& d, hipStream_t stream) {
    MV_REQUIRE(d.x != nullptr && d.w_packed != nullptr && d.y != nullptr, "conv1d: null tensor");
    MV_REQUIRE(d.B > 0 && d.T_in > 0 && d.T_out > 0 && d.cin > 0 && d.cout > 0 && d.k > 0, "conv1d: bad geometry");
    MV_REQUIRE(d.dilation >= 1 && d.stride >= 1 && d.pad >= 0, "conv1d: bad dilation/stride/pad");
    MV_REQUIRE((int64_t)(d.T_out - 1) * d.stride - d.pad + (int64_t)(d.k - 1) * d.dilation < d.T_in + d.pad,
               "conv1d: T_out reaches beyond the padded input");
    if (d.pad_mode == MV_PAD_REFLECT) MV_REQUIRE(d.pad < d.T_in, "conv1d: reflect padding needs pad < T_in");
    MV_REQUIRE(d.x_dtype == MV_DT_F16 || d.x_dtype == MV_DT_F32, "conv1d: x dtype");
    MV_REQUIRE(d.y_dtype == MV_DT_F16 || d.y_dtype == MV_DT_F32, "conv1d: y dtype");
    MV_REQUIRE(d.pad_mode == MV_PAD_ZERO || d.pad_mode == MV_PAD_REFLECT, "conv1d: padding mode");
    MV_REQUIRE(d.pre_act >= MV_ACT_NONE && d.pre_act <= MV_ACT_SIGMOID && d.post_act >= MV_ACT_NONE && d.post_act <= MV_ACT_SIGMOID, "conv1d: activation code");
    // (a row may be SHORTER than cin -- the first block's window form reads 5 overlapping 80-channel rows as one of 400 -- but never of no or negative length)
    MV_REQUIRE(d.ldx > 0 && d.ldy > 0 && (d.x2 == nullptr || d.ldx2 > 0) && (d.sum_dst == nullptr || (d.ld_add > 0 && d.ld_sum > 0)),
               "conv1d: leading dimensions must be positive");
    MV_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "conv1d: in_scale/in_shift go together");
    MV_REQUIRE((d.scale == nullptr) == (d.shift == nullptr), "conv1d: scale/shift go together");
    // vector-access contract of the loader / epilogue
    const int xalign = d.x_dtype == MV_DT_F16 ? 8 : 4;
    MV_REQUIRE(d.cin % 8 == 0, "conv1d: input channels must be a multiple of 8");
    MV_REQUIRE(d.cout % 4 == 0, "conv1d: output channels must be a multiple of 4");
    for (const float* p : {d.bias, d.row_bias, d.scale, d.shift, d.gate})
        MV_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "conv1d: per-channel parameter arrays must be 16-byte aligned");
    MV_REQUIRE(d.ldx % xalign == 0 && (reinterpret_cast<uintptr_t>(d.x) & 15) == 0, "conv1d: x must be 16-byte aligned per row");
    if (d.x2 != nullptr)
        MV_REQUIRE(d.ldx2 % xalign == 0 && (reinterpret_cast<uintptr_t>(d.x2) & 15) == 0, "conv1d: x2 alignment");
    const int yalign_bytes = d.y_dtype == MV_DT_F16 ? 8 : 16;
    MV_REQUIRE(d.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(d.y) & (yalign_bytes - 1)) == 0, "conv1d: y alignment");
    if (d.gate != nullptr) MV_REQUIRE(d.gate_seg_len > 0, "conv1d: gate needs a segment length");
    if (d.in_scale != nullptr) MV_REQUIRE(d.pad == 0, "conv1d: the pre-activation on load is only defined for unpadded (1x1) convs");
    MV_REQUIRE((d.add_src == nullptr) == (d.sum_dst == nullptr), "conv1d: add_src/sum_dst go together");
    if (d.sum_dst != nullptr)
        MV_REQUIRE(d.ld_add % 4 == 0 && d.ld_sum % 4 == 0 && (reinterpret_cast<uintptr_t>(d.add_src) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(d.sum_dst) & 7) == 0,
                   "conv1d: second output alignment");
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
    a.add_src = reinterpret_cast<const half_t*>(d.add_src);
    a.sum_dst = reinterpret_cast<half_t*>(d.sum_dst);
    a.ld_add = d.ld_add;
    a.ld_sum = d.ld_sum;
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
    a.stat_sum = d.stat_sum;
    a.stat_sq = d.stat_sq;
    a.per_utt = 0;
    a.tiles_per_utt = 1;
    a.in_sum = d.in_stat_sum;
    a.in_sq = d.in_stat_sq;
    a.clock_probe = reinterpret_cast<unsigned long long*>(d.clock_probe);
    MV_REQUIRE((reinterpret_cast<uintptr_t>(d.clock_probe) & 7) == 0, "conv1d: clock_probe must be 8-byte aligned");
    const bool in_stats = d.in_stat_sum != nullptr;
    MV_REQUIRE((d.in_stat_sum == nullptr) == (d.in_stat_sq == nullptr), "conv1d: in_stat_sum / in_stat_sq go together");
    const int stats = d.stat_sum == nullptr ? 0 : (d.stat_sq == nullptr ? 1 : 2);
    if (d.stat_sq != nullptr) MV_REQUIRE(d.stat_sum != nullptr, "conv1d: stat_sq needs stat_sum");
    const bool f16 = d.x_dtype == MV_DT_F16;
    const bool has_x2 = d.x2 != nullptr, in_aff = d.in_scale != nullptr;
    // 256 x 256 tiles for wide layers with enough work to fill the chip (one workgroup per CU)
    MV_REQUIRE(d.tile == 0 || d.tile == 64 || d.tile == 128 || d.tile == 160 || d.tile == 256, "conv1d: tile must be 0 (auto), 64, 128, 160 or 256");
    const bool big_ok = f16 && !has_x2 && !in_aff && d.cout % 256 == 0;
    if (d.tile == 256) MV_REQUIRE(big_ok, "conv1d: 256-wide tiles need the plain fp16 path and cout % 256 == 0");
    // (auto: from 5/8 of a round of 256 x 256 tiles on -- 48 utterances of 3 s through a 1024-channel layer are 224 tiles = 39.5 us in one round, against 57-74 us
    //  on 128-row tiles and 60 us on 64 x 64 ones: profiles/r12j_conv_tiles_by_batch.log)
    const bool big = !in_stats && big_ok && d.tile != 128 && d.tile != 64 && d.tile != 160 &&
                     (d.tile == 256 || (int64_t)ceil_div(a.n_rows, 256) * (d.cout / 256) * 8 >= (int64_t)cu_count() * 5);
    // persistent form of the 256^2 kernel: fp16 output, plain bias / ReLU / affine epilogue
    const bool persist = big && d.y_dtype == MV_DT_F16 && d.sum_dst == nullptr && d.row_bias == nullptr && d.gate == nullptr &&
                         d.ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0 &&
                         (d.pre_act == MV_ACT_NONE || d.pre_act == MV_ACT_RELU) &&
                         (d.post_act == MV_ACT_NONE || d.post_act == MV_ACT_RELU) &&
                         (int64_t)d.k * conv1d_cin_pad(d.cin) >= 2 * CV_BK;  // at least two K stages per tile
    // 128 x 160 tile of the direct path: two workgroups per CU = 2 * CUs slots; taken when it saves a whole round of
    // workgroups (B*T = 76 288 rows x 128 channels: 477 tiles in one round instead of 596 tiles in two)
    const bool direct = f16 && !has_x2 && !in_aff;
    bool wide = false;
    if (direct && !big) {
        if (d.tile == 160 || in_stats) {
            wide = true;
        } else if (d.tile == 0) {
            const int64_t slots = 2 * (int64_t)cu_count(), cot = ceil_div(d.cout, CV_TC);
            const int64_t cost128 = ceil_div(ceil_div(a.n_rows, 128) * cot, slots) * 128;
            const int64_t cost160 = ceil_div(ceil_div(a.n_rows, 160) * cot, slots) * 160;
            wide = cost160 < cost128;
        }
    }
    if (d.tile == 160) MV_REQUIRE(direct && !big, "conv1d: the 160-row tile belongs to the plain fp16 path");
    // Small problems (one utterance, a handful: predict() / small predict_batch calls): 128 x 128 tiles leave most of the chip idle -- one 3 s
    // utterance through a 1024 -> 1024 layer is 3 x 8 = 24 workgroups walking 16 serial K stages each.  64 x 64 tiles give four times the
    // workgroups (each K stage is a quarter of the bytes: shorter round trips), still one launch.
    const bool small = direct && !big && !wide && d.cout >= 64 &&
                       (d.tile == 64 || (d.tile == 0 && ceil_div(a.n_rows, CV_TN) * ceil_div(d.cout, CV_TC) * 2 <= (int64_t)cu_count()));
    if (stats)
        MV_REQUIRE(persist && d.k == 1 && d.cin % CV_BK == 0 && d.T_out >= 64,
                   "conv1d: fused time statistics need the persistent 1x1 kernel (fp16 in/out, cout % 256 == 0, cin % 64 == 0, "
                   "plain bias / ReLU / affine epilogue, T_out >= 64)");
    if (in_stats && direct && d.k == 1 && d.stride == 1 && d.pad == 0 && d.T_in == d.T_out && d.cout <= CV_TC && d.tile == 0 && d.cin % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(d.in_stat_sum) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.in_stat_sq) & 15) == 0 &&
        (int64_t)d.B * ceil_div(d.T_out, CV_IN_STATS_TN) * 2 <= (int64_t)cu_count() && d.B <= 16384) {
        // small batch: the statistics from their own launch (bit-identical partial rows), the conv on small tiles
        a.per_utt = 1;
        a.tiles_per_utt = (int)ceil_div(d.T_out, CV_IN_STATS_TN);
        static DeviceOnce once;
        int slot;
        if (device_once_pending(once, &slot)) {
            if (MV_SET_MAX_SMEM(conv1d_in_stats_kernel, CV_IN_STATS_TN * CV_BK * 2) != hipSuccess) return fail(MV_ERR_HIP, "conv1d: cannot reserve dynamic LDS");
            device_once_done(once, slot);
        }
        MV_LAUNCH(conv1d_in_stats_kernel, ((unsigned)(d.B * a.tiles_per_utt), (unsigned)(a.cin_pad / CV_BK), 1), (CV_THREADS, 1, 1), CV_IN_STATS_TN * CV_BK * 2, stream, a);
        int rc = check_launch("conv1d_in_stats_kernel");
        if (rc != MV_OK) return rc;
        MvConv1dDesc d2 = d;
        d2.in_stat_sum = nullptr;
        d2.in_stat_sq = nullptr;
        return conv1d_launch(d2, stream);
    }
    if (in_stats) {
        MV_REQUIRE(direct && wide && d.k == 1 && d.stride == 1 && d.pad == 0 && d.T_in == d.T_out && d.cout <= CV_TC && d.tile != 128 &&
                       d.tile != 256 && (reinterpret_cast<uintptr_t>(d.in_stat_sum) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.in_stat_sq) & 15) == 0,
                   "conv1d: fused input statistics need the direct fp16 1x1 path with the 160-row tile (stride 1, no padding, cout <= 128)");
        a.per_utt = 1;
        a.tiles_per_utt = (int)ceil_div(d.T_out, CV_IN_STATS_TN);
    }
    const int tn = big ? 256 : (wide ? 160 : (small ? 64 : CV_TN)), tc = big ? 256 : (small ? 64 : CV_TC);
    a.n_tiles = a.per_utt ? d.B * a.tiles_per_utt : (int)ceil_div(a.n_rows, tn);
    a.co_tiles = (int)ceil_div(d.cout, tc);
    const int grid = a.n_tiles * a.co_tiles;
    static DeviceOnce smem_set;   // (per device: the attribute belongs to the current device's code object)
    int smem_set_slot;
    if (device_once_pending(smem_set, &smem_set_slot)) {
        if (MV_SET_MAX_SMEM((conv1d_glds_kernel<2, 2, 4, 4>), CV_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_kernel<2, 2, 2, 2, false, CV_SMALL_NS>), CV_LDS_BYTES_SMALL) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_kernel<4, 2, 2, 4, false, CV_TAIL_NS>), CV_LDS_BYTES_TAIL) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_kernel<2, 2, 4, 5>), CV_LDS_BYTES_WIDE) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_kernel<2, 2, 4, 5, true>), CV_LDS_BYTES_WIDE + CV_IN_STATS_BUF_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_kernel<2, 4, 8, 4>), CV_LDS_BYTES_BIG) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_persistent_kernel<true, 0>), CVP_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_persistent_kernel<true, 1>), CVP_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_persistent_kernel<true, 2>), CVP_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_glds_persistent_kernel<false, 0>), CVP_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM(conv1d_ring_persistent_kernel, CVR_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_mfma_kernel<float, false, false>), CV_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_mfma_kernel<half_t, true, false>), CV_LDS_BYTES) != hipSuccess ||
            MV_SET_MAX_SMEM((conv1d_mfma_kernel<half_t, false, true>), CV_LDS_BYTES) != hipSuccess)
            return fail(MV_ERR_HIP, "conv1d: cannot reserve dynamic LDS");
        device_once_done(smem_set, smem_set_slot);
    }
    const bool simple = d.k == 1 && d.cin % CV_BK == 0;
    // dense 1x1 rows (input row == output row) with 32-bit byte offsets into both tensors: the ring kernel's loader
    const bool dense_rows = simple && d.stride == 1 && d.pad == 0 && d.T_in == d.T_out &&
                            (int64_t)a.n_rows * d.ldx * 2 < ((int64_t)1 << 32) && (int64_t)d.cout * a.cin_pad * 2 < ((int64_t)1 << 32);
    // (a post-activation stays with the double-buffer kernel: the ring kernel's epilogue leaves the fp16 saturation to the hardware and has no v_med3 to
    //  fold a lower bound into -- no layer of the models has one behind its BatchNorm)
    const bool ring = persist && stats == 0 && dense_rows && d.post_act == MV_ACT_NONE;
    // The ring walk's tail.  A launch lasts as many tile times as its longest walk: 3600 tiles of an MFA layer (256 utterances of 300 frames x 3072
    // channels) on 256 workgroups are 14.06 rounds, so 15 -- the last one with 16 workgroups at work.  (The headline batch has 298 frames: 3576 tiles =
    // 13.97 rounds, nothing to split -- since the tile ids are dense; with the holes of the walk before round 6 it was 15 rounds as well.)  Splitting K over workgroups (stream-K) would fill it, but sums a tile
    // in another order, and WHICH tiles are split follows the batch: a row's bits would depend on its neighbours.  Splitting a tile's ROWS and CHANNELS
    // keeps every output element's sum as it is (test_gpu_embedding_bits_do_not_depend_on_the_batch_size holds the 64 / 128 / 256 tiles to one
    // accumulation order): when the last partial round's tiles, cut into sixteen 64 x 64 or four 128 x 128 sub-tiles, still fit one round of the chip,
    // the ring kernel walks the whole rounds only and the sub-tiles run at once on the small-tile kernels (four-stage rings).
    // Measured (r15ae / r15ag, per-launch events, MFA layer of the headline batch): the last round of the unsplit launch costs 50-55 us, not a tile time
    // of 84 -- its workgroups have the memory system to themselves; 192 quarters (48 tiles: the walk with holes of r15ae) take 44 us on eight waves of
    // 32 channels x 64 rows (60 on four of 64 x 64: one wave per SIMD cannot overlap its fragment reads with its MFMAs; two instead of three stages in
    // flight change nothing: the 128 x 128 tile is bound by LDS bandwidth, 96 KiB of fragment reads + 32 KiB of transfers per K stage = ~0.9 us where
    // the ring kernel's stage of four times the FLOPs takes 1.7).  So a tail that needs more than one round of sub-tiles does not pay (the K = 1024
    // layers: 176 tiles in the last round), and K stages below eight are not worth a second launch.
    int tail_vb0 = 0, tail_nv = 0, tail_split = 0;
    double tail_work = 0.0;
    if (ring) {
        const int blocks = persistent_blocks(d);
        const int total_ids = a.n_tiles * a.co_tiles;
        const int whole = total_ids / blocks * blocks;
        const int left = total_ids - whole;
        if (whole > 0 && left > 0 && 4 * left <= blocks && a.cin_pad / CV_BK >= 8) {
            tail_vb0 = whole;
            tail_nv = left;
            tail_split = 16 * left <= blocks ? 4 : 2;
            for (int v = whole; v < total_ids; ++v) {
                int nt = 0, ct = 0;
                tile_of_index_geom(a.n_tiles, a.co_tiles, v, nt, ct);
                const int rows = a.n_rows - nt * 256 < 256 ? a.n_rows - nt * 256 : 256;
                tail_work += 2.0 * rows * 256.0 * (double)d.cin;
            }
        }
    }
    const int prof = prof_begin(ring ? MV_PROF_CONV1D_RING : MV_PROF_CONV1D, 2.0 * a.n_rows * (double)d.cin * d.cout * d.k - tail_work, stream);
    if (persist) {
        const int64_t tiles = (int64_t)a.n_tiles * a.co_tiles;
        const int pgrid = (int)(tiles < persistent_blocks(d) ? round_up(tiles, 8) : persistent_blocks(d));
        if (ring) {
            a.ring_vb_end = tail_vb0;
            MV_LAUNCH(conv1d_ring_persistent_kernel, (pgrid, 1, 1), (512, 1, 1), CVR_LDS_BYTES, stream, a);
            if (tail_nv > 0) {
                prof_end(prof, stream);
                int rc = check_launch("conv1d_ring_persistent_kernel");
                if (rc != MV_OK) return rc;
                ConvArgs t = a;
                t.ring_vb_end = 0;
                t.clock_probe = nullptr;
                t.tail_vb0 = tail_vb0;
                t.tail_nv = tail_nv;
                t.tail_n_tiles = a.n_tiles;
                t.tail_co_tiles = a.co_tiles;
                t.tail_split = tail_split;
                const int tprof = prof_begin(MV_PROF_CONV1D, tail_work, stream);
                if (tail_split == 4) {
                    t.n_tiles = (int)ceil_div(a.n_rows, 64);
                    t.co_tiles = (int)ceil_div(d.cout, 64);
                    MV_LAUNCH((conv1d_glds_kernel<2, 2, 2, 2, false, CV_SMALL_NS>), (16 * tail_nv, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES_SMALL, stream, t);
                } else {
                    t.n_tiles = (int)ceil_div(a.n_rows, CV_TN);
                    t.co_tiles = (int)ceil_div(d.cout, CV_TC);
                    MV_LAUNCH((conv1d_glds_kernel<4, 2, 2, 4, false, CV_TAIL_NS>), (4 * tail_nv, 1, 1), (512, 1, 1), CV_LDS_BYTES_TAIL, stream, t);
                }
                prof_end(tprof, stream);
                return check_launch("conv1d_glds_kernel (ring tail)");
            }
        } else if (stats == 2) {
            MV_LAUNCH((conv1d_glds_persistent_kernel<true, 2>), (pgrid, 1, 1), (512, 1, 1), CVP_LDS_BYTES, stream, a);
        } else if (stats == 1) {
            MV_LAUNCH((conv1d_glds_persistent_kernel<true, 1>), (pgrid, 1, 1), (512, 1, 1), CVP_LDS_BYTES, stream, a);
        } else if (simple) {
            MV_LAUNCH((conv1d_glds_persistent_kernel<true, 0>), (pgrid, 1, 1), (512, 1, 1), CVP_LDS_BYTES, stream, a);
        } else {
            MV_LAUNCH((conv1d_glds_persistent_kernel<false, 0>), (pgrid, 1, 1), (512, 1, 1), CVP_LDS_BYTES, stream, a);
        }
    } else if (big) {
        MV_LAUNCH((conv1d_glds_kernel<2, 4, 8, 4>), (grid, 1, 1), (512, 1, 1), CV_LDS_BYTES_BIG, stream, a);
    } else if (wide && in_stats) {
        MV_LAUNCH((conv1d_glds_kernel<2, 2, 4, 5, true>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES_WIDE + CV_IN_STATS_BUF_BYTES, stream, a);
    } else if (wide) {
        MV_LAUNCH((conv1d_glds_kernel<2, 2, 4, 5>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES_WIDE, stream, a);
    } else if (small) {
        MV_LAUNCH((conv1d_glds_kernel<2, 2, 2, 2, false, CV_SMALL_NS>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES_SMALL, stream, a);
    } else if (f16 && !has_x2 && !in_aff) {
        MV_LAUNCH((conv1d_glds_kernel<2, 2, 4, 4>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    } else if (f16 && has_x2 && !in_aff) {
        MV_LAUNCH((conv1d_mfma_kernel<half_t, true, false>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    } else if (f16 && !has_x2 && in_aff) {
        MV_LAUNCH((conv1d_mfma_kernel<half_t, false, true>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    } else if (!f16 && !has_x2 && !in_aff) {
        MV_LAUNCH((conv1d_mfma_kernel<float, false, false>), (grid, 1, 1), (CV_THREADS, 1, 1), CV_LDS_BYTES, stream, a);
    } else {
        return fail(MV_ERR_UNSUPPORTED, "conv1d: this combination of input dtype / second input / pre-activation is not built");
    }
    prof_end(prof, stream);
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

int64_t mv_conv1d_stats_elems(int32_t B, int32_t T_out, int32_t cout) {
    return mv::ceil_div((int64_t)B * T_out, 256) * 4 * 2 * (int64_t)cout;  // [64-row blocks of whole 256-row tiles][2 slots][cout]
}

int mv_conv1d_stats_finish(const float* stat_sum, const float* stat_sq, const float* shift, int32_t B, int32_t T_out, int32_t cout,
                           float* mean, float* std, int64_t ld_out, float clamp_eps, mv_stream_t stream) {
    if (std != nullptr) MV_REQUIRE(stat_sq != nullptr, "mv_conv1d_stats_finish: std needs the sums of squares");
    return mv::conv_stats_finish_launch(stat_sum, std != nullptr ? stat_sq : nullptr, shift, B, T_out, cout, mean, std, ld_out, clamp_eps,
                                        static_cast<hipStream_t>(stream));
}

int64_t mv_conv1d_in_stats_elems(int32_t B, int32_t T_in, int32_t cin) { return mv::conv_in_stats_elems(B, T_in, cin); }

int mv_conv1d_in_stats_finish(const float* in_stat_sum, const float* in_stat_sq, int32_t B, int32_t T_in, int32_t cin, float* mean,
                              float* std, int64_t ld_out, float clamp_eps, mv_stream_t stream) {
    return mv::conv_in_stats_finish_launch(in_stat_sum, in_stat_sq, B, T_in, cin, mean, std, ld_out, clamp_eps, static_cast<hipStream_t>(stream));
}

int mv_conv1d_forward(const MvConv1dDesc* d, mv_stream_t stream) {
    MV_REQUIRE(d != nullptr, "mv_conv1d_forward: null descriptor");
    return mv::conv1d_launch(*d, static_cast<hipStream_t>(stream));
}

}  // extern "C"
