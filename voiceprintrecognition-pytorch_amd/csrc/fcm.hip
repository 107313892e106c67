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
#include "s16map.h"

namespace mv {

constexpr int FCM_TT = 128;  // frames per tile
constexpr int FCM_C = 32;    // feature maps

// 256 zero bytes: source of every padded 16-byte chunk of the band kernel's row transfers
__device__ __attribute__((aligned(256))) const unsigned char g_fcm_zero_page[256] = {0};

// ---- first conv: one input map (the fp32 features [B, T, F], read transposed), K = 9 -> plain VALU -----------------
// Workgroup = (utterance, band of 16 frequency rows, 64 frames).  The feature tile (66 frames x 18 bins, zero padded) is staged in
// LDS with loads that run along the frequency axis (the contiguous one of [B, T, F]); thread = (frame, 8 of the 32 maps) then walks
// the 16 rows of the band with a sliding 3 x 3 window (3 new LDS reads per position), its 72 weights + 8 biases in registers, and
// writes 16 bytes per position: a row of the band leaves as one contiguous 4 KiB run of the [B, F, T, 32] output.  (First form:
// nine scattered global loads per thread and position, 320 bytes apart across the wave -- 181 us for a 397 MB output.)
constexpr int FC1_FB = 16, FC1_TT = 64;

__global__ __launch_bounds__(256) void fcm_conv1_kernel(const float* feats, half_t* out, const float* w, const float* bias,
                                                        int B, int T, int F, int n_tt, int n_fb) {
    __shared__ float tile[(FC1_TT + 2) * (FC1_FB + 2)];
    const int tid = threadIdx.x;
    int wg = blockIdx.x;
    const int tt = wg % n_tt;
    wg /= n_tt;
    const int fb = wg % n_fb;
    const int b = wg / n_fb;
    const int t0 = tt * FC1_TT, f0 = fb * FC1_FB;
    for (int i = tid; i < (FC1_TT + 2) * (FC1_FB + 2); i += 256) {
        const int tr = i / (FC1_FB + 2), fc = i - tr * (FC1_FB + 2);
        const int t = t0 + tr - 1, f = f0 + fc - 1;
        tile[i] = (t >= 0 && t < T && f >= 0 && f < F) ? feats[((int64_t)b * T + t) * F + f] : 0.0f;
    }
    const int cc = tid & 3, tl = tid >> 2;  // 8 maps, frame inside the tile
    float wr[8][9], br[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        br[e] = bias[cc * 8 + e];
#pragma unroll
        for (int j = 0; j < 9; ++j) wr[e][j] = w[(cc * 8 + e) * 9 + j];
    }
    __syncthreads();
    const int t = t0 + tl;
    // window x[df][dt] = feature(f + df - 1, t + dt - 1) = tile[(tl + dt) * (FB + 2) + (fl + df)]
    float x[3][3];
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) x[df + 1][dt] = tile[(tl + dt) * (FC1_FB + 2) + df];
    for (int fl = 0; fl < FC1_FB; ++fl) {
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            x[0][dt] = x[1][dt];
            x[1][dt] = x[2][dt];
            x[2][dt] = tile[(tl + dt) * (FC1_FB + 2) + fl + 2];
        }
        const int f = f0 + fl;
        if (f >= F) break;  // uniform
        half8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = br[e];
#pragma unroll
            for (int df = 0; df < 3; ++df)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) acc = fmaf(wr[e][df * 3 + dt], x[df][dt], acc);
            o[e] = (half_t)fmed3(acc, 0.0f, 65504.0f);
        }
        if (t < T) *reinterpret_cast<half8v*>(out + (((int64_t)b * F + f) * T + t) * FCM_C + cc * 8) = o;
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


// ---- band kernel: one workgroup walks a band of output frequency rows with the input rows in an LDS ring --------------
// The row kernel above stages three input rows per (utterance, output row) workgroup and computes with nothing in flight: every
// input row is fetched three times (from L2 at best) and each 128-frame tile pays a full memory round trip (measured 2.3-2.5
// TB/s on tensors that are read and written exactly once).  Here a workgroup owns (utterance, time tile of 64*NI frames, band
// of output rows) and walks the band downwards:
//   * input rows enter a ring of RING row slots by LDS-DMA (global_load_lds, 16 B per lane, no registers), each row ONCE, 2-4
//     rows ahead of the row being computed; the residual input (shortcut 1x1 tap / identity add of BasicResBlock) rides a
//     second ring of 2 slots; one counted s_waitcnt + one barrier per output row;
//   * the nine (ten) [32 x 32] tap matrices stay in registers as MFMA A fragments with their ROWS PERMUTED
//     (A row i of tile mi <-> output map 8*(i>>2) + 4*mi + (i&3)), so a lane ends up with 8 consecutive maps of one
//     position: one 16-byte store, a 16-lane group writes a whole 1 KiB run;
//   * the identity residual is a tenth tap with the (permuted) identity matrix: exact in fp16 x fp32-accumulate.
// LDS: RING slots of (64*NI + 2) positions x 64 B rounded up to whole 1 KiB transfers (+ 2 residual slots + one dump KiB that
// swallows the transfers of waves whose share of a row is one short, so every wave issues the same number per row and the
// waits can be counted).  NI = 5: 7 x 21 KiB + 1 = 148 KiB (no residual), 5 x 21 + 2 x 20 + 1 = 146 KiB.
template <int NI, int SF, bool HAS2>
struct FcmBand {
    static constexpr int RING = HAS2 ? 5 : 7;
    static constexpr int POS = 64 * NI;                 // output positions per time tile (4 waves x NI x 16)
    static constexpr int NTR = 4 * NI + 1;              // 1 KiB transfers per input row slot ((POS + 2) * 64 B rounded up)
    static constexpr int TPW = NI + 1;                  // transfers per wave and row
    static constexpr int SLOT_BYTES = NTR * 1024;
    static constexpr int RES_BYTES = POS * 64;
    static constexpr int DUMP_OFF = RING * SLOT_BYTES + (HAS2 ? 2 * RES_BYTES : 0);
    static constexpr int LDS_BYTES = DUMP_OFF + 1024;
    static constexpr int AHEAD = (RING - 3 - SF) * TPW;  // transfers that may stay in flight when a row starts
};

template <int NI, int SF, bool HAS2>
__global__ __launch_bounds__(256) void fcm_band_kernel(FcmConvArgs a, int n_ttiles, int n_bands, int band_rows) {
    typedef FcmBand<NI, SF, HAS2> G;
    MV_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = MV_UNIFORM(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    int wg = blockIdx.x;
    const int band = wg % n_bands;
    wg /= n_bands;
    const int tt = wg % n_ttiles;
    const int b = wg / n_ttiles;
    const int t0 = tt * G::POS;
    const int f0 = band * band_rows;
    const int f1 = f0 + band_rows < a.Fout ? f0 + band_rows : a.Fout;
    const int nsteps = f1 - f0;
    if (nsteps <= 0) return;
    const int g0 = f0 * SF - 1;                 // first input row of the band (may be -1: zero padding)
    const int rel_last = (nsteps - 1) * SF + 2;  // last input row the band needs (relative to g0)

    // ---- weights: A fragments with permuted rows, bias of this lane's 8 maps ----
    const int ntaps = a.mode2 == 1 ? 10 : 9;
    half8v wf[10][2];
#pragma unroll
    for (int tap = 0; tap < 10; ++tap)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co = 8 * (fr >> 2) + 4 * mi + (fr & 3);
            if (tap < ntaps) {
                wf[tap][mi] = *reinterpret_cast<const half8v*>(a.w + ((tap * FCM_C + co) * FCM_C + 8 * fg));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[tap][mi][e] = (half_t)((HAS2 && a.mode2 == 2 && 8 * fg + e == co) ? 1.0f : 0.0f);
            }
        }
    float bias[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mi][r] = a.bias[8 * fg + 4 * mi + r];

    // ---- this lane's share of a row transfer: element offset inside an input row (-1 = zero padding in time) ----
    const half_t* zero = reinterpret_cast<const half_t*>(g_fcm_zero_page);
    int xoff[G::TPW];
    bool xlive[G::TPW];
#pragma unroll
    for (int i = 0; i < G::TPW; ++i) {
        const int j = wave + 4 * i;
        const int q = j * 64 + lane;         // 16-byte chunk inside the slot
        const int pi = q >> 2;               // slot position: frame t0 + pi - 1
        const int c = (q & 3) ^ ((pi >> 1) & 3);
        const int t = t0 + pi - 1;
        xlive[i] = j < G::NTR;
        xoff[i] = (xlive[i] && pi < G::POS + 2 && t >= 0 && t < a.T) ? t * FCM_C + c * 8 : -1;
    }
    int roff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int pi = q >> 2;
        const int c = (q & 3) ^ ((pi >> 1) & 3);
        const int t = t0 + pi;
        roff[i] = t < a.T ? t * FCM_C + c * 8 : -1;
    }
    char* const dump = smem + G::DUMP_OFF;

    int islot = 0;  // ring slot of the next row to be requested
    int irel = 0;   // its index relative to g0
    auto issue_row = [&]() {
        const int fin = g0 + irel;
        const bool rok = fin >= 0 && fin < a.Fin && irel <= rel_last;  // uniform
        const half_t* base = a.x + ((int64_t)b * a.Fin + (rok ? fin : 0)) * a.T * FCM_C;
        char* slot = smem + islot * G::SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < G::TPW; ++i) {
            const half_t* src = (rok && xoff[i] >= 0) ? base + xoff[i] : zero;
            glds16(src, xlive[i] ? slot + (wave + 4 * i) * 1024 : dump);
        }
        ++irel;
        islot = islot + 1 == G::RING ? 0 : islot + 1;
    };
    auto issue_res = [&](int k) {
        const bool rok = k < nsteps;
        const int f2 = (f0 + (rok ? k : 0)) * a.sf2;
        const half_t* base = a.x2 + ((int64_t)b * a.F2 + f2) * a.T * FCM_C;
        char* slot = smem + G::RING * G::SLOT_BYTES + (k & 1) * G::RES_BYTES;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const half_t* src = (rok && roff[i] >= 0) ? base + roff[i] : zero;
            glds16(src, slot + (wave + 4 * i) * 1024);
        }
    };

    if (HAS2) issue_res(0);
#pragma unroll 1
    for (int r = 0; r < G::RING - SF; ++r) issue_row();

    int cslot = 0;  // ring slot of the first input row of the current output row
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        wait_vm<G::AHEAD>();   // everything but the youngest AHEAD transfers of this wave has landed ...
        lds_barrier();         // ... in every wave; all waves are done with the slots the next requests overwrite
        if (HAS2) issue_res(k + 1);
#pragma unroll
        for (int s = 0; s < SF; ++s) issue_row();

        float4v acc[2][NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = float4v{bias[mi][0], bias[mi][1], bias[mi][2], bias[mi][3]};
        const int pbase = wave * (16 * NI) + fr;
#pragma unroll
        for (int df = 0; df < 3; ++df) {
            int sl = cslot + df;
            sl = sl >= G::RING ? sl - G::RING : sl;
            const char* row = smem + sl * G::SLOT_BYTES;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const half8v bf = *reinterpret_cast<const half8v*>(row + fcm_lds_off(pbase + ni * 16 + dt, fg));
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[df * 3 + dt][mi], bf, acc[mi][ni], 0, 0, 0);
                }
            }
        }
        if (HAS2) {
            const char* row = smem + G::RING * G::SLOT_BYTES + (k & 1) * G::RES_BYTES;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const half8v bf = *reinterpret_cast<const half8v*>(row + fcm_lds_off(pbase + ni * 16, fg));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[9][mi], bf, acc[mi][ni], 0, 0, 0);
            }
        }
        // ---- epilogue: ReLU, 8 consecutive maps of one position per lane ----
        const int fo = f0 + k;
        half_t* yrow = a.y + (int64_t)b * a.y_sB + (int64_t)fo * a.y_sF + 8 * fg;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 + pbase + ni * 16;
            half8v o;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[4 * mi + r] = (half_t)fmed3(acc[mi][ni][r], 0.0f, 65504.0f);
            if (t < a.T) *MV_AS_GLOBAL(half8v, yrow + (int64_t)t * a.y_sT) = o;
        }
        cslot += SF;
        cslot = cslot >= G::RING ? cslot - G::RING : cslot;
    }
}

// fp32 head (campplus.hip): [B, F8, T, 32] fp32 -> [B, T, F8, 32] fp16; thread = 8 maps of one (b, f, t)
// S16: the maps in the split-fp16 form of conv2ds.hip (s16map.h) instead of fp32
template <bool S16>
__global__ __launch_bounds__(256) void fcm_rows_from_f32_kernel(const float* maps, half_t* rows, int64_t n8, int T, int F8, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c8 = (int)(i & 3);
    int64_t p = i >> 2;               // (b * F8 + f) * T + t
    const int t = (int)(p % T);
    p /= T;
    const int f = (int)(p % F8);
    const int64_t b = p / F8;
    float4v lo, hi;
    if (S16) {   // eight channels c8 * 8 ... of a pixel: unit c8 / 2, quadruples 2 * (c8 & 1) and 2 * (c8 & 1) + 1
        const half_t* unit = reinterpret_cast<const half_t*>(maps) + ((i >> 2) * 32 + (c8 >> 1) * 16) * 2 + (c8 & 1) * 8;
        lo = s16_load4(unit);
        hi = s16_load4(unit + 4);
    } else {
        lo = *reinterpret_cast<const float4v*>(maps + i * 8);
        hi = *reinterpret_cast<const float4v*>(maps + i * 8 + 4);
    }
    half8v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (half_t)s16_clamp(lo[e] * scale);   // saturates at the fp16 range like the fp16 head's maps; a NaN stays a NaN
        o[4 + e] = (half_t)s16_clamp(hi[e] * scale);
    }
    *reinterpret_cast<half8v*>(rows + ((b * T + t) * F8 + f) * FCM_C + c8 * 8) = o;
}

int fcm_rows_from_f32_launch(const float* maps, half_t* rows, int B, int T, int F8, hipStream_t stream) {
    MV_REQUIRE(maps != nullptr && rows != nullptr && B > 0 && T > 0 && F8 > 0, "fcm_rows_from_f32: bad argument");
    const int64_t n8 = (int64_t)B * F8 * T * 4;
    MV_REQUIRE(ceil_div(n8, 256) < ((int64_t)1 << 31), "fcm_rows_from_f32: grid too large");
    MV_LAUNCH(fcm_rows_from_f32_kernel<false>, ((unsigned)ceil_div(n8, 256), 1, 1), (256, 1, 1), 0, stream, maps, rows, n8, T, F8, 1.0f);
    return check_launch("fcm_rows_from_f32_kernel");
}

int fcm_rows_from_s16_launch(const half_t* maps, half_t* rows, int B, int T, int F8, hipStream_t stream, float scale) {
    MV_REQUIRE(maps != nullptr && rows != nullptr && B > 0 && T > 0 && F8 > 0, "fcm_rows_from_s16: bad argument");
    const int64_t n8 = (int64_t)B * F8 * T * 4;
    MV_REQUIRE(ceil_div(n8, 256) < ((int64_t)1 << 31), "fcm_rows_from_s16: grid too large");
    MV_LAUNCH(fcm_rows_from_f32_kernel<true>, ((unsigned)ceil_div(n8, 256), 1, 1), (256, 1, 1), 0, stream, reinterpret_cast<const float*>(maps), rows, n8, T, F8, scale);
    return check_launch("fcm_rows_from_f32_kernel");
}

int fcm_conv1_launch(const float* feats, half_t* out, const float* w, const float* bias, int B, int T, int F,
                     hipStream_t stream) {
    MV_REQUIRE(feats != nullptr && out != nullptr && w != nullptr && bias != nullptr && B > 0 && T > 0 && F > 0, "fcm_conv1: bad argument");
    const int n_tt = (int)ceil_div(T, FC1_TT), n_fb = (int)ceil_div(F, FC1_FB);
    MV_REQUIRE((int64_t)B * n_tt * n_fb < ((int64_t)1 << 31), "fcm_conv1: grid too large");
    MV_LAUNCH(fcm_conv1_kernel, ((unsigned)(B * n_tt * n_fb), 1, 1), (256, 1, 1), 0, stream, feats, out, w, bias, B, T, F, n_tt, n_fb);
    return check_launch("fcm_conv1_kernel");
}

template <int NI, int SF, bool HAS2>
static int fcm_band_launch_one(const FcmConvArgs& a, int n_ttiles, hipStream_t stream) {
    typedef FcmBand<NI, SF, HAS2> G;
    static DeviceOnce attr_set;   // (per device: the attribute belongs to the current device's code object)
    int attr_set_slot;
    if (device_once_pending(attr_set, &attr_set_slot)) {
        if (MV_SET_MAX_SMEM((fcm_band_kernel<NI, SF, HAS2>), G::LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "fcm band kernel: LDS size rejected");
        device_once_done(attr_set, attr_set_slot);
    }
    // one workgroup per CU is resident (LDS): split the output rows into bands only while (utterance, time tile) pairs alone do
    // not fill the chip; every band re-reads its two halo rows
    const int64_t pairs = (int64_t)a.B * n_ttiles;
    const int cus = device_cu_count();
    int n_bands = (int)ceil_div((int64_t)cus, pairs);
    const int max_bands = a.Fout / 4 > 0 ? a.Fout / 4 : 1;
    n_bands = n_bands < 1 ? 1 : (n_bands > max_bands ? max_bands : n_bands);
    const int band_rows = (int)ceil_div(a.Fout, n_bands);
    n_bands = (int)ceil_div(a.Fout, band_rows);
    MV_REQUIRE(pairs * n_bands < ((int64_t)1 << 31), "fcm_conv3x3: grid too large");
    MV_LAUNCH((fcm_band_kernel<NI, SF, HAS2>), ((unsigned)(pairs * n_bands), 1, 1), (256, 1, 1), G::LDS_BYTES, stream, a, n_ttiles, n_bands,
              band_rows);
    return check_launch("fcm_band_kernel");
}

template <int NI>
static int fcm_band_launch_ni(const FcmConvArgs& a, int n_ttiles, hipStream_t stream) {
    if (a.mode2 != 0) return fcm_band_launch_one<NI, 1, true>(a, n_ttiles, stream);
    if (a.sf == 2) return fcm_band_launch_one<NI, 2, false>(a, n_ttiles, stream);
    return fcm_band_launch_one<NI, 1, false>(a, n_ttiles, stream);
}

int fcm_conv3x3_launch(const half_t* x, int Fin, int sf, const half_t* x2, int F2, int sf2, int mode2, const half_t* w,
                       const float* bias, half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int B, int T, int Fout,
                       hipStream_t stream) {
    MV_REQUIRE(x != nullptr && w != nullptr && bias != nullptr && y != nullptr, "fcm_conv3x3: null tensor");
    MV_REQUIRE(mode2 == 0 || x2 != nullptr, "fcm_conv3x3: residual input missing");
    MV_REQUIRE(B > 0 && T > 0 && Fin > 0 && Fout > 0, "fcm_conv3x3: bad geometry");
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
    // band kernel: 16-byte stores need 8-map alignment of the output rows; the residual conv is never strided (BasicResBlock)
    const bool band_ok = (sf == 1 || sf == 2) && (mode2 == 0 || sf == 1) && y_sB % 8 == 0 && y_sF % 8 == 0 && y_sT % 8 == 0 &&
                         (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (int64_t)T * FCM_C * (int64_t)(Fin > F2 ? Fin : F2) < ((int64_t)1 << 31);
    if (band_ok) {  // (else: the one-row-per-workgroup kernel below, 8-byte stores, any 4-element-aligned output layout)
        // time tiles of 64 * NI frames, NI <= 5: as few tiles as possible, then the smallest NI that covers T
        const int n16 = (int)ceil_div(T, 16);
        const int n_ttiles = (int)ceil_div(n16, 20);
        const int ni = (int)ceil_div(n16, 4 * n_ttiles);
        switch (ni) {
            case 1: return fcm_band_launch_ni<1>(a, n_ttiles, stream);
            case 2: return fcm_band_launch_ni<2>(a, n_ttiles, stream);
            case 3: return fcm_band_launch_ni<3>(a, n_ttiles, stream);
            case 4: return fcm_band_launch_ni<4>(a, n_ttiles, stream);
            default: return fcm_band_launch_ni<5>(a, n_ttiles, stream);
        }
    }
    MV_LAUNCH(fcm_conv3x3_kernel, ((unsigned)(B * Fout), 1, 1), (256, 1, 1), 0, stream, a);
    return check_launch("fcm_conv3x3_kernel");
}

}  // namespace mv

extern "C" {
int mv_fcm_conv3x3_f16(const void* x, int32_t Fin, int32_t sf, const void* x2, int32_t F2, int32_t sf2, int32_t mode2, const void* w,
                       const float* bias, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, int32_t Fout,
                       mv_stream_t stream) {
    MV_REQUIRE(Fin > 0 && (sf == 1 || sf == 2) && Fout == (Fin - 1) / sf + 1, "fcm_conv3x3: Fout must be (Fin - 1) / sf + 1");
    MV_REQUIRE(mode2 >= 0 && mode2 <= 2, "fcm_conv3x3: mode2 must be 0, 1 or 2");
    MV_REQUIRE(mode2 == 0 || (sf2 >= 1 && F2 > (Fout - 1) * sf2), "fcm_conv3x3: residual rows out of range");
    return mv::fcm_conv3x3_launch(reinterpret_cast<const half_t*>(x), Fin, sf, reinterpret_cast<const half_t*>(x2), F2, sf2, mode2,
                                  reinterpret_cast<const half_t*>(w), bias, reinterpret_cast<half_t*>(y), y_sB, y_sF, y_sT, B, T, Fout,
                                  static_cast<hipStream_t>(stream));
}
}
