// CAM++ front-end, one BasicResBlock per launch (mvector/models/campplus.py:221-254):
//     mid = ReLU(BN1(conv3x3_stride(sf,1)(x)))        out = ReLU(BN2(conv3x3(mid)) + shortcut(x))
// with shortcut = BN(conv1x1_stride(sf,1)(x)) when the block is strided, the identity otherwise.  fcm.hip runs the two convs as two
// launches with the intermediate map making a round trip through HBM (write 1, read 1 of [B, Fout, T, 32] fp16) and the block input
// read twice (once as conv input, once as residual).  Here the intermediate map never leaves the CU and x is read once:
// per block HBM sees one read of x and one write of out -- the nine 3x3 convs of the head go from 770 to 410 row-units of traffic
// (one unit = B * T * 64 bytes), and the fp16 rounding of the intermediate map disappears from the error budget
// (profiles/r06_campp_error_budget.log).
//
// Workgroup = (utterance, time tile, band of output frequency rows) walking the band downwards one MID row per step, 8 waves in two
// groups that share each SIMD's matrix pipe (one wave of each group per SIMD):
//   * PRODUCERS (waves 0-3): the block input rows enter an LDS ring by LDS-DMA exactly as in fcm_band_kernel (each row once, LEAD
//     steps ahead, counted s_waitcnt, zero page for the padding in time and frequency); conv1 (9 MFMA taps, BN1 folded, ReLU)
//     produces mid row m for the tile's positions plus one column of halo on each side and writes it as fp16 into one of TWO LDS
//     slots (positions outside [0, T) are written as zeros: conv2 pads the mid map, not x);
//   * CONSUMERS (waves 4-7), one step behind: conv2 in scatter form -- mid row m contributes to output rows m+1, m, m-1 through the
//     tap rows df = 0, 1, 2, so the three output rows in progress live in three accumulator sets in registers and no ring of mid
//     rows is needed; the shortcut (tenth tap on the centre input row sf*m, which is in the ring while the producers work on mid row
//     m; identity = the permuted unit matrix, exact) goes into the accumulators of row m when they are born; after mid row m row m-1
//     is complete: bias, ReLU, one 16-byte store per position (8 maps);
//   * each group keeps its tap matrices in registers as MFMA A fragments with the rows permuted so that a lane owns consecutive maps
//     of one position (fcm.hip): a producer wave both map tiles of a quarter of the positions (8 maps per lane: 16-byte LDS
//     writes), a consumer wave one map tile of half of the positions (4 maps per lane, 8-byte stores).
// ONE barrier per step.  The first form of this kernel ran both convs on the same 4 waves (one wave per SIMD, two barriers per
// step): 146-155 us for the 40-row blocks, the matrix pipe idle during both epilogues, the row requests and the barriers
// (profiles/r06b); with two groups the epilogue / request / barrier time of one wave is matrix time of the other.
// Bands only exist while (utterance, time tile) pairs do not fill the chip (small batches); a band recomputes its two halo mid rows.
#include "kernels.h"

namespace mv {

constexpr int FBK_C = 32;  // feature maps

__device__ __attribute__((aligned(256))) const unsigned char g_fcmblk_zero_page[256] = {0};

struct FcmBlockArgs {
    const half_t* x;    // [B, Fin, T, 32]
    const half_t* w1;   // [9][32 co][32 ci], BN1 folded
    const float* b1;    // [32]
    const half_t* w2;   // [9 or 10][32][32], BN2 folded; tap 9 = shortcut 1x1 conv with its BN folded
    const float* b2;    // [32] BN2 shift (+ shortcut BN shift)
    half_t* y;
    int64_t y_sB, y_sF, y_sT;
    int B, T, Fin, Fout, shortcut, tile_out;  // tile_out: output positions per time tile (<= 64 * NT - 2)
    // C1 form (the head's first block): x is not read -- its rows are head.conv1 + bn1 + ReLU of the features, evaluated by the producers
    const float* feats;  // [B, T, Fin] fp32
    const half_t* c1a;   // [2 map tiles][64 lanes][8]: the MFMA A fragments of fcm_c1_pack
    const float* c1b;    // [32] folded BN shift
};

// byte offset of the 16-byte chunk `chunk` of position `pos` inside a row slot (64 B per position, chunks XOR-swizzled so that
// the 16 lanes x 4 chunks of a fragment read hit distinct banks)
__device__ __forceinline__ int fbk_off(int pos, int chunk) { return pos * 64 + ((chunk ^ ((pos >> 1) & 3)) << 4); }

template <int NT, int SF>
struct FcmBlk {
    static constexpr int MW = 64 * NT;             // mid positions per tile (4 waves x NT x 16): t0 - 1 ... t0 + MW - 2
    static constexpr int NTR = 4 * NT + 1;         // 1 KiB transfers per input row slot ((MW + 2) positions x 64 B rounded up)
    static constexpr int TPW = NT + 1;             // transfers per producer wave and row
    static constexpr int SLOT_BYTES = NTR * 1024;
    static constexpr int MID_BYTES = (MW + 2) * 64;  // + 2 positions that the masked last outputs read
    static constexpr int LDS_MAX = 160 * 1024;
    static constexpr int lead_fit = ((LDS_MAX - 1024 - 2 * MID_BYTES) / SLOT_BYTES - 3) / SF;
    static constexpr int LEAD = lead_fit > 4 ? 4 : lead_fit;  // steps of input rows in flight beyond the current one
    static constexpr int RING = 3 + SF * LEAD;
    static constexpr int MID_OFF = RING * SLOT_BYTES;
    static constexpr int DUMP_OFF = MID_OFF + 2 * MID_BYTES;
    static constexpr int LDS_BYTES = DUMP_OFF + 1024;
    static constexpr int AHEAD = (LEAD - 1) * SF * TPW;  // transfers that may stay in flight when a step starts
    static_assert(LEAD >= 1 && LDS_BYTES <= LDS_MAX && AHEAD <= 63, "fcm block kernel: ring does not fit");
};

template <int PH>
struct FbkPhase {  // accumulator sets of the consumers' step i (PH = i % 3): born / centre / completed
    static constexpr int N = PH, C = (PH + 2) % 3, P = (PH + 1) % 3;
};

// C1: the block input (head.conv1 + bn1 + ReLU of the fp32 features, campplus.py:262-264,283 -- one input map, 3 x 3, zero padding) is
// evaluated by the producers straight into the ring instead of being read from HBM: one MFMA per 16 positions and map tile with
//     K slot 8 dt + 2 df + p:  p = 0: x_hi, p = 1: x_lo of mel bin f - 1 + df at frame t + dt - 1;  slots 8 dt + 6, + 7 and dt = 3: unused
// (x = x_hi + x_lo as two fp16 values: the features enter exactly; the folded weights as fp16 like every other FCM conv).  A lane's B
// fragment is three packed (hi, lo) registers, two of them kept from the row below: one 4-byte load and one split per tile and row, no
// staging in LDS.  Measured (B = 256, T = 298, one box each; profiles/r09b, r09c): the block on the stored map 138-155 us per launch, with the
// conv inside 168-190 us (first form, three bins loaded and split per row: 179-197 us) against the 98 us launch it replaces; CAM++ end to end
// 122.2 k -> 128.3 k utterances/s (2.095 -> 1.995 ms per step), MV_FCM_C1=0 | 1 alternating in one call.  The [B, 80, T, 32] map
// (390 MB per 256 x 3 s batch, written by fcm_conv1_kernel and read back here) and its launch disappear.
template <int NT, int SF, bool C1 = false>
__global__ __launch_bounds__(512) void fcm_block_kernel(FcmBlockArgs a, int n_ttiles, int n_bands, int band_rows) {
    static_assert(!C1 || SF == 2, "the first block of the head is the strided one");
    typedef FcmBlk<NT, SF> G;
    MV_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = MV_UNIFORM(tid >> 6);
    const int wave = wave8 & 3;          // position share inside the group
    const bool producer = wave8 < 4;     // uniform
    const int fr = lane & 15, fg = lane >> 4;
    int wg = blockIdx.x;
    const int band = wg % n_bands;
    wg /= n_bands;
    const int tt = wg % n_ttiles;
    const int b = wg / n_ttiles;
    const int t0 = tt * a.tile_out;
    const int f0 = band * band_rows;
    const int f1 = f0 + band_rows < a.Fout ? f0 + band_rows : a.Fout;
    if (f1 <= f0) return;
    const int mlo = f0 > 0 ? f0 - 1 : 0;                     // mid rows the band needs
    const int mhi = f1 < a.Fout ? f1 : a.Fout - 1;
    const int nsteps = mhi - mlo + 1;
    const int rbase = SF * mlo - 1;                          // input row of ring index 0 (may be -1: zero padding)
    const int rel_last = SF * (nsteps - 1) + 2;              // last ring index the band needs

    if (producer) {
        const int pbase = wave * (16 * NT) + fr;
        // this lane's fragment addresses for the three time taps: position pbase + dt of a ring slot (tile ni is 16 positions = 1024
        // bytes further: 16 keeps the swizzle bits)
        unsigned a_in[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) a_in[dt] = lds_addr(smem) + (unsigned)fbk_off(pbase + dt, fg);
        // ================= conv1: x rows (ring) -> mid row (LDS) =================
        half8v w1f[9][2];
        float4v bias1v[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co = 8 * (fr >> 2) + 4 * mi + (fr & 3);  // A row fr of tile mi <-> map co: a lane ends up with maps 8*fg .. 8*fg+7
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) w1f[tap][mi] = *reinterpret_cast<const half8v*>(a.w1 + ((tap * FBK_C + co) * FBK_C + 8 * fg));
#pragma unroll
            for (int r = 0; r < 4; ++r) bias1v[mi][r] = a.b1[8 * fg + 4 * mi + r];
        }
        // this lane's share of a row transfer: slot position pi holds frame t0 + pi - 2.  Transfer i of this wave is the slot's KiB
        // j = wave + 4 * i = positions 16 * j ... 16 * j + 15; whether all / none of its lanes lie inside [0, T) is known per wave,
        // so a row request is TPW x (scalar slot address, scalar row base, per-lane 32-bit offset): the first form selected a 64-bit
        // source pointer per lane and transfer and cost ~1000 cycles of issue per row (profiles/r06j timeline)
        const half_t* zero = reinterpret_cast<const half_t*>(g_fcmblk_zero_page);
        unsigned xoff[G::TPW];   // byte offset inside an input row
        bool xok[G::TPW];        // this lane reads the row (otherwise zeros)
        int xkind[G::TPW];       // uniform: 2 = every lane reads the row, 0 = none does (padding / dump transfer), 1 = mixed
#pragma unroll
        for (int i = 0; i < G::TPW; ++i) {
            const int j = wave + 4 * i;
            const int q = j * 64 + lane;
            const int pi = q >> 2;
            const int c = (q & 3) ^ ((pi >> 1) & 3);
            const int t = t0 + pi - 2;
            const bool live = j < G::NTR;
            xok[i] = live && pi < G::MW + 2 && t >= 0 && t < a.T;
            xoff[i] = xok[i] ? (unsigned)(t * FBK_C + c * 8) * 2u : 0u;
            const int tlo = t0 + 16 * j - 2, thi = tlo + 15;
            const bool all = live && 16 * j + 15 < G::MW + 2 && tlo >= 0 && thi < a.T;
            const bool none = !live || 16 * j >= G::MW + 2 || thi < 0 || tlo >= a.T;
            xkind[i] = all ? 2 : (none ? 0 : 1);
        }
        const unsigned lds0 = lds_addr(smem);
        const half8v zero8 = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
        int islot = 0, irel = 0;  // ring slot / ring index of the next row to be requested
        struct RowReq {            // one row request, uniform: source row (or the zero page) and ring slot
            const half_t* base;
            unsigned slot;
            int fin, slot_off;     // input row, byte offset of the slot (C1)
            bool rok, win;         // win (C1): mel bin fin + 1 exists and the row is part of the band's sequence
        };
        auto next_row = [&]() {
            const int fin = rbase + irel;
            RowReq r;
            r.rok = fin >= 0 && fin < a.Fin && irel <= rel_last;
            r.fin = fin;
            r.win = fin + 1 >= 0 && fin + 1 < a.Fin && irel <= rel_last;
            r.slot_off = islot * G::SLOT_BYTES;
            r.base = C1 ? nullptr : a.x + ((int64_t)b * a.Fin + (r.rok ? fin : 0)) * a.T * FBK_C;
            r.slot = lds0 + (unsigned)(islot * G::SLOT_BYTES);
            ++irel;
            islot = islot + 1 == G::RING ? 0 : islot + 1;
            return r;
        };
        // transfer i (0 ... TPW - 1) of a row request
        auto issue_part = [&](const RowReq& r, int i) __attribute__((always_inline)) {
            const int j = wave + 4 * i;
            const unsigned dst = j < G::NTR ? r.slot + (unsigned)(j * 1024) : lds0 + (unsigned)G::DUMP_OFF;
            if (r.rok && xkind[i] == 1) {  // a tile edge inside this transfer: per-lane source (row or zero page)
                glds16_untracked(xok[i] ? reinterpret_cast<const char*>(r.base) + xoff[i] : reinterpret_cast<const char*>(zero), dst);
            } else {                       // everything from the row, or everything from the zero page: scalar selects only
                const bool from_row = r.rok && xkind[i] == 2;
                glds16_untracked_so_fresh(from_row ? static_cast<const void*>(r.base) : static_cast<const void*>(zero), from_row ? xoff[i] : 0u, dst);
            }
        };
        auto issue_row = [&]() {
            const RowReq r = next_row();
#pragma unroll
            for (int i = 0; i < G::TPW; ++i) issue_part(r, i);
        };
        // ---- C1: the ring rows are computed here ----
        // tile ni < NT of this wave = positions wave * 16 NT + 16 ni ... + 15 of the slot; tile NT = the slot's last two positions
        // (MW, MW + 1: wave 3, and only when the tile's last outputs reach them).  Position q holds frame t = t0 - 2 + q; lane (fr, fg) of
        // a tile supplies the K slots of the time tap dt = fg (fg = 3: unused slots, it repeats dt = 1) and ends up with the maps
        // 8 fg ... 8 fg + 7 of position fr.  Everything that depends on the lane and the tile only -- frame offsets, validity, slot
        // offsets -- is the same for every row.  Rows are made in ascending order, so a lane keeps the two lower mel bins of its tap frame
        // (already split and packed) from the previous row: a row costs ONE 4-byte load and one split per tile.
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        constexpr int NC1 = C1 ? NT + 1 : 1;
        half8v c1a[2];
        float4v c1bv[2];
        int c1_toff[NC1];       // element offset of the tap's frame inside the utterance's features (clamped into [0, T))
        unsigned c1_qoff[NC1];  // byte offset of (position, chunk fg) inside a slot
        unsigned c1_w0[NC1], c1_w1[NC1];  // (x_hi, x_lo) of mel bins f - 1, f of the NEXT row f to be made, at this lane's tap frame
        // bit ni: the tile is made by this wave (uniform) / every frame it touches lies inside [0, T) (uniform) / this lane's tap frame / position does
        unsigned c1_need = 0, c1_tile_inside = 0, c1_tap_ok = 0, c1_pos_ok = 0;
        const float* fb = nullptr;
        if constexpr (C1) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                c1a[mi] = *reinterpret_cast<const half8v*>(a.c1a + (mi * 64 + lane) * 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) c1bv[mi][r] = a.c1b[8 * fg + 4 * mi + r];
            }
            fb = a.feats + (int64_t)b * a.T * a.Fin;
            const int dtl = fg < 3 ? fg : 1;
            const int q_last = a.tile_out + 3;  // the last output (tile_out - 1) reads mid positions up to tile_out + 1, those read slot positions up to this
#pragma unroll
            for (int ni = 0; ni < NC1; ++ni) {
                const int q0 = ni < NT ? wave * (16 * NT) + 16 * ni : G::MW;
                const int q = q0 + fr;
                const int t = t0 - 2 + q;
                const int tt_ = t + dtl - 1;
                const int tc = tt_ < 0 ? 0 : (tt_ >= a.T ? a.T - 1 : tt_);
                c1_toff[ni] = tc * a.Fin;
                c1_qoff[ni] = (unsigned)fbk_off(q, fg);
                if (q0 <= q_last && (ni < NT || wave == 3)) c1_need |= 1u << ni;
                if (tt_ >= 0 && tt_ < a.T) c1_tap_ok |= 1u << ni;
                if (t >= 0 && t < a.T) c1_pos_ok |= 1u << ni;
                const int tlo = t0 - 2 + q0;  // frames tlo - 1 ... tlo + 16 are touched by the tile
                if (tlo - 1 >= 0 && tlo + 16 < a.T) c1_tile_inside |= 1u << ni;
            }
            c1_need = MV_UNIFORM(c1_need);
            c1_tile_inside = MV_UNIFORM(c1_tile_inside);
        }
        // x -> (x_hi, x_lo) as two fp16 values in one register; zero for a tap frame outside [0, T) (only edge tiles pay for the select)
        auto c1_split = [&](float y, int ni) __attribute__((always_inline)) {
            if (!((c1_tile_inside >> ni) & 1)) y = ((c1_tap_ok >> ni) & 1) ? y : 0.0f;
            y = fmed3(y, -65504.0f, 65504.0f);
            half2v hl;
            hl[0] = (half_t)y;
            hl[1] = (half_t)(y - (float)hl[0]);
            return __builtin_bit_cast(unsigned, hl);
        };
        // the one new mel bin a row needs (f + 1; nothing when that is the padding above the map or the row is behind the band)
        auto c1_load = [&](const RowReq& r, float (&pf)[NC1]) __attribute__((always_inline)) {
            if (!r.win) return;
            const float* rowp = fb + (r.fin + 1);
#pragma unroll
            for (int ni = 0; ni < NC1; ++ni)
                if ((c1_need >> ni) & 1) pf[ni] = rowp[c1_toff[ni]];
        };
        // conv + bias + ReLU of the row into its ring slot (zeros for rows outside the map and frames outside [0, T): conv1 of the block pads
        // THIS map, not the features), then the window moves up one bin
        auto c1_make = [&](const RowReq& r, const float (&pf)[NC1]) __attribute__((always_inline)) {
            char* const slot = smem + r.slot_off;
#pragma unroll
            for (int ni = 0; ni < NC1; ++ni) {
                if (!((c1_need >> ni) & 1)) continue;
                const unsigned pn = r.win ? c1_split(pf[ni], ni) : 0u;
                half8v o = zero8;
                if (r.rok) {
                    const uint4v bu = {c1_w0[ni], c1_w1[ni], pn, 0u};
                    const half8v bf = __builtin_bit_cast(half8v, bu);
                    const float4v m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1a[0], bf, c1bv[0], 0, 0, 0);
                    const float4v m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1a[1], bf, c1bv[1], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (half_t)fmed3(m0[e], 0.0f, 65504.0f);
                        o[4 + e] = (half_t)fmed3(m1[e], 0.0f, 65504.0f);
                    }
                    if (!((c1_tile_inside >> ni) & 1)) o = ((c1_pos_ok >> ni) & 1) ? o : zero8;
                }
                *reinterpret_cast<half8v*>(slot + c1_qoff[ni]) = o;
                c1_w0[ni] = c1_w1[ni];
                c1_w1[ni] = pn;
            }
        };
        if constexpr (C1) {  // the window in front of the first row rbase: mel bins rbase - 1, rbase (a band may start inside the map)
#pragma unroll
            for (int ni = 0; ni < NC1; ++ni) {
                c1_w0[ni] = c1_w1[ni] = 0u;
                if (!((c1_need >> ni) & 1)) continue;
                if (rbase - 1 >= 0 && rbase - 1 < a.Fin) c1_w0[ni] = c1_split(fb[c1_toff[ni] + rbase - 1], ni);
                if (rbase >= 0 && rbase < a.Fin) c1_w1[ni] = c1_split(fb[c1_toff[ni] + rbase], ni);
            }
        }
#pragma unroll 1
        for (int r = 0; r < SF * (G::LEAD - 1) + 3; ++r) {
            if constexpr (C1) {
                const RowReq rq = next_row();
                float pf[NC1];
                c1_load(rq, pf);
                c1_make(rq, pf);
            } else {
                issue_row();
            }
        }

        int cslot = 0;  // ring slot of the first input row of the current step
#pragma unroll 1
        for (int i = 0; i < nsteps; ++i) {
            if constexpr (!C1) wait_vm<G::AHEAD>();  // the input rows of this step have landed (this wave's share) ...
            lds_barrier();        // ... in every wave; the consumers are done with the mid slot written now and with the ring slots requested next
            // this step's row requests go out one or two per tap, behind that tap's MFMAs: a transfer takes ~150 cycles to issue
            // (r06k timeline: 900 cycles for six in a row, with the matrix pipe idle), the ten MFMAs in front of it keep the pipe busy
            // for as long
            RowReq req[SF];
#pragma unroll
            for (int s = 0; s < SF; ++s) req[s] = next_row();
            // C1: the features of the rows made at the end of this step are requested now (12 bytes per tile and lane: their latency runs
            // under the nine taps)
            float pf[SF][NC1];
            if constexpr (C1) {
#pragma unroll
                for (int s = 0; s < SF; ++s) c1_load(req[s], pf[s]);
            }
            // one tap = NT fragment reads + 2 * NT MFMAs; the reads of tap k + 1 are issued before the MFMAs of tap k (two register
            // groups, counted lgkmcnt)
            float4v acc1[2][NT];  // born in the first tap: the bias is that MFMA's C operand
            unsigned rowb[3];  // ring slots of the three input rows (uniform byte offsets)
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                int sl = cslot + df;
                sl = sl >= G::RING ? sl - G::RING : sl;
                rowb[df] = (unsigned)(sl * G::SLOT_BYTES);
            }
            half8v bq[2][NT];
            lds_read_tiles<NT>(bq[0], a_in[0] + rowb[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap == 0) {
                    lds_read_tiles<NT>(bq[1], a_in[1] + rowb[0]);
                    mfma_tiles2_init<NT, NT>(acc1[0], acc1[1], w1f[0][0], w1f[0][1], bq[0], bias1v[0], bias1v[1]);
                } else if (tap < 8) {
                    lds_read_tiles<NT>(bq[(tap + 1) & 1], a_in[(tap + 1) % 3] + rowb[(tap + 1) / 3]);
                    mfma_tiles2<NT, NT>(acc1[0], acc1[1], w1f[tap][0], w1f[tap][1], bq[tap & 1]);
                } else {
                    mfma_tiles2<NT, 0>(acc1[0], acc1[1], w1f[8][0], w1f[8][1], bq[0]);
                }
                if constexpr (!C1) {
#pragma unroll
                    for (int q = 0; q < SF * G::TPW; ++q)
                        if (q * 9 / (SF * G::TPW) == tap) issue_part(req[q / G::TPW], q % G::TPW);
                }
            }
            mfma_hazard_pad();
            mfma_hazard_pad();
            // mid row i -> slot i & 1 (positions t0 - 1 + p; zeros outside [0, T))
            char* const mid = smem + G::MID_OFF + (i & 1) * G::MID_BYTES;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int p = pbase + ni * 16;
                const int t = t0 - 1 + p;
                const bool ok = t >= 0 && t < a.T;
                const int tfirst = t0 - 1 + wave * (16 * NT) + ni * 16;  // uniform: the tile's positions tfirst ... tfirst + 15
                half8v o;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[4 * mi + r] = (half_t)fmed3(acc1[mi][ni][r], 0.0f, 65504.0f);
                if (tfirst < 0 || tfirst + 15 >= a.T) o = ok ? o : zero8;  // only edge tiles pay for the selects (scalar branch)
                *reinterpret_cast<half8v*>(mid + fbk_off(p, fg)) = o;
            }
            if constexpr (C1) {  // the ring rows of step i + LEAD
#pragma unroll
                for (int s = 0; s < SF; ++s) c1_make(req[s], pf[s]);
            }
            cslot += SF;
            cslot = cslot >= G::RING ? cslot - G::RING : cslot;
        }
        lds_barrier();  // the consumers' last step
        return;
    }

    // ================= conv2 (scatter form) + shortcut + ReLU: mid rows (LDS) -> output rows =================
    // a consumer wave owns ONE map tile (16 of the 32 maps) over half of the tile's positions: three accumulator sets of 2 * NT
    // position tiles = 24 * NT registers beside 40 of weights (with both map tiles per wave the sets alone would take 240 of the
    // 256 registers a wave has at two waves per SIMD)
    constexpr int N2 = 2 * NT;
    const int cmi = wave & 1;
    const int pbase = (wave >> 1) * (16 * N2) + fr;
    unsigned a_in2, a_mid[3];
    a_in2 = lds_addr(smem) + (unsigned)fbk_off(pbase + 2, fg);
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) a_mid[dt] = lds_addr(smem) + (unsigned)(G::MID_OFF + fbk_off(pbase + dt, fg));
    half8v w2f[10];
    float4v bias2;
    {
        const int co = 8 * (fr >> 2) + 4 * cmi + (fr & 3);  // A row fr <-> map co: the lane ends up with maps 8*fg + 4*cmi .. + 3
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w2f[tap] = *reinterpret_cast<const half8v*>(a.w2 + ((tap * FBK_C + co) * FBK_C + 8 * fg));
        if (a.shortcut) {
            w2f[9] = *reinterpret_cast<const half8v*>(a.w2 + ((9 * FBK_C + co) * FBK_C + 8 * fg));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) w2f[9][e] = (half_t)(8 * fg + e == co ? 1.0f : 0.0f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bias2[r] = a.b2[8 * fg + 4 * cmi + r];
    }
    // accumulator sets [set][half of this wave's positions][tile]; a set is born holding the bias
    float4v acc2[3][2][NT];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc2[s][h][ni] = bias2;

    // output row `fo` from an accumulator set: ReLU, 4 consecutive maps of one position per lane.  Address = uniform row / tile base
    // (scalar registers) + one per-lane 32-bit byte offset: ten hoisted 64-bit tile addresses do not fit beside the accumulators.
    const unsigned ylane = (unsigned)(((int64_t)(t0 + pbase) * a.y_sT + 8 * fg + 4 * cmi) * 2);
    const int jlim = (a.tile_out < a.T - t0 ? a.tile_out : a.T - t0) - pbase;  // tile k of this lane is stored iff 16 * k < jlim
    auto store_row = [&](const float4v (&acc)[2][NT], int fo) __attribute__((always_inline)) {
        char* const yrow = reinterpret_cast<char*>(a.y + (int64_t)b * a.y_sB + (int64_t)fo * a.y_sF);  // uniform
        const int64_t tile_bytes = 32 * a.y_sT;                                                          // 16 positions
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int k = h * NT + ni;
                half4v o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)fmed3(acc[h][ni][r], 0.0f, 65504.0f);
                if (16 * k < jlim) *MV_AS_GLOBAL(half4v, yrow + k * tile_bytes + ylane) = o;
            }
    };

    int cslot = 0;
    // step i (0 ... nsteps): the accumulators of output row mlo + i are born (set N) and take the shortcut tap from the centre
    // input row of the producers' step i; mid row mlo + i - 1 (written in the producers' step i - 1) is scattered into the rows
    // mlo + i (N), mlo + i - 1 (C), mlo + i - 2 (P); row mlo + i - 2 is then complete.
    // Fragments are read half a wave-share (NT tiles) at a time into two register groups: the reads of one half are in flight under
    // the MFMAs of the other.
    auto step = [&](auto ph, int i) __attribute__((always_inline)) {
        typedef decltype(ph) PH;
        lds_barrier();
        half8v bq[2][NT];
        const bool sc = i < nsteps, sm = i > 0;  // uniform
        const unsigned mo = (unsigned)(((i - 1) & 1) * G::MID_BYTES);
        if (sc) {
            int sl = cslot + 1;  // centre input row = row sf * (mlo + i) of x; output position j <-> slot position j + 2
            sl = sl >= G::RING ? sl - G::RING : sl;
            const unsigned ra = a_in2 + (unsigned)(sl * G::SLOT_BYTES);
            lds_read_tiles<NT>(bq[0], ra);
            lds_read_tiles<NT>(bq[1], ra + NT * 1024);
            // the set is born here: bias as the C operand of its first MFMAs (no initialisation instructions)
            if (sm) {
                mfma_tiles1_init<NT, NT>(acc2[PH::N][0], w2f[9], bq[0], bias2);
                lds_read_tiles<NT>(bq[0], a_mid[0] + mo);
                mfma_tiles1_init<NT, NT>(acc2[PH::N][1], w2f[9], bq[1], bias2);
                lds_read_tiles<NT>(bq[1], a_mid[0] + mo + NT * 1024);
            } else {
                mfma_tiles1_init<NT, NT>(acc2[PH::N][0], w2f[9], bq[0], bias2);
                mfma_tiles1_init<NT, 0>(acc2[PH::N][1], w2f[9], bq[1], bias2);
            }
        } else {  // last step: no row is born (the set only collects the df = 0 taps of a row outside the band)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc2[PH::N][h][ni] = bias2;
            lds_read_tiles<NT>(bq[0], a_mid[0] + mo);
            lds_read_tiles<NT>(bq[1], a_mid[0] + mo + NT * 1024);
        }
        if (sm) {
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (dt == 2 && h == 1) {
                        mfma_tiles1<NT, 0>(acc2[PH::N][h], w2f[0 + dt], bq[h]);
                    } else {
                        mfma_tiles1<NT, NT>(acc2[PH::N][h], w2f[0 + dt], bq[h]);
                    }
                    mfma_tiles1<NT, NT>(acc2[PH::C][h], w2f[3 + dt], bq[h]);
                    mfma_tiles1<NT, NT>(acc2[PH::P][h], w2f[6 + dt], bq[h]);
                    if (dt < 2) lds_read_tiles<NT>(bq[h], a_mid[dt + 1] + mo + h * (NT * 1024));
                }
            mfma_hazard_pad();
            mfma_hazard_pad();
            const int m = mlo + i - 1;  // the mid row just consumed
            if (m - 1 >= f0) store_row(acc2[PH::P], m - 1);
            if (i == nsteps && m < f1) store_row(acc2[PH::C], m);  // the band ends at the last row of the map
        }
        cslot += SF;
        cslot = cslot >= G::RING ? cslot - G::RING : cslot;
    };
#pragma unroll 1
    for (int i = 0; i <= nsteps; i += 3) {
        step(FbkPhase<0>(), i);
        if (i + 1 <= nsteps) step(FbkPhase<1>(), i + 1);
        if (i + 2 <= nsteps) step(FbkPhase<2>(), i + 2);
    }
}

template <int NT, int SF, bool C1 = false>
static int fcm_block_launch_one(const FcmBlockArgs& a, int n_ttiles, hipStream_t stream) {
    typedef FcmBlk<NT, SF> G;
    static DeviceOnce attr_set;   // (per device: the attribute belongs to the current device's code object)
    int attr_set_slot;
    if (device_once_pending(attr_set, &attr_set_slot)) {
        if (MV_SET_MAX_SMEM((fcm_block_kernel<NT, SF, C1>), G::LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "fcm block kernel: LDS size rejected");
        device_once_done(attr_set, attr_set_slot);
    }
    // one workgroup per CU is resident (LDS): bands only while (utterance, time tile) pairs alone do not fill the chip
    const int64_t pairs = (int64_t)a.B * n_ttiles;
    const int cus = device_cu_count();
    int n_bands = (int)ceil_div((int64_t)cus, pairs);
    const int max_bands = a.Fout / 4 > 0 ? a.Fout / 4 : 1;
    n_bands = n_bands < 1 ? 1 : (n_bands > max_bands ? max_bands : n_bands);
    const int band_rows = (int)ceil_div(a.Fout, n_bands);
    n_bands = (int)ceil_div(a.Fout, band_rows);
    MV_REQUIRE(pairs * n_bands < ((int64_t)1 << 31), "fcm_block: grid too large");
    MV_LAUNCH((fcm_block_kernel<NT, SF, C1>), ((unsigned)(pairs * n_bands), 1, 1), (512, 1, 1), G::LDS_BYTES, stream, a, n_ttiles, n_bands, band_rows);
    return check_launch("fcm_block_kernel");
}

template <int NT>
static int fcm_block_launch_nt(const FcmBlockArgs& a, int n_ttiles, hipStream_t stream) {
    if (a.feats != nullptr) return fcm_block_launch_one<NT, 2, true>(a, n_ttiles, stream);
    return a.Fin == a.Fout ? fcm_block_launch_one<NT, 1>(a, n_ttiles, stream) : fcm_block_launch_one<NT, 2>(a, n_ttiles, stream);
}

// head.conv1 (BN folded, [32 maps][3 df][3 dt] fp32) as the two MFMA A fragments of the C1 form: lane (fr, fg) of map tile mi holds row
// A[fr] = map 8 (fr >> 2) + 4 mi + (fr & 3), K slots 8 fg ... 8 fg + 7 = the weights of time tap dt = fg for the mel taps df = 0, 0, 1, 1, 2, 2
// (each meets x_hi and x_lo of its bin), two unused slots; fg = 3 is unused.  out: [2][64][8] fp16
void fcm_c1_pack(const float* w, half_t* out) {
    for (int mi = 0; mi < 2; ++mi)
        for (int lane = 0; lane < 64; ++lane) {
            const int fr = lane & 15, fg = lane >> 4;
            const int co = 8 * (fr >> 2) + 4 * mi + (fr & 3);
            for (int e = 0; e < 8; ++e) {
                float v = 0.0f;
                if (fg < 3 && e < 6) v = w[co * 9 + (e >> 1) * 3 + fg];
                out[(mi * 64 + lane) * 8 + e] = (half_t)v;
            }
        }
}

bool fcm_block_supported(const half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int T, int Fin) {
    return y_sB % 8 == 0 && y_sF % 8 == 0 && y_sT % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
           (int64_t)T * FBK_C * (int64_t)Fin < ((int64_t)1 << 31);
}

int fcm_block_launch(const half_t* x, int Fin, int sf, const half_t* w1, const float* b1, const half_t* w2, const float* b2, int shortcut,
                     half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int B, int T, hipStream_t stream, const float* feats,
                     const half_t* c1a, const float* c1b) {
    MV_REQUIRE((x != nullptr) != (feats != nullptr), "fcm_block: either the input map or the features it is made from");
    MV_REQUIRE(w1 != nullptr && b1 != nullptr && w2 != nullptr && b2 != nullptr && y != nullptr, "fcm_block: null tensor");
    MV_REQUIRE(B > 0 && T > 0 && Fin > 0 && (sf == 1 || sf == 2), "fcm_block: bad geometry");
    MV_REQUIRE(feats == nullptr || (c1a != nullptr && c1b != nullptr && sf == 2 && Fin >= 3 && (int64_t)B * T * Fin < ((int64_t)1 << 31)),
               "fcm_block: the first-conv form needs its weights, the strided block and at least 3 mel bins");
    MV_REQUIRE(sf == 1 || shortcut != 0, "fcm_block: a strided block needs its shortcut conv (the identity cannot change the row count)");
    MV_REQUIRE(fcm_block_supported(y, y_sB, y_sF, y_sT, T, Fin), "fcm_block: output rows must be 16-byte aligned and a map below 2^31 elements");
    FcmBlockArgs a;
    a.x = x;
    a.feats = feats;
    a.c1a = c1a;
    a.c1b = c1b;
    a.w1 = w1;
    a.b1 = b1;
    a.w2 = w2;
    a.b2 = b2;
    a.y = y;
    a.y_sB = y_sB;
    a.y_sF = y_sF;
    a.y_sT = y_sT;
    a.B = B;
    a.T = T;
    a.Fin = Fin;
    a.Fout = (Fin - 1) / sf + 1;
    a.shortcut = shortcut;
    // time tiles: as few as possible (every tile recomputes one column of mid halo per side), equal shares, then the smallest NT
    // that covers a share (+ 2 halo columns)
    constexpr int nt_max = 5;
    const int n_ttiles = (int)ceil_div(T, 64 * nt_max - 2);
    a.tile_out = (int)ceil_div(T, n_ttiles);
    const int nt = (int)ceil_div(a.tile_out + 2, 64);
    switch (nt) {
        case 1: return fcm_block_launch_nt<1>(a, n_ttiles, stream);
        case 2: return fcm_block_launch_nt<2>(a, n_ttiles, stream);
        case 3: return fcm_block_launch_nt<3>(a, n_ttiles, stream);
        case 4: return fcm_block_launch_nt<4>(a, n_ttiles, stream);
        default: return fcm_block_launch_nt<5>(a, n_ttiles, stream);
    }
}

}  // namespace mv

extern "C" {
int mv_fcm_c1_pack(const float* w, void* out) {
    if (w == nullptr || out == nullptr) return mv::fail(MV_ERR_INVALID_ARGUMENT, "mv_fcm_c1_pack: null argument");
    mv::fcm_c1_pack(w, reinterpret_cast<half_t*>(out));
    return MV_OK;
}

int mv_fcm_block_c1_f16(const float* feats, int32_t F, const void* c1a, const float* c1b, const void* w1, const float* b1, const void* w2,
                        const float* b2, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, mv_stream_t stream) {
    return mv::fcm_block_launch(nullptr, F, 2, reinterpret_cast<const half_t*>(w1), b1, reinterpret_cast<const half_t*>(w2), b2, 1,
                                reinterpret_cast<half_t*>(y), y_sB, y_sF, y_sT, B, T, static_cast<hipStream_t>(stream), feats,
                                reinterpret_cast<const half_t*>(c1a), c1b);
}

int mv_fcm_block_f16(const void* x, int32_t Fin, int32_t sf, const void* w1, const float* b1, const void* w2, const float* b2,
                     int32_t shortcut, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, mv_stream_t stream) {
    return mv::fcm_block_launch(reinterpret_cast<const half_t*>(x), Fin, sf, reinterpret_cast<const half_t*>(w1), b1,
                                reinterpret_cast<const half_t*>(w2), b2, shortcut, reinterpret_cast<half_t*>(y), y_sB, y_sF, y_sT, B, T,
                                static_cast<hipStream_t>(stream));
}
}
