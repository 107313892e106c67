// 2-D convolutions of the ERes2Net family (mvector/models/eres2net.py:23-30) on the fp16 matrix pipe with fp32-grade operands
// ("split" form, round 4).  conv2d.hip evaluates the same layers with fp32 maps and weights on v_mfma_f32_16x16x4_f32, 1/16 of the fp16
// MFMA rate; the family does not tolerate 11-bit operands, but it does tolerate 22-bit ones:
//
//   every operand value v is carried as a pair of fp16 numbers  hi = fp16(V), lo = fp16(V - hi)  of the scaled value V = v * 2^k, and a
//   product sum is evaluated as  sum Whi.Xhi + Whi.Xlo + Wlo.Xhi  in ONE fp32 accumulator (three v_mfma_f32_16x16x32_f16 per 32 channels:
//   3/16 of the fp32 pipe's time); the dropped Wlo.Xlo term is 2^-22 of the product.  Activations are scaled by 2^6 (a map value of
//   2e-3 still has a normal lo part; fp16 overflows at |v| = 1023.5, the split saturates there), the weights of a layer by the power of two
//   that puts their largest magnitude into [512, 1024); the epilogue multiplies the accumulator by 2^-(6 + s) -- exact.  Emulated on the
//   oracle (every conv of the seeded models evaluated this way, products in fp64): 1 - cos 4e-9 on the m32 models against the 1e-6 of the goldens.
//
// Map format "S16": channel-last like conv2d.hip's fp32 maps and with the same 4 bytes per channel, so pointers, leading dimensions and
// channel slices (multiples of 16) are what they are there -- but a unit of 16 channels is stored as [16 x hi | 16 x lo] fp16.  A pixel's
// 32-channel K chunk is then 128 contiguous bytes = 8 granules of 16 bytes (unit 0 hi ch 0-7, hi ch 8-15, lo ch 0-7, lo ch 8-15, unit 1 ...)
// that LDS-DMA moves without touching a register, and lane group q of an MFMA operand reads granule (q >> 1) * 4 + (q & 1) for its hi half
// and the granule two further for lo.  Weights are packed the same way per (output channel, tap): [cout16][taps][2 * nchunks units][32].
//
// Kernel: persistent workgroups; a unit of work = (utterance, pixel tile of 8 segments of 16 output pixels) of one tile of CT <= 12 blocks of 16
// output channels.
//   3x3: the segments are R <= 8 consecutive rows of one 16-column strip, so the input patch is (R - 1) * stride + 3 rows of 15 * stride + 3
//        (padded to a multiple of 8) columns -- 1.4 x the output pixels where a flat list of segments reads 3.4 x;
//   1x1: 128 consecutive pixels of the flattened utterance plane (no column padding of narrow maps).
//   One or more PRODUCER waves keep the patches of the next K stages (3x3: one 32-channel chunk each; 1x1: three) travelling into a ring
//   of LDS slots (global_load_lds, zero padding and the channel concatenation of AFF = the source address of a granule); they wait for
//   their own transfers and meet the consumers at one barrier per stage.  The CONSUMER waves split the output channels (NBW blocks each) and
//   share the pixels: every wave reads the B fragments (pixels) of all 8 segments from LDS -- bank-conflict free through the granule ^ (entry & 7)
//   swizzle applied on the source side -- and its own A fragments (weights) straight from global memory / L2, two steps ahead in three rotating
//   register sets, so no weight byte is fetched twice by a workgroup (layers with few output channels: two groups of consumer waves also split the
//   pixels).  Accumulators: segments x NBW x 4 registers.
//   Epilogue (in the scaled domain of the stored values): scale, bias, clamp [+ residual] | SiLU | AFF mix, optionally a second output
//   y2 = y + add  (the "sp + spx[i]" input of the next 3x3 conv of a Res2Net block, eres2net.py:92, so that its loader stays a plain copy); operands
//   requested per batch of segments, 16 bytes per lane in both directions (v_permlane16_swap trades the hi / lo halves between lane rows).
//   Launch shapes (cs_plan below: blocks per wave, wave roles, rows per tile, ring depth, workgroups per CU) follow the layer; none changes a bit of
//   the result -- every output element is the same K-ordered sum in every shape.
#include <vector>

#include "kernels.h"
#include "s16map.h"

namespace mv {

constexpr int CS_SEGS = 8;             // 16-pixel segments per workgroup
constexpr int CS_P_MAX = 32;           // LDS-DMA transfers of one stage a producer wave may own
constexpr int CS_CHUNK1_BYTES = CS_SEGS * 16 * 128;  // 1x1: one 32-channel chunk of the 128 pixels
constexpr int CS_KCH1 = 3;                           // 1x1: chunks per stage (three steps: the weight register sets rotate with period 3)

__device__ __attribute__((aligned(256))) const unsigned char g_cs_zero_page[256] = {0};

struct Conv2dsArgs {
    const half_t* x;
    const half_t* x2;    // second source of the channel concatenation (units >= cin1u)
    const half_t* w;     // [cout16][taps][wunits][32]
    const float* bias;   // [cout16]
    const half_t* res;   // epi 0: optional residual; epi 2: first AFF operand
    const half_t* res2;  // epi 2: second AFF operand
    const half_t* add;   // optional: y2 = y + add
    half_t* y;
    half_t* y2;
    unsigned* peak;      // optional: largest |stored value * 64| before the clamp, as float bits (s16map.h)
    int64_t ldx, ldx2, ldres, ldres2, ldadd, ldy, ldy2;  // channels (= 4-byte elements) between pixels
    int cin1u, cinu, nchunks, wunits, cout16;
    int B, H, W, Ho, Wo, sh, sw, epi;
    float lo, hi, oscale;
    int R, ncs, tiles, CT, ncons, nprod;   // rows per 3x3 tile, column strips, pixel tiles per utterance, blocks per channel tile, wave roles
    int pc, pcv, pc_magic;                  // 3x3: allocated / valid patch columns, 65536 / pc rounded up
    int ns, pp, wg_per_ct;                  // ring stages, transfers per producer wave and stage (padded), workgroups per channel tile
};

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
#define MV_CS_W(N) case N: wait_vm<N>(); break;
        MV_CS_W(0) MV_CS_W(1) MV_CS_W(2) MV_CS_W(3) MV_CS_W(4) MV_CS_W(5) MV_CS_W(6) MV_CS_W(7) MV_CS_W(8) MV_CS_W(9) MV_CS_W(10) MV_CS_W(11)
        MV_CS_W(12) MV_CS_W(13) MV_CS_W(14) MV_CS_W(15) MV_CS_W(16) MV_CS_W(17) MV_CS_W(18) MV_CS_W(19) MV_CS_W(20) MV_CS_W(21) MV_CS_W(22)
        MV_CS_W(23) MV_CS_W(24) MV_CS_W(25) MV_CS_W(26) MV_CS_W(27) MV_CS_W(28) MV_CS_W(29) MV_CS_W(30) MV_CS_W(31) MV_CS_W(32) MV_CS_W(33)
        MV_CS_W(34) MV_CS_W(35) MV_CS_W(36) MV_CS_W(37) MV_CS_W(38) MV_CS_W(39) MV_CS_W(40) MV_CS_W(41) MV_CS_W(42) MV_CS_W(43) MV_CS_W(44)
        MV_CS_W(45) MV_CS_W(46) MV_CS_W(47) MV_CS_W(48)
#undef MV_CS_W
        default: wait_vm<0>(); break;
    }
}

// Persistent workgroups: workgroup (channel tile ct, index widx) walks the pixel tiles widx, widx + wg_per_ct, ... of all utterances; its sequence of
// K stages (tiles x stages per tile) runs through ONE ring of `ns` LDS slots, so the producers are ns - 1 stages ahead across tile boundaries -- the
// patch of the next tile travels while the consumers finish this one and store it.  One barrier per stage:
//   producer:  [issue stages 0 .. ns-2]   for g: wait until stage g has landed (the younger stages may stay in flight: counted wait, every producer
//              wave issues exactly pp transfers per stage -- the padding ones read the zero page into a dump KiB); barrier g; issue stage g + ns - 1
//              into the slot of stage g - 1, which every consumer has left when it arrives at barrier g
//   consumer:  for g: barrier g; MFMAs of stage g (weights two steps ahead in registers); after a tile's last stage its epilogue
// SPW = segments per consumer wave: 8 (the waves split the output channels and share all pixels) or, for layers with few output channels, 4 / 2 / 1
// (consumer wave w works on segments (w % PG) * SPW ..., PG = 8 / SPW, of channel group w / PG: more waves on the matrix pipe where one or two
// blocks of 16 channels would leave three SIMDs idle; the small weight slices are then read by PG waves)
// TRACK: the instantiation that reports the largest stored value to a.peak (MvConv2dsDesc.peak; the CAM++ exact head).  A template parameter, not a
// branch: the running maximum and its operands cost 9-18 VGPRs in every shape (r14c: the 54.9 M ERes2NetV2 692 -> 517 utt/s with the tracking code
// compiled into the one kernel, although its branch was never taken), so launches without a peak word run the kernel without that code.
template <int KS, int NBW, int SPW, int MAXT, bool TRACK>
__global__ __launch_bounds__(MAXT) void conv2ds_kernel(Conv2dsArgs a) {
    MV_DYN_SMEM(smem);
    constexpr int TAPS = KS * KS;
    constexpr int KCH = KS == 3 ? 1 : CS_KCH1;            // K chunks per stage
    constexpr int G0 = NBW >= 2 ? 2 : 4;                  // segments per MFMA group: >= 4 independent accumulators between dependent MFMAs
    constexpr int G = G0 < SPW ? G0 : SPW;
    constexpr int PG = CS_SEGS / SPW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = MV_UNIFORM(tid >> 6);
    const int ct = blockIdx.x / a.wg_per_ct, widx = blockIdx.x - ct * a.wg_per_ct;
    const int nst = (a.nchunks + KCH - 1) / KCH;
    const int ptiles = a.B * a.tiles;                                                  // pixel tiles of the launch
    const int my_tiles = widx < ptiles ? (ptiles - widx + a.wg_per_ct - 1) / a.wg_per_ct : 0;
    const int nstages = my_tiles * nst;
    const int chunk_bytes = KS == 3 ? ((a.R - 1) * a.sh + 3) * a.pc * 128 : CS_CHUNK1_BYTES;
    const int stage_bytes = chunk_bytes * KCH;
    const unsigned lds0 = lds_addr(smem);
    const int HWo = a.Ho * a.Wo;

    if (wave >= a.ncons) {
        // ---------------- producer ----------------
        // Rolled loops over the transfers of a stage with scalar bookkeeping: source = base of (utterance, source tensor, chunk, this lane's
        // granule) + pixel * bytes per pixel -- one 64-bit multiply-add -- or the zero page.  (The first form, an unrolled loop over a register
        // array of pixels, compiled to ~50 instructions per transfer with scalar registers spilled into lanes: the producers, not the matrix
        // pipe, set the pace of the 1x1 layers, r12t timeline.)
        const int pw = wave - a.ncons;
        const int psel = lane >> 3, slot = lane & 7;
        const int g = slot ^ psel;                  // granule of the 128-byte chunk row this lane fetches (entry & 7 == psel: rows are 8k entries)
        const int usel = g >> 2;                    // unit of the chunk
        const unsigned lane_off = (unsigned)((g & 3) * 16);   // bytes inside the unit
        const int ni = stage_bytes >> 10;           // transfers per stage
        const int nprod = a.nprod, pp = a.pp, ns = a.ns;
        const unsigned dump = lds0 + (unsigned)(ns * stage_bytes + pw * 1024);
        const char* zero = reinterpret_cast<const char*>(g_cs_zero_page);
        const int cpr = KS == 3 ? a.pc >> 3 : 16;   // transfers per patch row | per chunk
        int ti = 0, ci = 0, slot_i = 0;
        int b = 0;
        int h0 = 0, wlane = 0;                      // 3x3: input row of patch row 0; this lane's input column of column group 0
        unsigned cmask = 0;                         // 3x3: column groups in which this lane's column exists (inside the patch and the map)
        int p0 = 0;                                 // 1x1: first pixel of the tile
        auto enter_tile = [&](int tile_index) __attribute__((always_inline)) {
            const int pt = widx + tile_index * a.wg_per_ct;
            b = pt / a.tiles;
            const int t = pt - b * a.tiles;
            if (KS == 3) {
                const int rt = t / a.ncs;
                h0 = rt * a.R * a.sh - 1;
                const int w0 = (t - rt * a.ncs) * 16 * a.sw - 1;
                wlane = w0 + psel;
                cmask = 0;
                for (int cg = 0; cg < cpr; ++cg) {
                    const int col = cg * 8 + psel, wi = w0 + col;
                    if (col < a.pcv && wi >= 0 && wi < a.W) cmask |= 1u << cg;
                }
            } else {
                p0 = t * (CS_SEGS * 16);
            }
        };
        enter_tile(0);
        auto issue_stage = [&]() __attribute__((always_inline)) {
            const unsigned base = lds0 + (unsigned)(slot_i * stage_bytes);
            if (++slot_i == ns) slot_i = 0;
            // this lane's source per chunk of the stage: address of pixel 0 (its unit and granule included) and bytes per pixel; 0 = nothing
            // (a unit behind the last one)
            uint64_t sb[KCH];
            unsigned ldb[KCH];
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                const int u = 2 * (ci * KCH + kc) + usel;
                sb[kc] = 0;
                ldb[kc] = 0;
                if (u < a.cin1u) {
                    sb[kc] = (uint64_t)(reinterpret_cast<const char*>(a.x) + ((int64_t)b * a.H * a.W * a.ldx + 16 * u) * 4 + lane_off);
                    ldb[kc] = (unsigned)(a.ldx * 4);
                } else if (u < a.cinu) {
                    sb[kc] = (uint64_t)(reinterpret_cast<const char*>(a.x2) + ((int64_t)b * a.H * a.W * a.ldx2 + 16 * (u - a.cin1u)) * 4 + lane_off);
                    ldb[kc] = (unsigned)(a.ldx2 * 4);
                }
            }
            int count = 0;
            if (KS == 3) {
                int pr = pw / cpr, cg = pw - pr * cpr;   // patch row and column group of transfer k
#pragma unroll 1
                for (int k = pw; k < ni; k += nprod) {
                    const int hi = h0 + pr;
                    const bool ok = (unsigned)hi < (unsigned)a.H && ((cmask >> cg) & 1u) != 0 && sb[0] != 0;
                    const unsigned pix = (unsigned)(hi * a.W + wlane + cg * 8);   // (meaningless where !ok)
                    const uint64_t src = ok ? sb[0] + (uint64_t)pix * ldb[0] : (uint64_t)zero;
                    glds16_untracked(reinterpret_cast<const void*>(src), base + (unsigned)(k * 1024));
                    ++count;
                    cg += nprod;
                    while (cg >= cpr) {
                        cg -= cpr;
                        ++pr;
                    }
                }
            } else {
#pragma unroll 1
                for (int k = pw; k < ni; k += nprod) {
                    const int kc = k >> 4, kk = k & 15;   // chunk of the stage, transfer of the chunk (the chunks cover the same pixels)
                    uint64_t cb = sb[0];
                    unsigned cl = ldb[0];
#pragma unroll
                    for (int z = 1; z < KCH; ++z)
                        if (kc == z) {  // uniform
                            cb = sb[z];
                            cl = ldb[z];
                        }
                    const int p = p0 + kk * 8 + psel;
                    unsigned pix = (unsigned)p;
                    if (a.sh != 1 || a.sw != 1) {  // uniform: strided 1x1 (first block of a stage)
                        const int ho = p / a.Wo, wo = p - ho * a.Wo;
                        pix = (unsigned)(ho * a.sh * a.W + wo * a.sw);
                    }
                    const bool ok = p < HWo && cb != 0;
                    const uint64_t src = ok ? cb + (uint64_t)pix * cl : (uint64_t)zero;
                    glds16_untracked(reinterpret_cast<const void*>(src), base + (unsigned)(k * 1024));
                    ++count;
                }
            }
#pragma unroll 1
            for (; count < pp; ++count) glds16_untracked(zero, dump);   // every producer wave issues exactly pp transfers per stage (counted waits)
            if (++ci == nst) {
                ci = 0;
                ++ti;
                if (ti < my_tiles) enter_tile(ti);
            }
        };
        int issued = 0;
        for (; issued < ns - 1 && issued < nstages; ++issued) issue_stage();
#pragma unroll 1
        for (int gs = 0; gs < nstages; ++gs) {
            wait_vm_dyn((issued - gs - 1) * pp);
            lds_barrier();
            if (issued < nstages) {
                issue_stage();
                ++issued;
            }
        }
        return;
    }

    // ---------------- consumer ----------------
    const int j16 = lane & 15, q = lane >> 4;
    const int ghi = (q >> 1) * 4 + (q & 1);
    const int nblk_total = a.cout16 >> 4;
    const int cgrp = wave / PG, useg0 = (wave - cgrp * PG) * SPW;   // channel group and first segment of this wave
    // the blocks of this channel tile, spread evenly over its channel groups (10 blocks on 4 waves: 3, 3, 2, 2)
    const int ncg = a.ncons / PG;
    int tile_blocks = nblk_total - ct * a.CT;
    tile_blocks = tile_blocks < a.CT ? tile_blocks : a.CT;
    const int base_nb = tile_blocks / ncg, rem_nb = tile_blocks - base_nb * ncg;
    const int blk0 = ct * a.CT + cgrp * base_nb + (cgrp < rem_nb ? cgrp : rem_nb);   // this wave's first block of 16 output channels
    const int nb = MV_UNIFORM(base_nb + (cgrp < rem_nb ? 1 : 0));                     // ... and how many (<= NBW)

    // A fragments: lane (row j16 of a block, K group q) reads 16 bytes of hi and the 16 bytes 32 further of lo
    const int64_t wrow_halves = (int64_t)TAPS * a.wunits * 32;
    const half_t* wrow[NBW];
    float4v bias64[NBW];   // 64 * bias of this lane's four channels per block: the epilogue works on the stored (scaled) values
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int blk = i < nb ? blk0 + i : (blk0 < nblk_total ? blk0 : 0);   // clamped: loaded, never used
        wrow[i] = a.w + ((int64_t)blk * 16 + j16) * wrow_halves + ghi * 8;
        const float4v bv = *reinterpret_cast<const float4v*>(a.bias + blk * 16 + q * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias64[i][r] = bv[r] * CS_XSCALE;
    }
    const float osc64 = a.oscale * CS_XSCALE;
    const float lo64 = fminf(fmaxf(a.lo * CS_XSCALE, -65504.0f), 65504.0f), hi64 = fminf(fmaxf(a.hi * CS_XSCALE, -65504.0f), 65504.0f);
    constexpr bool track = TRACK;
    float pk = 0.0f;
    // step j of stage c of a tile: 3x3 -> tap j of chunk c; 1x1 -> chunk c * KCH + j.  The weights do not depend on the pixel tile.
    constexpr int SPS = KS == 3 ? TAPS : KCH;
    static_assert(SPS % 3 == 0, "the weight register sets rotate with period 3");
    auto a_offset = [&](int c, int j) __attribute__((always_inline)) -> int64_t {
        if (KS == 3) return ((int64_t)j * a.wunits + 2 * c) * 32;
        const int ch = c * KCH + j;
        return (int64_t)(ch < a.nchunks ? ch : a.nchunks - 1) * 64;
    };
    // three register sets of weights: the set of step s is s % 3 (a stage has 9 or 3 steps, so it always starts on set 0 and the roles are
    // compile-time inside it); the request for step s + 2 goes into the set step s - 1 has left.  (Two sets with three blocks per wave -- a step
    // of 72 MFMAs covers an L2 round trip -- need the step sequence twice, once per stage parity: that form spilled 300 - 500 registers.)
    constexpr int NSET = 3;
    half8v wh[NSET][NBW], wl[NSET][NBW];
    int pc_c = 0, pc_j = 0;   // (stage, step) the next weight request is for
    auto request = [&](int set) __attribute__((always_inline)) {
        const int64_t off = a_offset(pc_c, pc_j);
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            wh[set][i] = *reinterpret_cast<const half8v*>(wrow[i] + off);
            wl[set][i] = *reinterpret_cast<const half8v*>(wrow[i] + off + 16);
        }
        if (++pc_j == SPS) {
            pc_j = 0;
            if (++pc_c == nst) pc_c = 0;
        }
    };
    request(0);
    request(1);
    // B fragments: entry e of the patch at e * 128, granule g at ((g ^ (e & 7)) << 4); rows of a 3x3 patch are a multiple of 8 entries
    // apart, so segment u is a constant further than segment 0
    const int seg_stride = KS == 3 ? a.sh * a.pc * 128 : 16 * 128;
    // 3x3: the patch holds the rows of R segments.  A group of G segments that starts inside the tile may reach behind them (R = 5 in groups of
    // two: segment 5) -- such a segment is never stored, and its fragment reads are pointed at the last one that exists, so that no read leaves
    // the stage (in the last ring slot: the workgroup's LDS).  Relative to this wave's first segment; scalar.
    const int seg_last = MV_UNIFORM((KS == 3 ? a.R : CS_SEGS) - 1 - useg0);

    float4v acc[SPW][NBW];
    int c = 0, tile_index = 0, slot_c = 0;
    int b = 0, ho0 = 0, wo0 = 0, p0 = 0, nvalid = 0;
#pragma unroll 1
    for (int gs = 0; gs < nstages; ++gs) {
        if (c == 0) {
            // ---- a new tile ----
            const int pt = widx + tile_index * a.wg_per_ct;
            b = pt / a.tiles;
            const int t = pt - b * a.tiles;
            if (KS == 3) {
                const int rt = t / a.ncs;
                ho0 = rt * a.R;
                wo0 = (t - rt * a.ncs) * 16;
                const int rows = a.Ho - ho0;
                nvalid = rows < a.R ? rows : a.R;
            } else {
                p0 = t * (CS_SEGS * 16);
                const int left = HWo - p0;
                nvalid = left >= CS_SEGS * 16 ? CS_SEGS : (left + 15) >> 4;
            }
            nvalid = MV_UNIFORM(nvalid);
#pragma unroll
            for (int u = 0; u < SPW; ++u)
#pragma unroll
                for (int i = 0; i < NBW; ++i) acc[u][i] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
        lds_barrier();  // stage gs has landed; every wave has left the slot the producers refill next
        const char* buf = smem + slot_c * stage_bytes;
        if (++slot_c == a.ns) slot_c = 0;
        // one step: the MFMAs of (tap | chunk) j on weight set SET
        auto step = [&](int j, const half8v (&ah)[NBW], const half8v (&al)[NBW]) __attribute__((always_inline)) {
            const bool live = KS == 3 || c * KCH + j < a.nchunks;  // uniform: the chunk behind the last one of an odd count does not exist
            if (live && nb > 0) {
                int e0;
                const char* cb;
                if (KS == 3) {
                    const int kh = j / 3, kw = j - kh * 3;
                    e0 = kh * a.pc + j16 * a.sw + kw;
                    cb = buf;
                } else {
                    e0 = j16;
                    cb = buf + j * chunk_bytes;
                }
                const char* ph = cb + e0 * 128 + ((ghi ^ (e0 & 7)) << 4) + useg0 * seg_stride;
                const char* pl = cb + e0 * 128 + (((ghi + 2) ^ (e0 & 7)) << 4) + useg0 * seg_stride;
                // groups of G segments.  Every group waits for its own fragment reads (the uniform branches on the segment count end basic blocks):
                // the MFMA phase of a stage runs at 50 - 60 % of the issue rate (r12w / r12y timelines).  Requesting the reads of group g + 1 before
                // the MFMAs of group g measured slower (r12x: 3x3 160 -> 160 at 16 x 20 x 75 81 -> 89 us), and so did a branch-free copy of this loop
                // for full tiles that the scheduler interleaves by itself (r12ac, same call: 3x3 80 -> 80 75.5 -> 87 us, 16 -> 16 29 -> 32).
                constexpr int NG = SPW / G;
                half8v bh[G], bl[G];
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int u0 = gi * G;
                    if (useg0 + u0 < nvalid) {  // uniform
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            const int su = KS == 3 && u0 + u > seg_last ? seg_last : u0 + u;   // uniform
                            bh[u] = *reinterpret_cast<const half8v*>(ph + su * seg_stride);
                            bl[u] = *reinterpret_cast<const half8v*>(pl + su * seg_stride);
                        }
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
#pragma unroll
                            for (int u = 0; u < G; ++u) acc[u0 + u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[u], acc[u0 + u][i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
#pragma unroll
                            for (int u = 0; u < G; ++u) acc[u0 + u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[u], acc[u0 + u][i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
#pragma unroll
                            for (int u = 0; u < G; ++u) acc[u0 + u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[u], acc[u0 + u][i], 0, 0, 0);
                    }
                }
            }
        };
        // sets 0, 1, 2 in turn: a stage starts on set 0 because its step count (9 taps | 3 chunks) is a multiple of 3
#pragma unroll 1
        for (int jj = 0; jj < SPS; jj += 3) {
            request(2);
            step(jj, wh[0], wl[0]);
            request(0);
            step(jj + 1, wh[1], wl[1]);
            request(1);
            step(jj + 2, wh[2], wl[2]);
        }
        if (++c < nst) continue;
        c = 0;
        ++tile_index;

        // ---------------- epilogue of the tile: D[channel 4q + r][pixel j16], in the scaled domain X = 64 * value ----------------
        // Batches of EB segments: all operand loads of a batch are requested before the first is used (clamped addresses instead of branches:
        // only the stores are predicated), so a tile pays the memory latency once per batch.
        constexpr int EB0 = NBW <= 2 ? 4 : 1;   // (three blocks per wave: 253 registers with one segment's operands in flight)
        constexpr int EB = EB0 < SPW ? EB0 : SPW;
        // Layers with a residual and nothing else (every conv3: scale, bias, + residual, clamp) in batches of TWO segments with their own per-unit code: a
        // batch's wait for its residual loads is an s_waitcnt vmcnt(0) that also waits for the previous batch's stores, so a tile pays one store round trip per
        // batch -- eight per tile with one-segment batches (the general code has no registers for more at three blocks per wave; this lean copy has).
        constexpr int EBR0 = NBW <= 2 ? 4 : 2;
        constexpr int EBR = EBR0 < SPW ? EBR0 : SPW;
        const bool res_only = a.epi == 0 && a.res != nullptr && a.y2 == nullptr;   // uniform
        if (res_only) {
#pragma unroll
            for (int u0 = 0; u0 < SPW; u0 += EBR) {
                if (useg0 + u0 >= nvalid) break;  // uniform
                int64_t pixo[EBR];
                bool ok[EBR];
#pragma unroll
                for (int u = 0; u < EBR; ++u) {
                    const int su = useg0 + u0 + u;
                    const int uu = su < nvalid ? su : nvalid - 1;
                    if (KS == 3) {
                        const int wo = wo0 + j16;
                        ok[u] = su < nvalid && wo < a.Wo;
                        pixo[u] = (int64_t)b * HWo + (int64_t)(ho0 + uu) * a.Wo + (wo < a.Wo ? wo : a.Wo - 1);
                    } else {
                        const int p = p0 + uu * 16 + j16;
                        ok[u] = su < nvalid && p < HWo;
                        pixo[u] = (int64_t)b * HWo + (p < HWo ? p : HWo - 1);
                    }
                }
                const int64_t qoff = (q & 1) * 16 + (q >> 1) * 8;
                uint4v rr[EBR][NBW];
#pragma unroll
                for (int u = 0; u < EBR; ++u)
#pragma unroll
                    for (int i = 0; i < NBW; ++i)
                        rr[u][i] = *MV_GLOBAL_PTR(uint4v, a.res + pixo[u] * a.ldres * 2 + (int64_t)(i < nb ? blk0 + i : blk0) * 32 + qoff);
#pragma unroll
                for (int u = 0; u < EBR; ++u)
#pragma unroll
                    for (int i = 0; i < NBW; ++i) {
                        if (i >= nb) break;  // uniform
                        const float4v o1 = s16_unswap4(rr[u][i]);
                        float4v X;
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = acc[u0 + u][i][r] * osc64 + bias64[i][r] + o1[r];
                        if (track && ok[u]) pk = s16_peak_of(pk, float4v{fmaxf(X[0], a.lo * CS_XSCALE), fmaxf(X[1], a.lo * CS_XSCALE), fmaxf(X[2], a.lo * CS_XSCALE), fmaxf(X[3], a.lo * CS_XSCALE)});
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r], lo64, hi64);
                        const uint4v w1 = s16_swap4(X);
                        if (ok[u]) *reinterpret_cast<uint4v*>(a.y + pixo[u] * a.ldy * 2 + (int64_t)(blk0 + i) * 32 + qoff) = w1;
                    }
            }
        }
#pragma unroll
        for (int u0 = 0; u0 < SPW && !res_only; u0 += EB) {
            if (useg0 + u0 >= nvalid) break;  // uniform
            int64_t pixo[EB];
            bool ok[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int su = useg0 + u0 + u;                           // segment of the tile
                const int uu = su < nvalid ? su : nvalid - 1;           // clamped row: loaded, not stored
                if (KS == 3) {
                    const int wo = wo0 + j16;
                    ok[u] = su < nvalid && wo < a.Wo;
                    pixo[u] = (int64_t)b * HWo + (int64_t)(ho0 + uu) * a.Wo + (wo < a.Wo ? wo : a.Wo - 1);
                } else {
                    const int p = p0 + uu * 16 + j16;
                    ok[u] = su < nvalid && p < HWo;
                    pixo[u] = (int64_t)b * HWo + (p < HWo ? p : HWo - 1);
                }
            }
            // A lane's four channels are 8 bytes of hi and 8 bytes of lo, 32 bytes apart.  Memory sees 16 bytes per lane instead: lane group q = 0 / 1 / 2 / 3
            // moves the hi halves of channels 0-7, the lo halves of 0-7, hi of 8-15, lo of 8-15 of the unit, and v_permlane16_swap trades the
            // odd / even 16-lane rows of the register pairs on the way (the same exchange in both directions): 64 contiguous bytes per pixel
            // and instruction instead of two times 32.
            const int64_t qoff = (q & 1) * 16 + (q >> 1) * 8;   // halves inside the unit: (q & 1) * 32 + (q >> 1) * 16 bytes
            const bool has1 = a.res != nullptr, has2 = a.epi == 2, has3 = a.y2 != nullptr;   // uniform
            if (a.epi == 0 && !has1 && !has3) {
                // Plain layers (scale, bias, clamp -- every conv1 and every 3 x 3 of the family): their own copy of the per-unit code.  Through the general
                // code below a unit is ~100 instructions and half a dozen uniform branches (epilogue mode, operands, second output) for four fma, four clamps,
                // the hi / lo split and one store; on the big maps a tile's 24 units were 3/4 of a consumer wave's time (r14z timeline).
#pragma unroll
                for (int u = 0; u < EB; ++u)
#pragma unroll
                    for (int i = 0; i < NBW; ++i) {
                        if (i >= nb) break;  // uniform
                        float4v X;
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = acc[u0 + u][i][r] * osc64 + bias64[i][r];
                        if (track && ok[u]) pk = s16_peak_of(pk, float4v{fmaxf(X[0], a.lo * CS_XSCALE), fmaxf(X[1], a.lo * CS_XSCALE), fmaxf(X[2], a.lo * CS_XSCALE), fmaxf(X[3], a.lo * CS_XSCALE)});
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r], lo64, hi64);
                        const uint4v w1 = s16_swap4(X);
                        if (ok[u]) *reinterpret_cast<uint4v*>(a.y + pixo[u] * a.ldy * 2 + (int64_t)(blk0 + i) * 32 + qoff) = w1;
                    }
                continue;
            }
            uint4v r1[EB][NBW], r2[EB][NBW];
#pragma unroll
            for (int u = 0; u < EB; ++u)
#pragma unroll
                for (int i = 0; i < NBW; ++i) r1[u][i] = r2[u][i] = uint4v{0u, 0u, 0u, 0u};
            // Layers WITH operands to read back (residual, AFF inputs, the addend of the second output): ONE block that requests them and, before it ends,
            // uses them on every path (MV_OPAQUE) -- loaded under `if (has1)` and read under another `if (has1)`, the compiler's wait-count insertion sees the
            // path on which a load is never waited for, carries its registers as pending around the batch loop and protects each load of the NEXT batch
            // (same registers) with its own s_waitcnt vmcnt(0): six memory round trips one behind the other per batch (r14o, tools/isa_audit.py).  Layers
            // WITHOUT operands never enter the block: its wait is an s_waitcnt vmcnt(0), which also waits for the previous batch's STORES -- executed by
            // every layer it cost the plain 1x1 layers of the big maps 8 x 1.2 us per tile (r14z timeline: epilogue 9.9 us of a 13.5 us tile).
            if (has1 || has2 || has3) {
#pragma unroll
                for (int u = 0; u < EB; ++u)
#pragma unroll
                    for (int i = 0; i < NBW; ++i) {
                        const int64_t coff = (int64_t)(i < nb ? blk0 + i : blk0) * 32 + qoff;   // halves inside a pixel
                        if (has1) r1[u][i] = *MV_GLOBAL_PTR(uint4v, a.res + pixo[u] * a.ldres * 2 + coff);
                        if (has2 || has3)
                            r2[u][i] = *MV_GLOBAL_PTR(uint4v, has2 ? a.res2 + pixo[u] * a.ldres2 * 2 + coff : a.add + pixo[u] * a.ldadd * 2 + coff);
                    }
#pragma unroll
                for (int u = 0; u < EB; ++u)
#pragma unroll
                    for (int i = 0; i < NBW; ++i) {
                        MV_OPAQUE(r1[u][i]);
                        MV_OPAQUE(r2[u][i]);
                    }
            }
#pragma unroll
            for (int u = 0; u < EB; ++u)
#pragma unroll
                for (int i = 0; i < NBW; ++i) {
                    if (i >= nb) break;  // uniform
                    const int64_t coff = (int64_t)(blk0 + i) * 32 + qoff;
                    float4v o1 = {0.0f, 0.0f, 0.0f, 0.0f}, o2 = {0.0f, 0.0f, 0.0f, 0.0f};   // the operands' values (scaled) of this lane's four channels
                    if (has1) o1 = s16_unswap4(r1[u][i]);
                    if (has2 || has3) o2 = s16_unswap4(r2[u][i]);
                    float4v X;
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[r] = acc[u0 + u][i][r] * osc64 + bias64[i][r];
                    if (a.epi == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] += o1[r];
                        if (track && ok[u]) pk = s16_peak_of(pk, float4v{fmaxf(X[0], a.lo * CS_XSCALE), fmaxf(X[1], a.lo * CS_XSCALE), fmaxf(X[2], a.lo * CS_XSCALE), fmaxf(X[3], a.lo * CS_XSCALE)});
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r], lo64, hi64);
                    } else if (a.epi == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = X[r] * CS_XSCALE_INV;
                            X[r] = v / (1.0f + expf(-v)) * CS_XSCALE;
                        }
                        if (track && ok[u]) pk = s16_peak_of(pk, X);
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float th = tanhf(X[r] * CS_XSCALE_INV);  // x_att = 1 + th;  out = x * x_att + y * (2 - x_att)
                            X[r] = o1[r] * (1.0f + th) + o2[r] * (1.0f - th);
                        }
                        if (track && ok[u]) pk = s16_peak_of(pk, X);
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r]);
                    }
                    const uint4v w1 = s16_swap4(X);
                    if (ok[u]) *reinterpret_cast<uint4v*>(a.y + pixo[u] * a.ldy * 2 + coff) = w1;
                    if (has3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] += o2[r];
                        if (track && ok[u]) pk = s16_peak_of(pk, X);
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = s16_clamp(X[r]);
                        const uint4v w2 = s16_swap4(X);
                        if (ok[u]) *reinterpret_cast<uint4v*>(a.y2 + pixo[u] * a.ldy2 * 2 + coff) = w2;
                    }
                }
        }
    }
    if (track) s16_peak_commit(a.peak, pk);   // (every lane of the consumer wave is here)
}

// ---- launch ------------------------------------------------------------------------------------------------------------------------------
namespace {

struct CsPlan {
    int nbw, spw, CT, ctiles, ncons, nprod, R, ncs, tiles, pc, pcv, ns, pp, wg_per_ct, wgs_per_cu;
    size_t lds;
};

// Rows per 3x3 tile, by a cost model of one launch: rounds of tiles over the workgroup slots x cycles per tile.  A tile costs its K steps (the MFMAs of
// the segments the wave computes -- whole groups of G -- at ~60 % of the issue rate plus ~300 cycles per step of reads, waits and requests), an
// epilogue per row and a fixed part; few tall tiles leave CUs idle on small launches (16 x 20 x 75 at 160 channels: tiles of 4 rows 65 us, of 5 rows
// 81 us, r12u), many flat ones pay the per-step overhead more often (16 x 10 x 38 at 320 channels: 4 rows 115 us, 5 rows 78 us).  Never changes bits.
int cs_rows(int Ho, int stride, int64_t strips, int ctiles, int slots, int steps_per_tile, int nbw, int spw) {
    const int cap = stride == 2 ? 5 : CS_SEGS;   // (stride 2: 11 patch rows of 40 columns = 55 KiB per stage, a ring of two)
    const int g = nbw >= 2 ? 2 : (spw < 4 ? spw : 4);
    int best = 1;
    double best_cost = 0.0;
    for (int r = cap < Ho ? cap : Ho; r >= 1; --r) {
        const int segs = (int)round_up((int64_t)ceil_div(r, CS_SEGS / spw), g);   // segments a consumer wave computes per step
        const double step = segs * nbw * 3 * 16 / 0.6 + 300.0;
        const double tile = steps_per_tile * step + r * nbw * 400.0 + 3000.0;
        const int64_t tiles = ceil_div(Ho, r) * strips * ctiles;
        const double cost = (double)ceil_div(tiles, slots) * tile;
        if (best_cost == 0.0 || cost < best_cost * 0.9) {   // (ties and near-ties: the taller tile -- less patch overlap, and the model is coarse)
            best_cost = cost;
            best = r;
        }
    }
    return best;
}

int cs_plan(const MvConv2dsDesc& d, int Ho, int Wo, int sw, CsPlan* p) {
    const int nblk = d.cout16 / 16;
    // Wave roles.  Up to 7 blocks: one block per consumer wave (<= 168 registers, 12 waves per CU).  From 8 blocks on: FOUR consumer waves -- one per
    // SIMD, so no two of them share a matrix pipe (five or six waves leave one SIMD with twice the work: the barrier of a stage waits for it, r12w
    // timeline) -- with the blocks of a channel tile of <= 12 spread evenly over them (2 or 3 per wave, <= 256 registers, 8 waves per CU), and four
    // producer waves: one transfer costs a producer wave ~270 cycles, so two of them bound a 1x1 stage before the matrix pipe does (r12x timeline).
    // ... except where the epilogue dominates: a layer with operands to read back (residual, second output, AFF) and few K stages per tile keeps
    // two blocks per wave on up to six consumer waves (four segments' operands in flight per batch instead of one; such layers are HBM-bound and a
    // second channel tile would read the input twice).
    int nbw = d.nbw_hint, CT = d.ct_hint;
    const int stages_per_tile = (int)ceil_div(ceil_div(d.cin16, 32), d.ks == 3 ? 1 : CS_KCH1);
    const bool epilogue_bound = (d.res != nullptr || d.y2 != nullptr || d.epi == 2) && stages_per_tile <= 4;
    const bool operands = d.res != nullptr || d.y2 != nullptr || d.epi == 2;
    if (nbw == 0) {
        if (nblk == 5 && d.ks == 3) {
            nbw = 2;   // (r12ag, 3x3 80 -> 80 at 16 x 40 x 149: three consumer waves of 2 + 2 + 1 blocks 66.5 us, five of one block 75)
        } else if (nblk == 2 && d.ks == 1 && !operands) {
            nbw = 2;   // (r12ag, 1x1 64 -> 32 at 16 x 80 x 298: both blocks in each of four waves of two segments 33.8 us, one block per wave 45)
        } else if (nblk <= 7) {
            nbw = 1;
        } else if (epilogue_bound) {
            nbw = 2;
        } else {
            const int ctb = CT > 0 ? CT : (int)ceil_div(nblk, ceil_div(nblk, 12));
            nbw = (int)ceil_div(ctb, 4);
            nbw = nbw < 2 ? 2 : nbw;
        }
    }
    MV_REQUIRE(nbw >= 1 && nbw <= 3, "conv2ds: blocks per wave must be 1..3");
    const int max_waves = nbw == 1 ? 12 : 8;        // 168 / 256 registers per lane
    const bool six = nbw == 2 && (epilogue_bound || d.nprod_hint == 2);   // six consumer + two producer waves
    const int nprod_want = d.nprod_hint > 0 ? d.nprod_hint : (nbw == 1 || six ? 2 : 4);
    const int max_cons = nbw == 1 ? 8 : max_waves - nprod_want;
    if (CT == 0) {
        const int cap = max_cons * nbw;
        const int ctiles = (int)ceil_div(nblk, cap);
        CT = (int)ceil_div(nblk, ctiles);
    }
    MV_REQUIRE(CT >= 1 && ceil_div(CT, nbw) <= max_cons, "conv2ds: channel tile too wide");
    p->nbw = nbw;
    p->CT = CT;
    p->ctiles = (int)ceil_div(nblk, CT);
    const int cgroups = (int)ceil_div(CT < nblk ? CT : nblk, nbw);
    // few channel groups: two groups of consumer waves split the pixels (r12w, 16 -> 16 channels 3x3 at 16 x 80 x 298: 8 segments per wave 38.7 us,
    // 4: 28.1, 2: 34.4, 1: 62 -- one segment per wave is a chain of dependent MFMAs)
    int pg = 1;
    if (nbw == 1 && 2 * cgroups <= max_cons) pg = 2;
    if (nbw == 2 && nblk == 2 && d.ks == 1 && d.nbw_hint == 0) pg = 4;
    if (d.spw_hint > 0) {
        MV_REQUIRE((d.spw_hint == 1 || d.spw_hint == 2 || d.spw_hint == 4 || d.spw_hint == 8) && (d.spw_hint != 1 || nbw == 1) &&
                       (CS_SEGS / d.spw_hint) * cgroups <= max_cons,
                   "conv2ds: segments per wave must be 8, 4, 2 (or 1 with one block per wave), consumer waves within the workgroup's limit");
        pg = CS_SEGS / d.spw_hint;
    }
    p->spw = CS_SEGS / pg;
    p->ncons = pg * cgroups;
    int stage;
    if (d.ks == 3) {
        const int64_t strips = (int64_t)d.B * ceil_div(Wo, 16);
        const int steps_per_tile = (int)ceil_div(d.cin16, 32) * 9;
        p->R = d.rows_hint > 0 ? d.rows_hint : cs_rows(Ho, d.stride, strips, p->ctiles, device_cu_count() * (p->ncons + nprod_want <= max_waves / 2 ? 2 : 1), steps_per_tile, nbw, p->spw);
        MV_REQUIRE(p->R >= 1 && p->R <= CS_SEGS, "conv2ds: rows per tile must be 1..8");
        p->ncs = (int)ceil_div(Wo, 16);
        p->tiles = (int)ceil_div(Ho, p->R) * p->ncs;
        p->pcv = 15 * sw + 3;
        p->pc = (int)round_up(p->pcv, 8);
        stage = ((p->R - 1) * d.stride + 3) * p->pc * 128;
    } else {
        p->R = CS_SEGS;
        p->ncs = 0;
        p->tiles = (int)ceil_div((int64_t)Ho * Wo, CS_SEGS * 16);
        p->pc = p->pcv = 0;
        stage = CS_KCH1 * CS_CHUNK1_BYTES;
    }
    const int ni = stage / 1024;
    // workgroups per CU, producer waves, ring depth: two workgroups (r12o: 3x3 layers with few consumer waves gain 1.4 - 1.7 x from the second
    // workgroup's MFMAs under the first one's epilogue) while the waves and a ring of two fit twice -- with one producer wave if two do not fit --,
    // else one workgroup with the deepest ring (<= 4 stages, <= 48 transfers per wave in flight behind the awaited stage)
    const int wave_cap = nbw >= 2 ? 8 : (p->spw <= 2 ? 16 : 12);   // waves per CU at the variant's register count (<= 256 | <= 104 | <= 168 per lane)
    bool found = false;
    for (int wgs = 2; wgs >= 1 && !found; --wgs) {
        if (d.wgs_hint > 0 && wgs != d.wgs_hint) continue;
        for (int nprod = nprod_want; nprod >= 1 && !found; --nprod) {
            const int pp = (int)ceil_div(ni, nprod);
            if ((p->ncons + nprod) * wgs > wave_cap || p->ncons + nprod > max_waves || pp > CS_P_MAX) continue;
            const size_t budget = (size_t)160 * 1024 / wgs - (size_t)nprod * 1024;
            int ns = (int)(budget / stage < 4 ? budget / stage : 4);
            if (d.ring_hint > 0) ns = d.ring_hint <= ns ? d.ring_hint : 0;
            if (ns > 2 && (ns - 2) * pp > 48) ns = 2 + 48 / pp;
            if (ns < 2) continue;
            p->wgs_per_cu = wgs;
            p->nprod = nprod;
            p->pp = pp;
            p->ns = ns;
            found = true;
        }
    }
    MV_REQUIRE(found, "conv2ds: no launch shape fits (waves, LDS ring)");
    p->lds = (size_t)p->ns * stage + (size_t)p->nprod * 1024;
    const int wgs = p->wgs_per_cu;
    const int64_t ptiles = (int64_t)d.B * p->tiles;
    MV_REQUIRE(ptiles < ((int64_t)1 << 30), "conv2ds: too many pixel tiles");
    int64_t per_ct = (int64_t)device_cu_count() * wgs / p->ctiles;
    if (per_ct < 1) per_ct = 1;
    p->wg_per_ct = (int)(per_ct < ptiles ? per_ct : ptiles);
    return MV_OK;
}

template <int KS, int NBW, int SPW, int MAXT, bool TRACK>
int cs_launch_t(const Conv2dsArgs& a, const CsPlan& p, hipStream_t stream) {
    static DeviceOnce smem_set;
    int slot;
    if (device_once_pending(smem_set, &slot)) {
        if (MV_SET_MAX_SMEM((conv2ds_kernel<KS, NBW, SPW, MAXT, TRACK>), 160 * 1024) != hipSuccess) return fail(MV_ERR_HIP, "conv2ds: cannot reserve dynamic LDS");
        device_once_done(smem_set, slot);
    }
    MV_LAUNCH((conv2ds_kernel<KS, NBW, SPW, MAXT, TRACK>), ((unsigned)(p.wg_per_ct * p.ctiles), 1, 1), ((unsigned)((p.ncons + p.nprod) * 64), 1, 1), p.lds, stream, a);
    return MV_OK;
}

template <int KS, int NBW, int SPW, int MAXT>
int cs_launch(const Conv2dsArgs& a, const CsPlan& p, hipStream_t stream) {
    return a.peak != nullptr ? cs_launch_t<KS, NBW, SPW, MAXT, true>(a, p, stream) : cs_launch_t<KS, NBW, SPW, MAXT, false>(a, p, stream);
}

}  // namespace

int conv2ds_launch(const MvConv2dsDesc& d, hipStream_t stream) {
    MV_REQUIRE(d.x != nullptr && d.w != nullptr && d.bias != nullptr && d.y != nullptr, "conv2ds: null pointer");
    MV_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0, "conv2ds: empty input");
    MV_REQUIRE((d.ks == 1 || d.ks == 3) && (d.stride == 1 || d.stride == 2), "conv2ds: kernel 1 or 3, stride 1 or 2");
    MV_REQUIRE(d.stride_w == 0 || d.stride_w == 1 || d.stride_w == 2, "conv2ds: stride_w must be 0 (= stride), 1 or 2");
    const int sw = d.stride_w == 0 ? d.stride : d.stride_w;
    MV_REQUIRE(d.cin16 > 0 && d.cin16 % 16 == 0 && d.cout16 > 0 && d.cout16 % 16 == 0, "conv2ds: channels must be padded to 16");
    MV_REQUIRE(d.ldx % 16 == 0 && d.ldy % 16 == 0, "conv2ds: leading dimensions must be multiples of 16 channels");
    MV_REQUIRE(d.x2 == nullptr || (d.cin1 > 0 && d.cin1 % 16 == 0 && d.cin1 < d.cin16 && d.ldx2 % 16 == 0), "conv2ds: concat split");
    MV_REQUIRE(d.epi >= 0 && d.epi <= 2, "conv2ds: epilogue mode");
    if (d.epi == 2) MV_REQUIRE(d.res != nullptr && d.res2 != nullptr && d.ldres % 16 == 0 && d.ldres2 % 16 == 0, "conv2ds: AFF mix needs both operands");
    if (d.res != nullptr) MV_REQUIRE(d.ldres % 16 == 0, "conv2ds: residual leading dimension");
    MV_REQUIRE((d.y2 == nullptr) == (d.add == nullptr), "conv2ds: the second output needs its addend");
    if (d.y2 != nullptr) MV_REQUIRE(d.ldy2 % 16 == 0 && d.ldadd % 16 == 0, "conv2ds: second output leading dimensions");
    MV_REQUIRE(d.ldx > 0 && d.ldy > 0 && (d.x2 == nullptr || d.ldx2 > 0) && (d.res == nullptr || d.ldres > 0) && (d.res2 == nullptr || d.ldres2 > 0) &&
                   (d.y2 == nullptr || (d.ldy2 > 0 && d.ldadd > 0)),
               "conv2ds: leading dimensions must be positive");
    MV_REQUIRE(d.oscale > 0.0f && d.oscale < 3.0e38f, "conv2ds: output scale of the packed weights missing");
    MV_REQUIRE((int64_t)d.H * d.W * d.ldx < (int64_t)1 << 30 && (d.x2 == nullptr || (int64_t)d.H * d.W * d.ldx2 < (int64_t)1 << 30),
               "conv2ds: one utterance's map too large (4 GiB)");
    const int p = d.ks / 2;
    const int Ho = (d.H + 2 * p - d.ks) / d.stride + 1, Wo = (d.W + 2 * p - d.ks) / sw + 1;
    CsPlan plan;
    int rc = cs_plan(d, Ho, Wo, sw, &plan);
    if (rc != MV_OK) return rc;
    Conv2dsArgs a;
    a.x = static_cast<const half_t*>(d.x); a.x2 = static_cast<const half_t*>(d.x2); a.w = static_cast<const half_t*>(d.w); a.bias = d.bias;
    a.res = static_cast<const half_t*>(d.res); a.res2 = static_cast<const half_t*>(d.res2); a.add = static_cast<const half_t*>(d.add);
    a.y = static_cast<half_t*>(d.y); a.y2 = static_cast<half_t*>(d.y2);
    a.ldx = d.ldx; a.ldx2 = d.ldx2; a.ldres = d.ldres; a.ldres2 = d.ldres2; a.ldadd = d.ldadd; a.ldy = d.ldy; a.ldy2 = d.ldy2;
    a.cinu = d.cin16 / 16;
    a.cin1u = d.x2 != nullptr ? d.cin1 / 16 : a.cinu;
    a.nchunks = (a.cinu + 1) / 2;
    a.wunits = 2 * a.nchunks;
    a.cout16 = d.cout16;
    a.B = d.B; a.H = d.H; a.W = d.W; a.Ho = Ho; a.Wo = Wo; a.sh = d.stride; a.sw = sw; a.epi = d.epi;
    a.lo = d.lo; a.hi = d.hi; a.oscale = d.oscale;
    a.peak = d.peak;
    a.R = plan.R; a.ncs = plan.ncs; a.tiles = plan.tiles; a.CT = plan.CT; a.ncons = plan.ncons; a.nprod = plan.nprod;
    a.pc = plan.pc; a.pcv = plan.pcv; a.pc_magic = plan.pc > 0 ? 65536 / plan.pc + 1 : 0;
    a.ns = plan.ns; a.pp = plan.pp; a.wg_per_ct = plan.wg_per_ct;
    const int prof = prof_begin(MV_PROF_CONV2D, 2.0 * d.B * Ho * Wo * (double)(d.cin_alg > 0 ? d.cin_alg : d.cin16) *
                                                (d.cout_alg > 0 ? d.cout_alg : d.cout16) * d.ks * d.ks, stream);
#define MV_CS_DISPATCH(KS)                                                                       \
    do {                                                                                         \
        if (plan.nbw == 3) {                                                                     \
            if (plan.spw == 8) rc = cs_launch<KS, 3, 8, 512>(a, plan, stream);                   \
            else if (plan.spw == 4) rc = cs_launch<KS, 3, 4, 512>(a, plan, stream);              \
            else rc = cs_launch<KS, 3, 2, 512>(a, plan, stream);                                 \
        } else if (plan.nbw == 2) {                                                              \
            if (plan.spw == 8) rc = cs_launch<KS, 2, 8, 512>(a, plan, stream);                   \
            else if (plan.spw == 4) rc = cs_launch<KS, 2, 4, 512>(a, plan, stream);              \
            else rc = cs_launch<KS, 2, 2, 512>(a, plan, stream);                                 \
        } else {                                                                                 \
            if (plan.spw == 8) rc = cs_launch<KS, 1, 8, 768>(a, plan, stream);                   \
            else if (plan.spw == 4) rc = cs_launch<KS, 1, 4, 768>(a, plan, stream);              \
            else if (plan.spw == 2) rc = cs_launch<KS, 1, 2, 768>(a, plan, stream);              \
            else rc = cs_launch<KS, 1, 1, 768>(a, plan, stream);                                 \
        }                                                                                        \
    } while (0)
    if (d.ks == 3) MV_CS_DISPATCH(3);
    else MV_CS_DISPATCH(1);
#undef MV_CS_DISPATCH
    prof_end(prof, stream);
    if (rc != MV_OK) return rc;
    return check_launch("conv2ds_kernel");
}

// ---- weights: fp32 [cout16][taps][cin16] (BatchNorm folded) -> split fp16 [cout16][taps][wunits][hi 16 | lo 16], scaled by 2^s -------------
int64_t conv2ds_packed_floats(int cout16, int cin16, int ks) { return (int64_t)cout16 * ks * ks * round_up(cin16, 32); }

float conv2ds_pack_host(const float* w, int cout16, int cin16, int ks, half_t* out) {
    const int taps = ks * ks, wunits = (int)round_up(cin16, 32) / 16;
    const size_t n = (size_t)cout16 * taps * cin16;
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    int s = 0;
    if (mx > 0.0f && std::isfinite(mx)) {
        int e;
        frexpf(mx, &e);   // mx = f * 2^e, f in [0.5, 1)
        s = 10 - e;       // mx * 2^s in [512, 1024)
    }
    const float ws = ldexpf(1.0f, s);
    memset(out, 0, (size_t)cout16 * taps * wunits * 32 * sizeof(half_t));
    for (int co = 0; co < cout16; ++co)
        for (int tp = 0; tp < taps; ++tp)
            for (int ci = 0; ci < cin16; ++ci) {
                const float V = w[((size_t)co * taps + tp) * cin16 + ci] * ws;
                const half_t h = (half_t)V;
                const half_t l = (half_t)(V - (float)h);
                half_t* unit = out + (((size_t)co * taps + tp) * wunits + ci / 16) * 32;
                unit[ci % 16] = h;
                unit[16 + ci % 16] = l;
            }
    return ldexpf(1.0f, -s) * CS_XSCALE_INV;
}

// ---- fp32 <-> S16 maps (stem output, interchange with fp32 code, tests): n_units units of 16 consecutive channels ----------------------------
__global__ void map_split_kernel(const float* x, half_t* y, int64_t n_units) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_units * 4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
        s16_store4(y + (i >> 2) * 32 + (i & 3) * 4, v);
    }
}
__global__ void map_merge_kernel(const half_t* x, float* y, int64_t n_units) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_units * 4; i += (int64_t)gridDim.x * blockDim.x)
        *reinterpret_cast<float4v*>(y + i * 4) = s16_load4(x + (i >> 2) * 32 + (i & 3) * 4);
}

int map_split_launch(const float* x, void* y, int64_t n, hipStream_t stream) {
    MV_REQUIRE(x != nullptr && y != nullptr && n > 0 && n % 16 == 0, "map_split: needs whole units of 16 channels");
    const int64_t work = n / 4;
    const int grid = (int)(ceil_div(work, 256) < 8192 ? ceil_div(work, 256) : 8192);
    MV_LAUNCH(map_split_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, x, static_cast<half_t*>(y), n / 16);
    return check_launch("map_split_kernel");
}
int map_merge_launch(const void* x, float* y, int64_t n, hipStream_t stream) {
    MV_REQUIRE(x != nullptr && y != nullptr && n > 0 && n % 16 == 0, "map_merge: needs whole units of 16 channels");
    const int64_t work = n / 4;
    const int grid = (int)(ceil_div(work, 256) < 8192 ? ceil_div(work, 256) : 8192);
    MV_LAUNCH(map_merge_kernel, (grid, 1, 1), (256, 1, 1), 0, stream, static_cast<const half_t*>(x), y, n / 16);
    return check_launch("map_merge_kernel");
}

}  // namespace mv

extern "C" {

int64_t mv_conv2ds_packed_elems(int32_t cout, int32_t cin, int32_t ks) {
    return mv::conv2ds_packed_floats((int)mv::round_up(cout, 16), (int)mv::round_up(cin, 16), ks);
}

int mv_conv2ds_pack_weight(const float* w, const float* out_scale, int32_t cout, int32_t cin, int32_t ks, void* packed, float* oscale,
                           mv_stream_t stream) {
    MV_REQUIRE(w != nullptr && packed != nullptr && oscale != nullptr && cout > 0 && cin > 0 && (ks == 1 || ks == 3), "mv_conv2ds_pack_weight: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int taps = ks * ks, cout16 = (int)mv::round_up(cout, 16), cin16 = (int)mv::round_up(cin, 16);
    std::vector<float> hw((size_t)cout * cin * taps), hs;
    MV_HIP_OK(hipMemcpyAsync(hw.data(), w, hw.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    if (out_scale != nullptr) {
        hs.resize((size_t)cout);
        MV_HIP_OK(hipMemcpyAsync(hs.data(), out_scale, hs.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    MV_HIP_OK(hipStreamSynchronize(st));
    std::vector<float> dense((size_t)cout16 * taps * cin16, 0.0f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tp = 0; tp < taps; ++tp)
                dense[((size_t)co * taps + tp) * cin16 + ci] = hw[((size_t)co * cin + ci) * taps + tp] * (hs.empty() ? 1.0f : hs[co]);
    std::vector<half_t> out((size_t)mv::conv2ds_packed_floats(cout16, cin16, ks) * 2);
    *oscale = mv::conv2ds_pack_host(dense.data(), cout16, cin16, ks, out.data());
    MV_HIP_OK(hipMemcpyAsync(packed, out.data(), out.size() * sizeof(half_t), hipMemcpyHostToDevice, st));
    MV_HIP_OK(hipStreamSynchronize(st));
    return MV_OK;
}

int mv_conv2ds_forward(const MvConv2dsDesc* d, mv_stream_t stream) {
    MV_REQUIRE(d != nullptr, "mv_conv2ds_forward: null descriptor");
    return mv::conv2ds_launch(*d, static_cast<hipStream_t>(stream));
}

int mv_map_split_f32(const float* x, void* y, int64_t n, mv_stream_t stream) { return mv::map_split_launch(x, y, n, static_cast<hipStream_t>(stream)); }
int mv_map_merge_f32(const void* x, float* y, int64_t n, mv_stream_t stream) { return mv::map_merge_launch(x, y, n, static_cast<hipStream_t>(stream)); }

}  // extern "C"
