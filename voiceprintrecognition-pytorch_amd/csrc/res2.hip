// Fused Res2Net chain of an SE-Res2Block (mvector/models/ecapa_tdnn.py:39-51): the `scale-1` dependent dilated
// convolutions  y_j = BN(ReLU(conv_j(x_j + y_{j-1})))  of ONE utterance run inside ONE workgroup.
//
// Why: each step is a tiny GEMM (T x width x 3*width); as separate launches they are latency bound (7 dependent
// launches per block, every intermediate makes an HBM round trip).  Here the step input lives in LDS for the whole
// chain ([T_pad x width] fp16, XOR-swizzled rows), the step output stays in the MFMA accumulators until every wave has
// finished reading the input, is then added to the next channel group x_{j+1} and written back over the input buffer.
// Only the weights stream in (global -> LDS direct, 4-slot ring of 64-wide K stages); they are shared by all
// workgroups and stay in L2.  HBM traffic is the algorithmic minimum: read x once, write y once.
// The activation buffer carries the reflect padding physically (PAD halo rows on either side, rewritten by the epilogue
// together with the rows they mirror), so a tap is a constant row shift: the row swizzle (row & 15) is then the same for
// every time tile of a lane and the ten operand reads of a K step are one address + immediates.
//
// Measured dead ends (r02j, tools/bench_res2.py, old and new kernel in one box: 156-160 us both): requesting the next
// step's first weight stages ahead of the epilogue and the x_{j+1} / parameter loads in the middle of the K loop, with
// run-time counted vmcnt waits so that the y stores and those loads drain under the K loop; one early wait instead of the
// compiler's s_waitcnt vmcnt(0) in front of every conditional store block; keeping the row addresses from being hoisted
// (256 VGPRs + 4 spills -> 208, no spills); tile-level uniform branches in the epilogue.  All neutral together: the chain is
// not obviously bound by its store / load latencies or by the epilogue's instruction count.  In-kernel timeline of THIS kernel
// (MV_PROBE=1, tools/trace_res2.py, profiles/r02j_res2_chain_inkernel_timeline.log), wave 0, per step of ~42 k cycles: 5-9 k
// until the first weight stage (and the older x loads / y stores) has landed, 6 x 2.4 k K stages, ~16 k epilogue (20 stores,
// each behind a compiler-inserted s_waitcnt vmcnt(0)).  The restructured variant (tools/variants/) removes exactly those waits
// and still measures the same: its timeline is the next thing to take.
//
// Round 2 (r05c-r05h, tools/bench_res2.py old vs new in one box, profiles/r05_res2_chain_direct.log): width 128 runs the DIRECT form below
// (121-125 -> 107-110 us): weight fragments from global memory into registers two stages ahead, no weight ring and no barrier inside the K
// loop, hand-scheduled K stage, x_{j+1} by LDS-DMA in the ring's place.  Steps on the way: the hand-scheduled stage alone inside the ring form
// 116 us; + the next stage's activation fragments in flight across the stage barrier: the same (the stage barrier itself, not the LDS latency
// behind it, was the cost); direct weights one stage ahead + x_{j+1} in registers 113-116 (the L2 latency is longer than a stage); two stages
// ahead + x_{j+1} as ONE burst of LDS-DMA at the start of the step 116-126 (80 transfers per CU in front of stage 0's MFMAs); the burst
// spread over the stages, two transfers per wave and stage: 107-110.  What is left per step of ~27 k cycles (timeline r05g): four stages at
// the matrix pipe's pace (~1.1-1.3 k), two late ones (2.5-4.5 k: the first fragment loads behind the previous epilogue's y stores -- one
// in-order counter for loads and stores, and the chip-wide store burst of an epilogue takes thousands of cycles to retire), ~6 k epilogue,
// ~2.7 k in the barrier in front of the step.
//
// Round 3 (r10b/c, tools/variants/res2_chain_drip_stores.patch.txt, profiles/r10b_res2_chain_drip_stores_ab.log): "drip" form -- y_j written into the
// LDS region of x_{j+1} by the epilogue (slice 0 by the prologue) and stored to HBM during the NEXT step's K loop, two 16-byte stores per wave
// and stage right in front of the transfers that refill the rows, so that no burst of ten stores per lane sits in front of the next step's
// fragment loads.  Correct (parity tests) and 13 % SLOWER (107-108 -> 120-122 us): the untracked stores share the in-order counter with the
// fragment loads the compiler waits for, every fragment wait now also waits for the stores of the stage before (the chain without ANY y
// stores: 95 us, so the stores cost 15 us in the epilogue and 26 us in the loop).  Parked.
//
// Round 6 (r15bb-bd, ISA reading): every weight-fragment request used to sit behind a condition ("is there such a stage / a next step"), i.e. in a block the
// wave may skip, and at each join the compiler's wait-count insertion assumes a skipped request: its waits came out four to eight operations too strict --
// vmcnt(1) / vmcnt(0) in the middle of every step's last stage, right behind the four requests just issued.  The direct form is now straight-line code (k = 3
// only: six written-out stages, requests unconditional, the last step re-requests its own weights, the first two stages requested in front of the prologue copy)
// and the compiler's waits are the vmcnt(9) / vmcnt(8) the two-stage lead was written for; the epilogue's parameters are requested in front of the step's last
// ten MFMAs.  Chain alone 104-111 -> 100-105 us; IN SITU (per-dispatch medians of the headline step) 103-105 -> 102.5-103 us -- behind the tdnn1 GEMM the chain is
// not bound by these waits.  All ten y stores first and the LDS part in a second pass: +4 us (one LDS round trip per tile with nothing to hide it behind).
// Every second workgroup of an XCD started 1-6 us late (is the y-store burst of 256 workgroups in lockstep what the next step waits for?): slower by the delay.
//
// Long utterances (r10u): beyond 320 frames the launcher cuts an utterance into chunks (Res2Args::nchunks / useful / halo) -- the kernel body runs
// unchanged on a chunk's rows + halo, only the row base and the y store mask are per chunk: EcapaTdnn-1024 at 6 s 177 -> 204 k audio-seconds/s
// against one launch per step.
//
// Work split: 8 waves; wave w owns output channel tiles {MI*(w&3) .. +MI} and the time tiles of half (w>>2).
// torch.chunk / torch.cat never exist: slices are addressed inside the [B, T, C] tensors, slice 0 is copied through.
#include <type_traits>

#include "kernels.h"

namespace mv {

constexpr bool R2_DIRECT = true;  // width 128, T <= 304: weights straight into registers, hand-scheduled K stage (below)
constexpr int R2_XNEXT_BYTES = 77824;         // direct form: [<= 304 rows][128] fp16 of the next channel group, where the weight ring was
constexpr int R2_THREADS = 512;
constexpr int R2_NH = 10;            // time tiles (16 frames) per wave half -> T <= 320
constexpr int R2_MAX_STEPS = 15;
constexpr int R2_WSTAGE_BYTES = 128 * 64 * 2;  // one K stage of weights: [<=128 rows][64] fp16
constexpr int R2_RING = 4;                     // weight stages in flight (ring of LDS slots)
constexpr int R2_MAXPAD = 8;                   // halo rows on either side of the activation buffer
constexpr int R2_ROWS = 2 * R2_NH * 16 + 2 * R2_MAXPAD;  // rows of the activation buffer (all 20 time tiles + halo)

struct Res2Args {
    const half_t* x;   // [B, T, C]  tdnn1 output
    half_t* y;         // [B, T, C]  res2net output
    const half_t* w[R2_MAX_STEPS];      // packed [width][k][width_pad] per step
    const float* bias[R2_MAX_STEPS];
    const float* scale[R2_MAX_STEPS];
    const float* shift[R2_MAX_STEPS];
    int T, C, width, steps, k, dil, kpad;  // kpad = round_up(width, 64)
    // Long utterances (T beyond what the LDS buffer holds): nchunks workgroups per utterance.  Chunk c produces the rows [c * useful, (c + 1) * useful)
    // and runs the whole chain on them + `halo` = steps * pad rows on either side (clipped to the utterance): the receptive field of the chain,
    // so every produced row sees exactly what it sees in the unchunked run; the halo rows' own results (wrong towards an interior edge, where the
    // kernel reflects instead of seeing the neighbour's rows) are never stored.  nchunks = 1: the whole utterance, as before.
    int nchunks, useful, halo;
};

// (LDS-DMA transfers, counted waits, the LDS-only barrier and the fp16 saturation are the arch header's glds16 / wait_vm /
// lds_barrier / fmed3)
__device__ __forceinline__ void r2_glds16(const void* gsrc, char* lds_wave_base) { glds16(gsrc, lds_wave_base); }
template <int N>
__device__ __forceinline__ void r2_wait_vm() { wait_vm<N>(); }
__device__ __forceinline__ void r2_lds_barrier() { lds_barrier(); }
__device__ __forceinline__ float r2_clamp_h(float v) { return fmed3(v, -65504.0f, 65504.0f); }  // fp16 saturation


// MI = output channel tiles per wave (width = 64 * MI)
// NHT = time tiles (16 frames) per wave half: 10 (<= 320 frames per workgroup; direct form 304) or, direct form only, 5 (<= 160 frames: the chunks
// of a SMALL batch -- one utterance as three workgroups instead of one, each with half the matrix work, epilogue and LDS traffic per step)
template <int MI, bool DIRECT = false, int NHT = R2_NH>
__global__ __launch_bounds__(R2_THREADS) void res2_chain_kernel(Res2Args a) {
    static_assert(!DIRECT || MI == 2, "the direct form is written for width 128");
    static_assert(NHT == R2_NH || (DIRECT && NHT == 5), "5 time tiles per wave half exist in the direct form only");
    constexpr int ROWS_T = 2 * NHT * 16 + 2 * R2_MAXPAD;            // rows of the activation buffer
    constexpr int XNEXT_T = NHT == R2_NH ? R2_XNEXT_BYTES : 2 * NHT * 16 * 64 * MI * 2;   // direct form: the next channel group's rows
    MV_DYN_SMEM(smem);
    fp16_saturation_on();                // the epilogue's fp16 conversions and packed sums saturate in hardware (half_hwsat / pk_add_hwsat)
    constexpr int WIDTH = 64 * MI;
    constexpr int CPR = WIDTH / 8;       // 16-byte chunks per activation row
    constexpr int ROWB = WIDTH * 2;      // bytes per activation row
    constexpr int TP = MI;               // weight transfers per stage per wave (WIDTH/8 transfers over 8 waves)
    char* wbuf = smem;                               // R2_RING x R2_WSTAGE_BYTES  (direct form: the next channel group instead)
    char* abuf = smem + (DIRECT ? XNEXT_T : R2_RING * R2_WSTAGE_BYTES);   // [R2_ROWS][WIDTH] fp16, row r = t + PAD, chunk index ^= r & (CPR-1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = MV_UNIFORM(wave);
    const int fr = lane & 15, fg = lane >> 4;
    const int b = a.nchunks > 1 ? (int)blockIdx.x / a.nchunks : (int)blockIdx.x;
    const int chunk = a.nchunks > 1 ? (int)blockIdx.x - b * a.nchunks : 0;
    const int u0 = chunk * a.useful, u1 = u0 + a.useful < a.T ? u0 + a.useful : a.T;   // rows this workgroup produces
    const int l0 = u0 - a.halo > 0 ? u0 - a.halo : 0, l1 = u1 + a.halo < a.T ? u1 + a.halo : a.T;   // rows it works on
    const int T = l1 - l0;                        // local frame count: everything below is the unchunked kernel on rows [l0, l1)
    const int st0 = u0 - l0, st1 = u1 - l0;       // local rows whose results are stored
    const int half_k = (a.k - 1) / 2;
    const int PAD = half_k * a.dil;
    const int64_t rowbase = (int64_t)b * a.T + l0;
    const half_t* xb = a.x + rowbase * a.C;
    half_t* yb = a.y + rowbase * a.C;
    auto a_off = [&](int row, int chunk) { return row * ROWB + ((chunk ^ (row & (CPR - 1))) << 4); };

    const int cw = wave & 3;   // channel tile group
    const int th = wave >> 2;  // time half
    const int nh0 = th * NHT;  // first time tile of this wave
    // direct form: k = 3 and width 128 (launcher), so a step is SIX K stages of 64 channels, two per tap -- compile-time, the K loop is straight-line code
    // (behind a run-time trip count the wait-count insertion merged "loop not entered" with "loop left" at the loop's exit and again waited for all but the
    // youngest four requests, i.e. also for the stage requested one stage earlier)
    const int kstages_per_tap = DIRECT ? 2 : a.kpad / 64;
    const int nstages = DIRECT ? 6 : a.k * kstages_per_tap;
    const int wk = DIRECT ? 3 : a.k, wkpad = DIRECT ? 128 : a.kpad;   // weight layout [width][k][kpad]
    // weight transfers: a stage is [WIDTH rows][64] = WIDTH/8 transfers of 1 KiB; wave w issues transfers w, w+8
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ lrow;
    // operand rows of this lane: time tile ni starts at row (nh0 + ni) * 16 + fr (+ tap shift + PAD)
    const int lane_row0 = nh0 * 16 + fr;

    // ---- width 128, direct form (round 2) -------------------------------------------------------------------------------------------
    // In-kernel timeline of the ring form (r05c): a K stage = 1650-1750 cycles from the stage barrier to the next wait + 700-800 in the
    // barrier, for 1280 cycles of matrix work per SIMD (2 waves x 40 MFMAs); every step starts by waiting for its first weight stage,
    // and the ten x_{j+1} register loads in front of the last stage cost 3.6 k cycles of issue.  Here:
    //  * a wave's weight operands of a stage are just FOUR 16-byte fragments per lane (2 K halves x 2 channel tiles), so they skip LDS:
    //    each lane loads them from global memory (L2 resident: 96 KiB per step, the two time halves of a channel group share the lines
    //    through L1) TWO stages ahead into a rotation of three register sets -- one stage of lead does not cover the L2 latency (r05e);
    //  * no weight ring, and -- the activation buffer being constant inside a step -- NO barrier inside the K loop: the waves of a SIMD
    //    drift apart and keep its matrix pipe fed; the only barriers left are the two around the epilogue;
    //  * activation fragments: two register groups of five time tiles, read by inline assembly one phase ahead of the MFMAs that use
    //    them (counted lgkmcnt waits; the compiler's own schedule fell back to "one read, full wait, two MFMAs" in the second K half),
    //    the first group of stage s + 1 under the last phases of stage s;
    //  * the next channel group x_{j+1} arrives by LDS-DMA in the LDS the ring used to occupy (requested at the start of the step, read
    //    by the epilogue; rows XOR-swizzled through the source addresses), not through 40 registers per lane.
    half8v wf[3][2][2];  // [set][K half][channel tile]
    auto load_wf = [&](const half_t* wj, int s, half8v (&dst)[2][2]) {
        const int tap = s / kstages_per_tap;
        const int c0 = (s - tap * kstages_per_tap) * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = (cw * 2 + mi) * 16 + fr;
                dst[kk][mi] = *MV_GLOBAL_PTR(half8v, wj + ((int64_t)row * wk + tap) * wkpad + c0 + (kk * 4 + fg) * 8);
            }
        MV_VM_LOADS(4);   // (the direct form's r2_wait_vm<4>() behind the last stage counts these four)
    };
    // The fragments of the stage two stages ahead are requested UNCONDITIONALLY, by straight-line code: the last two stages of a step request stages 0 and 1 of
    // the next step's weights, the last step re-requests its own (values never used).  Behind a condition ("is there a next step") each request sat in a
    // skipped block, and at every join the compiler's wait-count insertion assumed it had NOT been issued: its waits for the fragments of the running stage came
    // out four to eight operations too strict -- vmcnt(1) / vmcnt(0) in the middle of the last stage, right behind the four requests just issued (one exposed L2
    // round trip per step), vmcnt(5) / vmcnt(4) elsewhere (a lead of one stage where two were written), and a scalar load of the weight pointer with
    // s_waitcnt lgkmcnt(0) -- which also drains the LDS fragment reads issued a phase ahead -- in front of every request (ISA of round 6, r15bb).
    // (the first two stages' fragments are requested in FRONT of the prologue copy below: its own waits retire them, so the first step starts without a wait the
    // compiler would otherwise carry into every step's stage 0 through the loop header -- and there it waits for the previous epilogue's y stores as well)
    if constexpr (DIRECT) {
        load_wf(a.w[0], 0, wf[0]);
        load_wf(a.w[0], 1, wf[1]);
    }

    // ---- slice 0 passes through; slice 1 (with its reflected halo) becomes the first step's input; rows beyond stay zero ----
    for (int i = tid; i < ROWS_T * CPR; i += R2_THREADS) {
        const int row = i / CPR, ch = i - row * CPR;
        const int t = row - PAD;
        half8v v1;
#pragma unroll
        for (int e = 0; e < 8; ++e) v1[e] = (half_t)0.0f;
        if (t >= -PAD && t < T + PAD) {
            const int ts = t < 0 ? -t : (t >= T ? 2 * (T - 1) - t : t);
            v1 = *reinterpret_cast<const half8v*>(xb + (int64_t)ts * a.C + WIDTH + ch * 8);
            if (t >= st0 && t < st1)
                *reinterpret_cast<half8v*>(yb + (int64_t)t * a.C + ch * 8) = *reinterpret_cast<const half8v*>(xb + (int64_t)t * a.C + ch * 8);
        }
        *reinterpret_cast<half8v*>(abuf + a_off(row, ch)) = v1;
    }


    for (int j = 1; j <= a.steps; ++j) {

        const half_t* wj = a.w[j - 1];
        [[maybe_unused]] const half_t* wnext = a.w[j < a.steps ? j : j - 1];   // direct form: the weights whose first two stages the last two stages request
        auto issue_w = [&](int s, int buf) {
            const int tap = s / kstages_per_tap;
            const int c0 = (s - tap * kstages_per_tap) * 64;
            for (int tr = wave; tr < WIDTH / 8; tr += 8) {
                const int co = tr * 8 + lrow;
                r2_glds16(wj + ((int64_t)co * a.k + tap) * a.kpad + c0 + kc * 8, wbuf + buf * R2_WSTAGE_BYTES + tr * 1024);
            }
        };
        // the next channel group x_{j+1} (rows clamped, stores predicated) is requested when the LAST K stage starts: its
        // latency runs under that stage's MFMAs and the 40 registers are not held through the whole K loop (which now keeps
        // two phases of operand fragments in flight).  (Unconditional loads: a select between a load and zero compiles to one
        // branch + full wait per load; the last step simply re-reads its own group and ignores the values.)
        const bool more = j < a.steps;
        const int next_group = more ? j + 1 : j;
        half4v xn[MI][NHT];   // MI == 1: this wave's 4 channels per tile
        half8v xp[NHT];       // MI == 2: 8 consecutive channels per lane (paired layout of the epilogue)
        const int co8 = (cw * 2 + (fg & 1)) * 16 + 8 * (fg >> 1);
        if constexpr (!DIRECT)
            for (int s0 = 0; s0 < R2_RING - 1 && s0 < nstages; ++s0) issue_w(s0, s0);
        float4v acc[MI][NHT];
        [[maybe_unused]] float4v pbias4[2], pscale4[2], pshift4[2];   // direct form: the epilogue's parameters, requested in front of the last K stage
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NHT; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        // Weight stages run up to R2_RING-1 ahead of the MFMAs: wait (counted) for stage s, barrier, refill the slot that
        // stage s-1 just released with stage s+3, compute stage s.  One barrier per stage; it also publishes the
        // activation buffer written by the previous step's epilogue.
        auto do_stage = [&](int s, auto LAST) __attribute__((always_inline)) {
            const int last_issued = s + R2_RING - 2 < nstages - 1 ? s + R2_RING - 2 : nstages - 1;
            const int younger = last_issued - s;  // stages issued after stage s
            if (younger >= 2) {
                r2_wait_vm<2 * TP>();
            } else if (younger == 1) {
                r2_wait_vm<TP>();
            } else {
                r2_wait_vm<0>();
            }

            r2_lds_barrier();

            if (s + R2_RING - 1 < nstages) issue_w(s + R2_RING - 1, (s + R2_RING - 1) % R2_RING);
            if constexpr (decltype(LAST)::value) {
#pragma unroll
                for (int ni = 0; ni < NHT; ++ni) {
                    int t = (nh0 + ni) * 16 + fr;
                    t = t < T ? t : T - 1;
                    MV_OPAQUE(t);
                    if constexpr (MI == 2) {
                        xp[ni] = *reinterpret_cast<const half8v*>(xb + (int64_t)t * a.C + next_group * WIDTH + co8);
                    } else {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            xn[mi][ni] = *reinterpret_cast<const half4v*>(xb + (int64_t)t * a.C + next_group * WIDTH + (cw * MI + mi) * 16 + 4 * fg);
                    }
                }
            }
            const int buf = s % R2_RING;
            const int tap = s / kstages_per_tap;
            const int c0 = (s - tap * kstages_per_tap) * 64;
            const int row0 = lane_row0 + (tap - half_k) * a.dil + PAD;  // >= 0; row0 + 16 * ni is tile ni's operand row
            const int sw = row0 & (CPR - 1);                            // the same for every tile: 16 rows per tile
            const char* wt = wbuf + buf * R2_WSTAGE_BYTES;
            // The stage's 2 K halves x 2 groups of 5 time tiles are four phases; the operand reads of phase p + 1 are issued
            // before the MFMAs of phase p, so an LDS round trip never sits in front of the matrix pipe (the compiler's own
            // order "reads, wait, MFMAs" per half costs 2.45 k cycles per stage for 1.3 k cycles of matrix work: in-kernel
            // timeline r02j).  WIDTH is a multiple of 64, so both K halves hold real weights.
            static_assert(DIRECT || NHT % 2 == 0, "time tiles are processed in two groups");
            constexpr int NG = NHT / 2 > 0 ? (NHT + 1) / 2 : 1;
            half8v af[2][MI];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = (cw * MI + mi) * 16 + fr;
                    af[kk][mi] = *reinterpret_cast<const half8v*>(wt + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                }
            const int bchunk = (c0 >> 3) + fg;
            const char* bp0 = abuf + row0 * ROWB + ((bchunk ^ sw) << 4);
            const char* bp1 = abuf + row0 * ROWB + (((bchunk + 4) ^ sw) << 4);
            half8v bq[4][NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) bq[0][g] = *reinterpret_cast<const half8v*>(bp0 + g * 16 * ROWB);
            auto phase = [&](auto PH) __attribute__((always_inline)) {
                constexpr int ph = decltype(PH)::value;
                if constexpr (ph < 3) {
                    const char* bp = ((ph + 1) >> 1) ? bp1 : bp0;
#pragma unroll
                    for (int g = 0; g < NG; ++g) bq[ph + 1][g] = *reinterpret_cast<const half8v*>(bp + (((ph + 1) & 1) * NG + g) * 16 * ROWB);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][(ph & 1) * NG + g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ph >> 1][mi], bq[ph][g], acc[mi][(ph & 1) * NG + g], 0, 0, 0);
                // keep the order written above: NG reads of the next phase, then this phase's MFMAs
                if constexpr (ph < 3) MV_SCHED_GROUP(0x100, NG);
                MV_SCHED_GROUP(0x008, NG * MI);
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
        };
        if constexpr (DIRECT) {
            constexpr int NG = 5;   // mfma10_step takes five time tiles: two groups per wave half (NHT = 10) or one (NHT = 5)
            // x_{j+1} transfers per wave and stage: ceil(76 / 8) = 10 per wave spread over the step's stages (>= 6: k >= 3)
            constexpr int XPS = 2;
            half8v bA[5], bB[5];
            auto b_addr = [&](int s, int khalf) {  // LDS byte address of time tile 0's fragment for K half `khalf` of stage s
                const int tap = s / kstages_per_tap;
                const int c0 = (s - tap * kstages_per_tap) * 64;
                const int row0 = lane_row0 + (tap - half_k) * a.dil + PAD;  // >= 0; row0 + 16 * ni is tile ni's operand row
                const int chunk = (c0 >> 3) + fg + 4 * khalf;
                return lds_addr(abuf) + (unsigned)(row0 * ROWB + ((chunk ^ (row0 & (CPR - 1))) << 4));
            };
            auto request_params = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int co = (cw * 2 + mi) * 16 + 4 * fg;
                    pbias4[mi] = *MV_GLOBAL_PTR(float4v, a.bias[j - 1] + co);
                    pscale4[mi] = *MV_GLOBAL_PTR(float4v, a.scale[j - 1] + co);
                    pshift4[mi] = *MV_GLOBAL_PTR(float4v, a.shift[j - 1] + co);
                }
                MV_VM_LOADS(6);
            };
            auto stage = [&](int s, half8v (&cur)[2][2], half8v (&ahead)[2][2], auto LAST, const half_t* w_ahead, int s_ahead) __attribute__((always_inline)) {
                load_wf(w_ahead, s_ahead, ahead);
                const unsigned bp1 = b_addr(s, 1);
                if constexpr (NHT == 5) {
                    // one group of five time tiles: bA holds K half 0 of this stage (requested one phase earlier), bB takes K half 1 while the
                    // MFMAs of half 0 run, bA then takes half 0 of the next stage under the MFMAs of half 1
                    lds_read5<0, 16 * ROWB>(bB, bp1);
                {
                    // Request x_{next_group}: 4 rows of 256 B per transfer, row r's 16-byte chunk c lands at chunk position c ^ (r & 15);
                    // XPS transfers per wave and stage, so that the memory pipe takes them between the weight loads instead of as one
                    // burst of 80 per CU that the waves would sit behind (r05f: 3-4 k cycles in front of the MFMAs of stage 0).
                    // Untracked, and issued behind the stage's first MFMAs: the compiler counts the waits for the weight fragments from
                    // the loads it knows, so a transfer in front of a fragment's first use would be waited for with it.
                    const int nrow4 = (T + 3) >> 2;
#pragma unroll
                    for (int u = 0; u < XPS; ++u) {
                        const int i = wave_u + 8 * (s * XPS + u);
                        if (i < nrow4) {
                            int row = 4 * i + (lane >> 4);
                            row = row < T ? row : T - 1;
                            const int chunk = (lane & 15) ^ (row & 15);
                            glds16_untracked(xb + (int64_t)row * a.C + next_group * WIDTH + chunk * 8, lds_addr(wbuf) + i * 1024);
                        }
                    }
                }
                    mfma10_step<5>(&acc[0][0], &acc[1][0], cur[0][0], cur[0][1], bA);
                    if constexpr (decltype(LAST)::value) {
                        mfma10_step<0>(&acc[0][0], &acc[1][0], cur[1][0], cur[1][1], bB);
                    } else {
                        const unsigned np0 = b_addr(s + 1, 0);
                        lds_read5<0, 16 * ROWB>(bA, np0);
                        mfma10_step<5>(&acc[0][0], &acc[1][0], cur[1][0], cur[1][1], bB);
                    }
                } else {
                mfma10_step<0>(&acc[0][0], &acc[1][0], cur[0][0], cur[0][1], bA);
                lds_read5<0, 16 * ROWB>(bA, bp1);  // (these registers were last sourced by the step above)
                {
                    // Request x_{next_group}: 4 rows of 256 B per transfer, row r's 16-byte chunk c lands at chunk position c ^ (r & 15);
                    // XPS transfers per wave and stage, so that the memory pipe takes them between the weight loads instead of as one
                    // burst of 80 per CU that the waves would sit behind (r05f: 3-4 k cycles in front of the MFMAs of stage 0).
                    // Untracked, and issued behind the stage's first MFMAs: the compiler counts the waits for the weight fragments from
                    // the loads it knows, so a transfer in front of a fragment's first use would be waited for with it.
                    const int nrow4 = (T + 3) >> 2;
#pragma unroll
                    for (int u = 0; u < XPS; ++u) {
                        const int i = wave_u + 8 * (s * XPS + u);
                        if (i < nrow4) {
                            int row = 4 * i + (lane >> 4);
                            row = row < T ? row : T - 1;
                            const int chunk = (lane & 15) ^ (row & 15);
                            glds16_untracked(xb + (int64_t)row * a.C + next_group * WIDTH + chunk * 8, lds_addr(wbuf) + i * 1024);
                        }
                    }
                }
                mfma10_step<5>(&acc[0][NG], &acc[1][NG], cur[0][0], cur[0][1], bB);
                lds_read5<NG * 16 * ROWB, 16 * ROWB>(bB, bp1);
                mfma10_step<5>(&acc[0][0], &acc[1][0], cur[1][0], cur[1][1], bA);
                if constexpr (decltype(LAST)::value) {
                    // the epilogue's bias / scale / shift are requested HERE, in front of the step's last ten MFMAs (the first fragment group and the K half 0
                    // fragments are dead: their registers take them) -- requested at the start of the epilogue they are one exposed L2 round trip per step
                    request_params();
                    mfma10_step<0>(&acc[0][NG], &acc[1][NG], cur[1][0], cur[1][1], bB);
                } else {
                    const unsigned np0 = b_addr(s + 1, 0);
                    lds_read5<0, 16 * ROWB>(bA, np0);
                    mfma10_step<5>(&acc[0][NG], &acc[1][NG], cur[1][0], cur[1][1], bB);
                    lds_read5<NG * 16 * ROWB, 16 * ROWB>(bB, np0);
                }
                }
            };
            r2_lds_barrier();  // publishes the activation buffer written by the previous epilogue (or the prologue); x_{j+1} of the
                               // previous step has been read by everyone
            {
                const unsigned bp0 = b_addr(0, 0);
                lds_read5<0, 16 * ROWB>(bA, bp0);
                if constexpr (NHT != 5) lds_read5<NG * 16 * ROWB, 16 * ROWB>(bB, bp0);
            }
            // six stages, the three fragment sets in rotation, every set requested two stages ahead of its use
            stage(0, wf[0], wf[2], std::false_type{}, wj, 2);
            stage(1, wf[1], wf[0], std::false_type{}, wj, 3);
            stage(2, wf[2], wf[1], std::false_type{}, wj, 4);
            stage(3, wf[0], wf[2], std::false_type{}, wj, 5);
            stage(4, wf[1], wf[0], std::false_type{}, wnext, 0);
            // Small-batch form: the epilogue's bias / scale / shift are requested HERE, one stage ahead of their use (older than the last stage's four
            // fragment loads, so the counted wait behind the K loop covers them) -- requested at the start of the epilogue they are one exposed L2 round
            // trip per step: one utterance 427.7 -> 420.5 us of GPU time (r14v).  The full-batch form requests them inside the last stage, in front of its last
            // ten MFMAs, where two fragment groups have died (r15bc; one stage earlier it measured -0.9 % on the headline in round 5: 18 more registers
            // through the whole stage -- and, as the ISA of round 6 showed, behind waits that drained them at once).
            if constexpr (NHT == 5) request_params();
            stage(5, wf[2], wf[1], std::true_type{}, wnext, 1);
            mfma_hazard_pad();  // the assembly MFMAs are invisible to the compiler's hazard padding
            // x_{j+1} has landed for this wave once at most the fragment loads of the last stage (younger than every transfer; none after
            // the last step) are outstanding; the barrier below then publishes every wave's rows
            // (at most 10 transfers per wave, XPS = 2 per stage: the last ones leave in stage 4 at the latest, and nstages >= 6, so the
            // four fragment loads of the last stage are younger.  Requesting the next step's stage-2 fragments here, ahead of the epilogue's
            // y stores, moved the late stage from 2 to 3 and made stage 0 and the epilogue longer: 115 us instead of 108, r05h.)
            // (the last step's four are its own fragments once more: never used, gone with the wave; the full-batch form's six parameter requests are younger still)
            if constexpr (NHT == 5) {
                r2_wait_vm<4>();
            } else {
                r2_wait_vm<10>();
            }
        } else {
            for (int s = 0; s + 1 < nstages; ++s) do_stage(s, std::false_type{});
            do_stage(nstages - 1, std::true_type{});  // peeled: the x_{j+1} registers exist from here on only
        }

        r2_lds_barrier();  // every wave is done with the activation buffer and the weight ring
        // ---- epilogue: y_j = BN(ReLU(acc + bias)); next input = x_{j+1} + y_j written over the activation buffer,
        //      rows 1..PAD and T-1-PAD..T-2 also into the halo rows that mirror them ----
        if constexpr (MI == 2) {
            // The two channel tiles of the wave are paired through v_permlane16_swap: afterwards a lane holds 8 consecutive
            // channels (even lane rows: tile 0, odd rows: tile 1), so y_j leaves in 16-byte stores (64 contiguous bytes per time
            // step and wave instead of two rounds of 32), x_{j+1} arrives in 16-byte loads and the next input is one 16-byte LDS
            // write per tile -- half the store / load / ds_write instructions of the 8-byte form (the epilogue was as long as the
            // K loop: in-kernel timeline r02j, store-issue bound).
            float4v bias4[2], scale4[2], shift4[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if constexpr (DIRECT) {   // (requested in front of the last K stage / of its last MFMA group)
                    bias4[mi] = pbias4[mi];
                    scale4[mi] = pscale4[mi];
                    shift4[mi] = pshift4[mi];
                } else {
                    const int co = (cw * 2 + mi) * 16 + 4 * fg;
                    bias4[mi] = *reinterpret_cast<const float4v*>(a.bias[j - 1] + co);
                    scale4[mi] = *reinterpret_cast<const float4v*>(a.scale[j - 1] + co);
                    shift4[mi] = *reinterpret_cast<const float4v*>(a.shift[j - 1] + co);
                }
            }
#pragma unroll
            for (int ni = 0; ni < NHT; ++ni) {
                int t = (nh0 + ni) * 16 + fr;
                MV_OPAQUE(t);
                unsigned w[2][2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    half4v hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = half_hwsat(fmaxf(acc[mi][ni][r] + bias4[mi][r], 0.0f) * scale4[mi][r] + shift4[mi][r]);   // (saturating: FP16_OVFL)
                    __builtin_memcpy(w[mi], &hv, 8);
                }
                row_swap_odd_even(w[0][0], w[1][0]);
                row_swap_odd_even(w[0][1], w[1][1]);
                const unsigned o[4] = {w[0][0], w[0][1], w[1][0], w[1][1]};
                half8v ov;
                __builtin_memcpy(&ov, o, 16);
                half8v xnext;
                if constexpr (!DIRECT) {
                    xnext = xp[ni];
                } else {
                    const int tc = t < T ? t : T - 1;
                    xnext = *reinterpret_cast<const half8v*>(wbuf + tc * ROWB + (((co8 >> 3) ^ (tc & 15)) << 4));
                }
                const half8v nv = pk_add_hwsat(ov, xnext);   // the packed sum saturates in hardware (FP16_OVFL)
                // (the y store stays HERE, between this tile's arithmetic and its LDS writes: all ten stores first and the LDS part in a second pass over the tiles
                // measured +4 us on the chain -- the second pass is one LDS round trip per tile with nothing to hide it behind, r15bc)
                if (t >= st0 && t < st1) *reinterpret_cast<half8v*>(yb + (int64_t)t * a.C + j * WIDTH + co8) = ov;
                if (t < T) {
                    if (more) {
                        *reinterpret_cast<half8v*>(abuf + a_off(t + PAD, co8 >> 3)) = nv;
                        if (t >= 1 && t <= PAD) *reinterpret_cast<half8v*>(abuf + a_off(PAD - t, co8 >> 3)) = nv;
                        if (t >= T - 1 - PAD && t <= T - 2) *reinterpret_cast<half8v*>(abuf + a_off(2 * (T - 1) - t + PAD, co8 >> 3)) = nv;
                    }
                }
            }
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int co = (cw * MI + mi) * 16 + 4 * fg;
            const float4v bias4 = *reinterpret_cast<const float4v*>(a.bias[j - 1] + co);
            const float4v scale4 = *reinterpret_cast<const float4v*>(a.scale[j - 1] + co);
            const float4v shift4 = *reinterpret_cast<const float4v*>(a.shift[j - 1] + co);
#pragma unroll
            for (int ni = 0; ni < NHT; ++ni) {
                const int t = (nh0 + ni) * 16 + fr;
                half4v hv, nv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = half_hwsat(fmaxf(acc[mi][ni][r] + bias4[r], 0.0f) * scale4[r] + shift4[r]);
                // x_{j+1} + y_j in packed fp16: the sum of two fp16 values rounded once IS what float-add-then-round gives;
                // saturation by packed min / max (the in-kernel timeline shows the epilogue as expensive as the whole K loop:
                // ~16 k of ~42 k cycles per step, all VALU)
                nv = pk_add_hwsat(hv, xn[mi][ni]);
                if (t >= st0 && t < st1) *reinterpret_cast<half4v*>(yb + (int64_t)t * a.C + j * WIDTH + co) = hv;
                if (t < T) {
                    if (more) {
                        const int cb = (co & 7) * 2;
                        *reinterpret_cast<half4v*>(abuf + a_off(t + PAD, co >> 3) + cb) = nv;
                        if (t >= 1 && t <= PAD) *reinterpret_cast<half4v*>(abuf + a_off(PAD - t, co >> 3) + cb) = nv;
                        if (t >= T - 1 - PAD && t <= T - 2)
                            *reinterpret_cast<half4v*>(abuf + a_off(2 * (T - 1) - t + PAD, co >> 3) + cb) = nv;
                    }
                }
            }
        }
        }
        // the next step's first stage barrier publishes these LDS writes

    }
}


static bool res2_direct(int T, int width, int k) {  // (six K stages per step, written out: k = 3, two 64-channel stages per tap)
    return R2_DIRECT && width == 128 && k == 3 && ((T + 3) & ~3) * width * 2 <= R2_XNEXT_BYTES;
}

constexpr int R2_SMALL_ROWS = 160;   // frames a workgroup of the 5-tile direct form holds (small batches)
constexpr int R2_SMALL_LDS = R2_SMALL_ROWS * 128 * 2 + (R2_SMALL_ROWS + 2 * R2_MAXPAD) * 128 * 2;   // next channel group + activation buffer

size_t res2_chain_lds_bytes(int T, int width, int k) {
    if (res2_direct(T, width, k)) return (size_t)R2_XNEXT_BYTES + (size_t)R2_ROWS * width * 2;  // 163 840 B: all of the CU's LDS
    return R2_RING * (size_t)R2_WSTAGE_BYTES + (size_t)R2_ROWS * width * 2;
}

// frames one workgroup can hold (direct form: the x_{j+1} region; ring form: the 20 time tiles of the accumulators)
static int res2_local_limit(int width, int k) { return R2_DIRECT && width == 128 && k == 3 ? 304 : 16 * 2 * R2_NH; }

// utterances beyond 320 frames: chunks of equal useful length with steps * pad halo rows per side (Res2Args); {1, T} when one workgroup holds it.
// Small batches of the direct form's geometry (B utterances on a chip of many more CUs) are cut into chunks of <= 160 frames for the 5-tile
// kernel whatever their length: one 3 s utterance = three workgroups with half the per-step work each instead of one (a produced row sees
// exactly the rows it sees in the unchunked run, so the bits are the same).
static void res2_chunking(int B, int T, int width, int steps, int k, int dil, int* nchunks, int* useful, bool* small) {
    *nchunks = 1;
    *useful = T;
    *small = false;
    const int halo2 = 2 * steps * (dil * (k - 1) / 2);
    if (R2_DIRECT && width == 128 && k == 3 && R2_SMALL_ROWS - halo2 >= 64 && T > R2_SMALL_ROWS / 2) {
        const int per = R2_SMALL_ROWS - halo2;
        const int n0 = (T + per - 1) / per;
        if ((int64_t)B * n0 <= device_cu_count()) {   // (the chunks fit the chip in one round: 64 x 3 s = 192 workgroups of 59-66 us against 64 of 91 us)
            *useful = T <= R2_SMALL_ROWS ? T : (T + n0 - 1) / n0;
            *nchunks = (T + *useful - 1) / *useful;
            *small = true;
            return;
        }
    }
    if (T <= 16 * 2 * R2_NH) return;
    const int per = res2_local_limit(width, k) - halo2;
    if (per < 64) return;  // (not worth it: the caller falls back to one launch per step)
    const int n0 = (T + per - 1) / per;
    *useful = (T + n0 - 1) / n0;
    *nchunks = (T + *useful - 1) / *useful;   // (rounding `useful` up can save a chunk: no chunk may start at or behind T)
}

bool res2_chain_supported(int T, int width, int steps, int k, int dil) {
    if (!((width == 64 || width == 128) && steps >= 1 && steps <= R2_MAX_STEPS && (k % 2) == 1 && dil * (k - 1) / 2 <= R2_MAXPAD &&
          dil * (k - 1) / 2 < T))
        return false;
    if (T <= 16 * 2 * R2_NH) return true;
    int nchunks, useful;
    bool small;
    res2_chunking(1 << 20, T, width, steps, k, dil, &nchunks, &useful, &small);   // (a batch that never takes the small-batch form)
    return nchunks > 1;
}

int res2_chain_launch(const half_t* x, half_t* y, const half_t* const* w, const float* const* bias, const float* const* scale,
                      const float* const* shift, int B, int T, int C, int width, int steps, int k, int dil, hipStream_t stream) {
    MV_REQUIRE(res2_chain_supported(T, width, steps, k, dil), "res2_chain: unsupported geometry");
    MV_REQUIRE(C == width * (steps + 1), "res2_chain: channels must be (steps + 1) * width");
    Res2Args a;
    a.x = x;
    a.y = y;
    for (int j = 0; j < steps; ++j) {
        a.w[j] = w[j];
        a.bias[j] = bias[j];
        a.scale[j] = scale[j];
        a.shift[j] = shift[j];
    }
    a.T = T;
    a.C = C;
    a.width = width;
    a.steps = steps;
    a.k = k;
    a.dil = dil;
    a.kpad = conv1d_cin_pad(width);
    bool small = false;
    res2_chunking(B, T, width, steps, k, dil, &a.nchunks, &a.useful, &small);
    a.halo = a.nchunks > 1 ? steps * (dil * (k - 1) / 2) : 0;
    const int Tl = a.nchunks > 1 ? (a.useful + 2 * a.halo < T ? a.useful + 2 * a.halo : T) : T;   // most frames a workgroup works on
    const size_t lds = res2_chain_lds_bytes(Tl, width, k);
    const unsigned grid = (unsigned)B * (unsigned)a.nchunks;
    if (small) {
        MV_REQUIRE(Tl <= R2_SMALL_ROWS, "res2_chain: small-batch chunk too long");
        if (MV_SET_MAX_SMEM((res2_chain_kernel<2, true, 5>), R2_SMALL_LDS) != hipSuccess) return fail(MV_ERR_HIP, "res2_chain: cannot reserve LDS");
        MV_LAUNCH((res2_chain_kernel<2, true, 5>), (grid, 1, 1), (R2_THREADS, 1, 1), R2_SMALL_LDS, stream, a);
    } else if (res2_direct(Tl, width, k)) {
        if (MV_SET_MAX_SMEM((res2_chain_kernel<2, true>), lds) != hipSuccess) return fail(MV_ERR_HIP, "res2_chain: cannot reserve LDS");
        MV_LAUNCH((res2_chain_kernel<2, true>), (grid, 1, 1), (R2_THREADS, 1, 1), lds, stream, a);
    } else if (width == 128) {
        if (MV_SET_MAX_SMEM(res2_chain_kernel<2>, lds) != hipSuccess) return fail(MV_ERR_HIP, "res2_chain: cannot reserve LDS");
        MV_LAUNCH(res2_chain_kernel<2>, (grid, 1, 1), (R2_THREADS, 1, 1), lds, stream, a);
    } else {
        if (MV_SET_MAX_SMEM(res2_chain_kernel<1>, lds) != hipSuccess) return fail(MV_ERR_HIP, "res2_chain: cannot reserve LDS");
        MV_LAUNCH(res2_chain_kernel<1>, (grid, 1, 1), (R2_THREADS, 1, 1), lds, stream, a);
    }
    return check_launch("res2_chain_kernel");
}

}  // namespace mv

extern "C" {

int mv_res2net_chain_f16(const void* x, void* y, const void* const* w_packed, const float* const* bias,
                         const float* const* scale, const float* const* shift, int32_t B, int32_t T, int32_t C,
                         int32_t groups, int32_t k, int32_t dilation, mv_stream_t stream) {
    MV_REQUIRE(x != nullptr && y != nullptr && w_packed != nullptr && bias != nullptr && scale != nullptr && shift != nullptr,
               "mv_res2net_chain_f16: null argument");
    MV_REQUIRE(groups >= 2 && C % groups == 0 && B > 0 && T > 0, "mv_res2net_chain_f16: bad geometry");
    const int width = C / groups;
    if (!mv::res2_chain_supported(T, width, groups - 1, k, dilation))
        return mv::fail(MV_ERR_UNSUPPORTED,
                        "mv_res2net_chain_f16: fused chain needs width 64/128, an odd kernel and padding <= 8 (< T)");
    return mv::res2_chain_launch(reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y),
                                 reinterpret_cast<const half_t* const*>(w_packed), bias, scale, shift, B, T, C, width, groups - 1, k,
                                 dilation, static_cast<hipStream_t>(stream));
}

}  // extern "C"
