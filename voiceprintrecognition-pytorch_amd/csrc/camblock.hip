// All CAMDenseTDNNLayers of one CAM++ block (mvector/models/campplus.py:153-181) in ONE launch, one workgroup per utterance walking
// the layers.  Per layer it is the algorithm of cam_dense_layer_kernel (camdense.hip: 1x1 GEMM with both operand streams on LDS-DMA
// rings and the BN1 + ReLU transform in place, bottleneck h in LDS, context gate on the workgroup's own lanes, k = 3 conv from LDS);
// what changes is what happens BETWEEN two layers.  As separate launches a layer costs 24.4 us of which 12.7 are fixed (profiles/r07b
// timeline): every workgroup of a launch starts at the same moment and waits for its parameters (5.6 us) and first stages, and the
// epilogue / context / k = 3 phases (6 us) run with the memory system idle.  Here layer l + 1's first operands travel under layer l's
// tail:
//   * after the stage loop of layer l the rings are idle: x stages 0 and 1 (channels 0..127: never this block's new channels) of layer
//     l + 1 are requested into ring slots 0 / 1 at once, its BN1 tables into registers;
//   * h lives in x slots 2 and 3 without halo rows (the k = 3 conv masks its out-of-range taps); the context's column sums leave the h
//     epilogue's registers (rows of v_add_f32_dpp, [2 time halves][2 segments][128] partial sums: no scratch, no pass over h);
//   * the next layer's BN1 tables go to LDS, its W1 stages 0 / 1 and context parameters are requested when layer l's context phase has
//     consumed its own (same registers), the k = 3 weights after layer l's k = 3 conv;
//   * layer entry (round 5): a COUNTED wait that leaves the y stores and the k = 3 weight requests in flight + barrier, then x stage 2 --
//     the stores only have to be in L2 when the last stage (the one with the new channels) is requested, and the stage loop's own
//     counted waits see to that; the request pattern of the stage loop (stage s requests x(s + 3) and W1(s + 2), five transfers per
//     wave and stage) is the one of cam_dense_layer_kernel from there on.
// The first form of "all layers in one launch" (r05t: the layer loop around the unchanged body, vmcnt(0) + barrier between layers)
// measured 3 % slower than separate launches -- it removed the launch boundary and kept every wait.
// Measured (profiles/r08a, r08b timeline): 23.1 us per layer against 24.4 (CAM++ 2.28 -> 2.25 ms per step).  The layer entry costs 1.4 us
// (was 1.0 + 5.6), the stage loop is unchanged (1.17 us per stage: five transfer issues, the in-place transform and a barrier per 20
// MFMAs of a wave -- issue-bound per CU, not bandwidth-bound: 31 GB/s per CU), the tail grew from 6 to 9 us: all 256 workgroups walk
// the layers in lockstep, so the prefetch of all of them is in flight while all of them issue their parameter loads and y stores.  Things
// the compiler does to such a loop and what keeps it from them: the tail's per-thread addresses are loop-invariant and were hoisted into
// ~130 registers (76 spilled, each reload an s_waitcnt vmcnt(0)): they hang on a per-layer MV_OPAQUE thread id; the layer descriptors
// were read with vector loads (the kernel stores to global memory) whose wait drained the prefetch: they are copied to LDS once and read
// into scalar registers; a select on a loaded table value puts the wait behind the load: index clamp instead; the compiler's own waits for
// the parameter loads are pulled to the layer entry with MV_OPAQUE touches.  Starting the odd workgroups 4 - 20 us late (so that half the
// chip is in its stage loop while the other half is in its tail) changed nothing: 2.23 - 2.27 ms for every delay (r08d).
// Round 3, r10j-r10n: one interleaved schedule per stage (the three-phase stage of rounds 2-3 was its A/B arm: tools/variants): CAM++ 134.5 k ->
// 137.5 k utt/s in one call (1.903 -> 1.862 ms), bit-identical.  Probes on that form (text / macro arms of this file, one call each, wrong results
// on purpose): the x transfers of the loop reading a constant page instead of the concat buffer (same instructions, no x traffic): no change at
// all -- the loop is NOT bound by the 2.34 GB per step its x stream moves (every layer re-reads its utterance's prefix; 32 x 305 KB per XCD do
// not fit an L2), and marking the stages behind the first k as non-temporal (so that the first k x 64 channels of every utterance stay in L2)
// changes nothing either (k = 0..8: 130.6-131.9 k against 131.9-132.6 k); without the transform (its LDS round trip: 20 KB read + 20 KB written
// per stage) -168 us per step of 2083, without the MFMAs -110 us; the transform as v_fma_mixlo/hi_f16 + v_pk_max_f16 (12 instead of 36 vector
// instructions per cell, the same bits: kept) no change.  What is left of a stage (0.6 of 1.17 us) is the barrier, the five transfer requests
// and the fourteen fragment reads of a wave.  W1 for free (no W1 transfers, no W1 fragment reads: the upper bound of the Res2Net chain's
// "weight fragments straight into registers" form, r10q): -4.4 % of the CAM++ step -- not worth the ~48 registers the kernel does not have.
// Round 5 (r14k-r14m, found by READING THE ISA -- tools/isa_audit.py -- after three rounds of timelines): CAM++ 136.0 k -> 153.1 k utt/s, one
// utterance 1326 -> 1166 us, the tail + entry of a layer 28 k -> 19 k ticks.  (1) the k = 3 phase's `if (tile < 10)` around each MFMA (uniform per
// wave, on a value the source had made opaque) had become 36 blocks of ds_read -> s_waitcnt lgkmcnt(0) -> MFMA -> accumulator copies, ~250 cycles
// each = the 10.4 k ticks the r08b timeline blamed on memory traffic: straight-line code with a spare tile for the waves that own two, fragment
// reads in six pipelined groups (4.3 k -> 3.0 k); (2) pointers read out of the layer descriptors in LDS carry no address space, so every
// parameter load was a FLAT load, which counts on lgkmcnt as well: each LDS wait behind one waited for L2 (MV_GLOBAL_PTR); (3) BN1 tables loaded
// under one `if (more)` and stored under another: the wait-count insertion sees the path on which they are never waited for, carried their
// registers as pending around the layer loop and put s_waitcnt vmcnt(0) in front of a fragment read IN EVERY STAGE once tracked loads were allowed
// to stay in flight across the entry (unconditional load + store); their store in front of the k = 3 phase instead of behind it (it was an
// s_waitcnt vmcnt(0) = a drain of the stores and weight requests at the end of every layer, 3.3 k ticks); (4) the counted entry wait, with the
// y stores as inline assembly (a tracked store beside tracked loads makes the compiler treat the counter as out of order: vmcnt(0) everywhere);
// (5) row sums as explicit v_add_f32_dpp chains (the SLP vectoriser had paired the additions into v_pk_add_f32, which has no DPP form).
// Measured and NOT kept: the next layer's W1 stages requested at the start of the tail (-0.3 %), its context parameters behind the k = 3 phase
// (requested behind the k = 3 phase instead: -2 %, 23 requests in one burst); the LDS form of the column sums (equal; removed).
#include "kernels.h"

namespace mv {

constexpr int CB_THREADS = 512;
constexpr int CB_TT = 10;                           // time tiles of 16 frames: T2 <= 160
constexpr int CB_ROWS = CB_TT * 16;
constexpr int CB_BN = 128;                          // bottleneck channels
constexpr int CB_G = 32;                            // growth rate
constexpr int CB_XS_BYTES = CB_ROWS * 128;          // one x stage: [160 rows][64 fp16]
constexpr int CB_WS_BYTES = CB_BN * 128;            // one W1 stage: [128 rows][64 fp16]
constexpr int CB_RING = 3, CB_XRING = 4;
constexpr int CB_MAX_SEG = 2;
constexpr int CB_MAX_CIN = 1024;                    // BN1 tables of two layers live in LDS
constexpr int CB_H_OFF = 2 * CB_XS_BYTES;           // h = x slots 2, 3: [160 rows][128 fp16], no halo rows
static_assert(CB_ROWS * CB_BN * 2 == 2 * CB_XS_BYTES, "h fills exactly two x slots");
constexpr int CB_F_OFF = CB_XRING * CB_XS_BYTES + CB_RING * CB_WS_BYTES;  // fp32 area behind the rings
constexpr int CB_F_CTX = 0, CB_F_G1 = CB_F_CTX + CB_MAX_SEG * CB_BN, CB_F_GATE = CB_F_G1 + CB_MAX_SEG * 64,
              CB_F_PART = CB_F_GATE + CB_MAX_SEG * CB_G,              // [2 time halves][CB_MAX_SEG][CB_BN] column sums of h
              CB_F_TAB = CB_F_PART + 2 * CB_MAX_SEG * CB_BN,          // [2 layers][scale | shift][CB_MAX_CIN]
              CB_F_END = CB_F_TAB + 4 * CB_MAX_CIN;
constexpr int CB_MAX_LAYERS = 24;
constexpr size_t CB_DESC_OFF = CB_F_OFF + CB_F_END * sizeof(float) + 1024;   // behind the KiB that swallows the padding transfers
constexpr size_t CB_LDS_BYTES = CB_DESC_OFF + CB_MAX_LAYERS * sizeof(MvCamLayerDesc);
static_assert(CB_LDS_BYTES <= 160 * 1024, "cam block kernel: LDS");

__device__ __attribute__((aligned(256))) const unsigned char g_cb_zero_page[256] = {0};

struct CamBlockArgs {
    half_t* x;             // [B, T2, ldx]: the block's concat buffer
    int64_t ldx;
    const MvCamLayerDesc* layers;  // device array
    int nlayers, T2, dil, seg_len;
};

__global__ __launch_bounds__(CB_THREADS) void cam_dense_block_kernel(CamBlockArgs a) {
    MV_DYN_SMEM(smem);
    char* xs = smem;
    char* ws = xs + CB_XRING * CB_XS_BYTES;
    char* hbuf = xs + CB_H_OFF;
    float* fsm = reinterpret_cast<float*>(smem + CB_F_OFF);
    float* ctx = fsm + CB_F_CTX;
    float* g1 = fsm + CB_F_G1;
    float* gate = fsm + CB_F_GATE;
    float* hsum = fsm + CB_F_PART;
    float* tabs = fsm + CB_F_TAB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x;
    const int T2 = a.T2;
    half_t* xb = a.x + (int64_t)b * T2 * a.ldx;
    const int cw = wave & 3, th = wave >> 2;
    const int lrow = lane >> 3, kc = (lane & 7) ^ lrow;
    const unsigned xs_addr = lds_addr(xs), ws_addr = lds_addr(ws);
    const unsigned dump_addr = lds_addr(smem + CB_F_OFF + CB_F_END * sizeof(float));
    const half_t* zero = reinterpret_cast<const half_t*>(g_cb_zero_page);
    const int wave_u = MV_UNIFORM(wave);
    // The per-layer descriptors are copied into LDS once and read from there into SCALAR registers: read from global memory the
    // compiler uses vector loads (the kernel also stores to global memory), and its wait for a descriptor field of layer l + 1 -- an
    // s_waitcnt vmcnt(0) in the middle of the tail -- drained the operand requests that had just gone out.
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(a.layers);
        unsigned* dst = reinterpret_cast<unsigned*>(smem + CB_DESC_OFF);
        for (int i = tid; i < a.nlayers * (int)(sizeof(MvCamLayerDesc) / 4); i += CB_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    auto layer_desc = [&](int l) {
        const unsigned* p = reinterpret_cast<const unsigned*>(smem + CB_DESC_OFF) + l * (int)(sizeof(MvCamLayerDesc) / 4);
        union {
            MvCamLayerDesc d;
            unsigned w[sizeof(MvCamLayerDesc) / 4];
        } u;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(MvCamLayerDesc) / 4); ++i) u.w[i] = (unsigned)MV_UNIFORM((int)p[i]);
        return u.d;
    };

    // ---- operand requests of a layer: every wave issues exactly 3 x transfers and 2 W1 transfers per stage (stages beyond the last one
    // and the x transfers 20..23 read a constant page into the dump KiB), so the waits can be counted ----
    auto issue_x_from = [&](int s, int nst, int wave_uu, int lrow_, int kc_) {
        const bool real = s < nst;
        const unsigned dst = xs_addr + (unsigned)((s & (CB_XRING - 1)) * CB_XS_BYTES);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tr = wave_uu + 8 * u;
            int row = tr * 8 + lrow_;
            row = row < T2 ? row : T2 - 1;
            const bool live = real && tr < CB_ROWS / 8;  // uniform
            glds16_untracked(live ? xb + (int64_t)row * a.ldx + s * 64 + kc_ * 8 : zero, live ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    auto issue_x = [&](int s, int nst) { issue_x_from(s, nst, wave_u, lrow, kc); };
    auto issue_w = [&](int s, int nst, const half_t* w1, int cin_pad) {
        const bool real = s < nst;
        const unsigned dst = ws_addr + (unsigned)((s % CB_RING) * CB_WS_BYTES);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tr = wave_u * 2 + u;
            const int co = tr * 8 + lrow;
            glds16_untracked(real ? w1 + (int64_t)co * cin_pad + s * 64 + kc * 8 : zero, real ? dst + (unsigned)(tr * 1024) : dump_addr);
        }
    };
    // BN1 + ReLU in fp32, in place (camdense.hip): rows >= T2 and channels >= cin become zero
    const int xchunk = tid & 7, xrow0 = tid >> 3;
    auto transform = [&](int s, int cin, const float* lbn_s, const float* lbn_t) {
        const int c = s * 64 + xchunk * 8;
        const bool live = c < cin;
        const float4v s0 = *reinterpret_cast<const float4v*>(lbn_s + c), s1 = *reinterpret_cast<const float4v*>(lbn_s + c + 4);
        const float4v t0 = *reinterpret_cast<const float4v*>(lbn_t + c), t1 = *reinterpret_cast<const float4v*>(lbn_t + c + 4);
        char* tile = xs + (s & (CB_XRING - 1)) * CB_XS_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int row = xrow0 + 64 * p;
            if (row < CB_ROWS) {
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
    // BN1 tables of a layer: two floats of scale and of shift per thread (cin_pad <= 1024) -- loaded early into registers, stored late
    auto load_tables = [&](const MvCamLayerDesc& L, float (&ts)[2], float (&tt)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CB_THREADS;
            const int ic = i < L.cin ? i : L.cin - 1;  // always a load, never a select on its result: the compiler waits where a value is first USED
            ts[j] = *MV_GLOBAL_PTR(float, L.bn1_s + ic);   // (pointers out of the descriptors in LDS carry no address space: without the cast every parameter
            tt[j] = *MV_GLOBAL_PTR(float, L.bn1_t + ic);   //  load is a FLAT load, which also counts on lgkmcnt -- the next LDS wait then waits for L2)
        }
    };
    auto store_tables = [&](int buf, int cin, const float (&ts)[2], const float (&tt)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CB_THREADS;
            tabs[buf * 2 * CB_MAX_CIN + i] = i < cin ? ts[j] : 0.0f;
            tabs[buf * 2 * CB_MAX_CIN + CB_MAX_CIN + i] = i < cin ? tt[j] : 0.0f;
        }
    };
    auto h_off = [&](int row, int chunk) { return row * (CB_BN * 2) + ((chunk ^ (row & 15)) << 4); };

    // ---- parameters of a layer's later phases (registers) ----
    float4v e_bn2s[2], e_bn2t[2], e_wa[4], e_wb;
    float e_ba, e_bb;
    half8v e_wl[3][4];
    auto load_ctx_params = [&](const MvCamLayerDesc& L) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            e_bn2s[mi] = *MV_GLOBAL_PTR(float4v, L.bn2_s + (cw * 2 + mi) * 16 + 4 * fg);
            e_bn2t[mi] = *MV_GLOBAL_PTR(float4v, L.bn2_t + (cw * 2 + mi) * 16 + 4 * fg);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e_wa[u] = *MV_GLOBAL_PTR(float4v, L.wa + (tid >> 3) * CB_BN + (tid & 7) * 16 + 4 * u);
        e_wb = *MV_GLOBAL_PTR(float4v, L.wb + (tid >> 4) * 64 + (tid & 15) * 4);
        e_ba = *MV_GLOBAL_PTR(float, L.ba + (tid >> 3));
        e_bb = *MV_GLOBAL_PTR(float, L.bb + (tid >> 4));
        MV_VM_LOADS(11);
    };
    auto load_wl = [&](const MvCamLayerDesc& L) {
        const half_t* wrow = L.wl + (int64_t)((wave & 1) * 16 + fr) * 3 * CB_BN + 8 * fg;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e_wl[tap][kk] = *MV_GLOBAL_PTR(half8v, wrow + tap * CB_BN + kk * 32);
        MV_VM_LOADS(12);   // (the layer entry's counted wait leaves these twelve and the y stores in flight)
    };

    // ---- first layer: the start-up of cam_dense_layer_kernel ----
    {
        const MvCamLayerDesc L0 = layer_desc(0);
        const int nst0 = L0.cin_pad / 64;
        float ts[2], tt[2];
        load_tables(L0, ts, tt);
        store_tables(0, L0.cin, ts, tt);
        load_ctx_params(L0);
        load_wl(L0);
        issue_x(0, nst0);
        issue_x(1, nst0);
        issue_w(0, nst0, L0.w1, L0.cin_pad);
        issue_w(1, nst0, L0.w1, L0.cin_pad);
        __syncthreads();  // tables visible
    }

#pragma unroll 1
    for (int l = 0; l < a.nlayers; ++l) {
        const MvCamLayerDesc L = layer_desc(l);
        const int nst = L.cin_pad / 64;
        const float* lbn_s = tabs + (l & 1) * 2 * CB_MAX_CIN;
        const float* lbn_t = lbn_s + CB_MAX_CIN;
        // ---- layer entry: x(0), x(1), W1(0), W1(1) have been requested (by the previous layer's tail or by the start-up); everything
        // this wave has in flight -- them, the parameter loads, the previous layer's y stores -- is waited for, then x(2) goes out ----
        // The youngest vector-memory operations of this wave are the previous layer's y stores and the twelve k = 3 weight requests behind them; what
        // the entry needs -- x(0), x(1), W1(0), W1(1), the context parameters -- is older.  The stores have to be in L2 before ANY wave requests the x
        // stage that holds the previous layer's 32 channels: that is the LAST stage, requested at stage nst - 4 behind that stage's counted wait (which
        // covers everything older than the last five transfers) and barrier.  So with more than four stages the entry waits for all but those
        // youngest operations, and stage 0 (whose operands the entry has seen land) waits for nothing.  (Loads and stores share vmcnt and retire in issue
        // order on gfx9 -- the compiler relies on the same: `load; store; store; use of the load` compiles to s_waitcnt vmcnt(2).)
        const bool lazy = l > 0 && nst > 4;   // uniform
        if (lazy) {
            int n_st = 0;   // y stores this wave issued: its time tiles with a frame below T2 (phase C)
#pragma unroll
            for (int j = 0; j < 3; ++j) n_st += ((wave_u >> 1) + 4 * j < CB_TT && ((wave_u >> 1) + 4 * j) * 16 < T2) ? 1 : 0;
            constexpr int NP = 12;   // parameter requests behind the stores: the twelve k = 3 weight loads
            if (n_st == 3) {   // (waits the compiler sees: it has the k = 3 weight and context parameter loads on its scoreboard)
                wait_vm_seen<NP + 3>();
            } else if (n_st == 2) {
                wait_vm_seen<NP + 2>();
            } else if (n_st == 1) {
                wait_vm_seen<NP + 1>();
            } else {
                wait_vm_seen<NP>();
            }
        } else {
            wait_vm_seen<0>();
        }
        __syncthreads();   // every wave's stores are in L2, the tables of this layer are in LDS, h (x slots 2, 3) is dead
        issue_x(2, nst);
        transform(0, L.cin, lbn_s, lbn_t);
        float4v acc[2][5];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        // ---- phase A: h = ReLU(BN2(W1 . ReLU(BN1(x)))) ----
#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            // W1(s) and x(s+1) have landed: younger are x(s+2) [3 transfers] and, from stage 1 on, W1(s+1) [2]
            if (s == 0) {
                if (!lazy) wait_vm<3>();   // (W1(1) was requested before x(2) at the layer boundary: older than the three transfers that may stay)
            } else {
                wait_vm<5>();
            }
            lds_barrier();  // ... in every wave; x(s) is transformed; every wave is done with x(s-1) and W1(s-1), whose slots are requested now
            // One schedule for the whole stage (r10j): the five transfer requests of x(s+3) / W1(s+2), the 20 MFMAs on x(s) / W1(s) and the in-place
            // BN1 + ReLU of x(s+1) are interleaved -- requests and transform cells between groups of MFMAs -- instead of running as three phases
            // that every wave of the workgroup enters at the same moment (requests ~750 cycles, MFMAs ~400, transform ~600 of a 2300-cycle stage).
            // The transform of the stage behind the last one works on a dead slot (its result is never read).
            const char* wt = ws + (s % CB_RING) * CB_WS_BYTES;
            const char* xt = xs + (s & (CB_XRING - 1)) * CB_XS_BYTES;
            const bool xreal = s + 3 < nst, wreal = s + 2 < nst;
            const unsigned xdst = xs_addr + (unsigned)(((s + 3) & (CB_XRING - 1)) * CB_XS_BYTES);
            const unsigned wdst = ws_addr + (unsigned)(((s + 2) % CB_RING) * CB_WS_BYTES);
            auto dma_x = [&](int u) {
                const int tr = wave_u + 8 * u;
                int row = tr * 8 + lrow;
                row = row < T2 ? row : T2 - 1;
                const bool live = xreal && tr < CB_ROWS / 8;  // uniform
                const half_t* src = live ? xb + (int64_t)row * a.ldx + (s + 3) * 64 + kc * 8 : zero;
                const unsigned dst = live ? xdst + (unsigned)(tr * 1024) : dump_addr;
                glds16_untracked(src, dst);
            };
            auto dma_w = [&](int u) {
                const int tr = wave_u * 2 + u;
                const int co = tr * 8 + lrow;
                glds16_untracked(wreal ? L.w1 + (int64_t)co * L.cin_pad + (s + 2) * 64 + kc * 8 : zero, wreal ? wdst + (unsigned)(tr * 1024) : dump_addr);
            };
            // transform cells of x(s+1)
            const int tc = (s + 1) * 64 + xchunk * 8;
            const bool tlive = tc < L.cin;
            const float4v ts0 = *reinterpret_cast<const float4v*>(lbn_s + tc), ts1 = *reinterpret_cast<const float4v*>(lbn_s + tc + 4);
            const float4v tt0 = *reinterpret_cast<const float4v*>(lbn_t + tc), tt1 = *reinterpret_cast<const float4v*>(lbn_t + tc + 4);
            char* ttile = xs + ((s + 1) & (CB_XRING - 1)) * CB_XS_BYTES;
            auto cell_ptr = [&](int p) {
                const int row = xrow0 + 64 * p;
                return reinterpret_cast<half8v*>(ttile + row * 128 + ((xchunk ^ (row & 7)) << 4));
            };
            auto cell_math = [&](int p, const half8v& r) {
                const int row = xrow0 + 64 * p;
                // fp32 FMA on the fp16 value, ONE rounding to fp16, then ReLU on the packed halves -- the same result as ReLU in fp32 before the
                // rounding (rounding is monotonic and keeps zero), but a form the compiler maps to v_fma_mixlo/hi_f16 + v_pk_max_f16: 12
                // instructions per 16-byte cell instead of 36 (cvt, fma, max, cvt per element)
                half8v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (half_t)__builtin_fmaf((float)r[e], ts0[e], tt0[e]);
                    o[4 + e] = (half_t)__builtin_fmaf((float)r[4 + e], ts1[e], tt1[e]);
                }
                {
                    const half8v z8 = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                    o = __builtin_elementwise_max(o, z8);
                }
                if (!(row < T2 && tlive)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)0.0f;
                }
                return o;
            };
            const bool cell2 = xrow0 + 128 < CB_ROWS;
            half8v af[2], bf[5], r0, r1, r2 = half8v{}, o0, o1;
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
            r0 = *cell_ptr(0);
            dma_x(0);
            mm(0);
            mm(1);
            o0 = cell_math(0, r0);
            dma_x(1);
            mm(2);
            mm(3);
            *cell_ptr(0) = o0;
            r1 = *cell_ptr(1);
            dma_x(2);
            mm(4);
            load_frags(1);
            o1 = cell_math(1, r1);
            dma_w(0);
            mm(0);
            mm(1);
            *cell_ptr(1) = o1;
            if (cell2) r2 = *cell_ptr(2);
            dma_w(1);
            mm(2);
            mm(3);
            mm(4);
            if (cell2) *cell_ptr(2) = cell_math(2, r2);
        }
        wait_vm<0>();   // only padding transfers are left: nothing may still be landing when the rings are reused
        // the parameter loads of this layer are the only vector loads the compiler tracks; touching what they deliver puts ITS wait for them
        // here, behind ours -- left where the values are first used (epilogue, context phase, k = 3 conv) it would be an s_waitcnt vmcnt(0)
        // in the middle of the tail, draining the next layer's operand requests that have just gone out
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
        lds_barrier();  // every wave is done with both rings
        // The tail's per-thread LDS / global addresses are derived from a thread id the optimiser must treat as new in every layer:
        // otherwise it hoists all of them out of the layer loop (they are loop-invariant) into ~130 long-lived registers and spills.
        int tid_t = tid;
        MV_OPAQUE(tid_t);
        const int lane_t = tid_t & 63, wave_t = tid_t >> 6;
        const int fr_t = lane_t & 15, fg_t = lane_t >> 4, cw_t = wave_t & 3, th_t = wave_t >> 2;
        // ---- the next layer's first operands travel under this layer's tail ----
        const bool more = l + 1 < a.nlayers;  // uniform
        MvCamLayerDesc Ln = L;
        float nts[2] = {0.0f, 0.0f}, ntt[2] = {0.0f, 0.0f};
        if (more) {
            Ln = layer_desc(l + 1);
            const int nstn = Ln.cin_pad / 64;
            const int lrow_t = lane_t >> 3, wave_ut = MV_UNIFORM(wave_t);   // (row addresses from the per-layer thread id: hoisted out of the layer loop they were spilled,
            issue_x_from(0, nstn, wave_ut, lrow_t, (lane_t & 7) ^ lrow_t);   //  and the reload's s_waitcnt vmcnt(0) sat behind the first of these requests)
            issue_x_from(1, nstn, wave_ut, lrow_t, (lane_t & 7) ^ lrow_t);
        }
        // UNCONDITIONAL (the last layer fetches its own tables again, into the idle buffer): loaded under `if (more)` and stored under a second
        // `if (more)`, the compiler's wait-count insertion sees a path on which the four loads are never waited for, carries their registers as
        // pending around the layer loop, and protects the fragment read that reuses one of them with an s_waitcnt vmcnt(0) in every stage
        load_tables(Ln, nts, ntt);
        // epilogue A: BN2 + ReLU -> h (fp16, swizzled 16-byte chunks: chunk ^= row & 15); frames >= T2 are zero
        for (int i = T2 * CB_BN * 2 + tid_t * 16; i < CB_ROWS * CB_BN * 2; i += CB_THREADS * 16)
            *reinterpret_cast<float4v*>(hbuf + i) = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        // column sums of h for the context, taken from the values on their way to LDS (before their rounding to fp16): per lane over its five time
        // tiles, split at the segment boundary by 0 / 1 weights, then over the 16 frames of a tile with DPP row sums; frames >= T2 do not count
        float m0[5], m1[5];
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) {
            const int t = (th_t * 5 + ni) * 16 + fr_t;
            m0[ni] = t < T2 && t < a.seg_len ? 1.0f : 0.0f;
            m1[ni] = t < T2 && t >= a.seg_len ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co = (cw_t * 2 + mi) * 16 + 4 * fg_t;
            const float4v sc = e_bn2s[mi], sh = e_bn2t[mi];
            float4v cs0 = float4v{0.0f, 0.0f, 0.0f, 0.0f}, cs1 = cs0;
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) {
                const int t = (th_t * 5 + ni) * 16 + fr_t;
                half4v hv;
                float4v hf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hf[r] = fmed3(acc[mi][ni][r] * sc[r] + sh[r], 0.0f, 65504.0f);   // ReLU and the fp16 saturation in one clamp
                    hv[r] = (half_t)hf[r];
                }
                if (t < T2) *reinterpret_cast<half4v*>(hbuf + h_off(t, co >> 3) + (co & 7) * 2) = hv;
                cs0 += hf * m0[ni];   // (the fp32 value: the reference's context is the mean of an unrounded h)
                cs1 += hf * m1[ni];
            }
            row16_sum8(cs0, cs1);
            if (fr_t == 0) {
                *reinterpret_cast<float4v*>(hsum + (th_t * CB_MAX_SEG + 0) * CB_BN + co) = cs0;
                *reinterpret_cast<float4v*>(hsum + (th_t * CB_MAX_SEG + 1) * CB_BN + co) = cs1;
            }
        }
        __syncthreads();

        // ---- phase B: context gate per 100-frame segment ----
        {
            const float inv_t = 1.0f / (float)T2;
            const int len0 = a.seg_len < T2 ? a.seg_len : T2, len1 = T2 - len0;
            const float inv_len[CB_MAX_SEG] = {1.0f / (float)len0, len1 > 0 ? 1.0f / (float)len1 : 0.0f};
            const float has_seg[CB_MAX_SEG] = {1.0f, len1 > 0 ? 1.0f : 0.0f};   // a segment without frames has context 0 (no rows of y read its gate)
            {   // g1 = ReLU(Wa ctx + ba): 8 threads per output row, both segments
                const int j = tid_t >> 3, part8 = tid_t & 7;
#pragma unroll
                for (int sg = 0; sg < CB_MAX_SEG; ++sg) {
                    float v = 0.0f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // ctx = mean over the utterance + mean over the segment, from the four partial sums of a channel
                        const int c = part8 * 16 + 4 * u;
                        const float4v own = *reinterpret_cast<const float4v*>(hsum + (0 * CB_MAX_SEG + sg) * CB_BN + c) +
                                            *reinterpret_cast<const float4v*>(hsum + (1 * CB_MAX_SEG + sg) * CB_BN + c);
                        const float4v oth = *reinterpret_cast<const float4v*>(hsum + (0 * CB_MAX_SEG + 1 - sg) * CB_BN + c) +
                                            *reinterpret_cast<const float4v*>(hsum + (1 * CB_MAX_SEG + 1 - sg) * CB_BN + c);
                        float4v c4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) c4[e] = has_seg[sg] * fmaf(own[e] + oth[e], inv_t, own[e] * inv_len[sg]);
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
            {   // gate = sigmoid(Wb g1 + bb): 16 threads per output row (one DPP row), both segments
                const int co = tid_t >> 4, part16 = tid_t & 15;
#pragma unroll
                for (int sg = 0; sg < CB_MAX_SEG; ++sg) {
                    const float4v g4 = *reinterpret_cast<const float4v*>(g1 + sg * 64 + part16 * 4);
                    float v = e_wb[0] * g4[0];
                    v = fmaf(e_wb[1], g4[1], v);
                    v = fmaf(e_wb[2], g4[2], v);
                    v = fmaf(e_wb[3], g4[3], v);
                    v = row16_sum(v);
                    if (part16 == 0) gate[sg * CB_G + co] = 1.0f / (1.0f + expf(-(v + e_bb)));
                }
            }
            __syncthreads();
        }
        // The next layer's BN1 tables (requested at the start of this tail, read by its transform() behind the entry barrier) go to LDS HERE: behind
        // the k = 3 phase the compiler's wait for them was an s_waitcnt vmcnt(0) that drained this layer's y stores and the twelve k = 3 weight
        // requests at the end of every layer (3.3 k ticks in the r08b timeline); here only the x requests of the tail's start are older.
        store_tables((l + 1) & 1, Ln.cin, nts, ntt);
        // The W ring is free again: the next layer's first two W1 stages (the layer entry waits for them).  Its context parameters take this layer's
        // registers here (behind the k = 3 phase, as one burst with the k = 3 weights, measured -2 %: r14m).
        if (more) {
            load_ctx_params(Ln);
            const int nstn = Ln.cin_pad / 64;
            issue_w(0, nstn, Ln.w1, Ln.cin_pad);
            issue_w(1, nstn, Ln.w1, Ln.cin_pad);
        }

        // ---- phase C: y = conv_k3(h) * gate -> channels [cin, cin + 32) of x; taps outside [0, T2) are the conv's zero padding ----
        {
            // channel tile ct, time tiles tg, tg + 4, tg + 8.  Waves of time groups 2 / 3 own two tiles: their third accumulator repeats tile 9 and is never
            // stored -- twelve spare MFMAs instead of a uniform branch per (tap, K step, tile): with the branches the compiler gave every one of the 36 MFMAs
            // its own block (ds_read, s_waitcnt lgkmcnt(0), MFMA, copies of the accumulators: ~250 cycles each, the 10.4 k ticks of the r08b timeline)
            const int ct = wave_t & 1, tg = wave_t >> 1;
            const bool third = tg + 8 < CB_TT;
            int trow[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) trow[j] = (j < 2 || third ? tg + 4 * j : CB_TT - 1) * 16 + fr_t;
            float4v yc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) yc[j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
            const half8v zero8 = half8v{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
            // six groups of (tap, two K steps) x three tiles: the six fragment reads of group g + 1 are in flight under the six MFMAs of group g (left
            // to itself the compiler reads two fragments at a time into the same eight registers: eighteen LDS round trips one behind the other)
            half8v fq[2][6];
            auto read_group = [&](int g, half8v (&f)[6]) {
                const int tap = g >> 1;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int r = trow[j] + (tap - 1) * a.dil;
                        const int rc = r < 0 ? 0 : (r < CB_ROWS ? r : CB_ROWS - 1);
                        f[h * 3 + j] = *reinterpret_cast<const half8v*>(hbuf + h_off(rc, ((g & 1) * 2 + h) * 4 + fg_t));
                    }
            };
            read_group(0, fq[0]);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (g + 1 < 6) read_group(g + 1, fq[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int tap = g >> 1;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int r = trow[j] + (tap - 1) * a.dil;
                        const bool in = r >= 0 && r < CB_ROWS;   // rows T2 .. 159 of h are zero
                        const half8v bfr = in ? fq[g & 1][h * 3 + j] : zero8;
                        yc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e_wl[tap][(g & 1) * 2 + h], bfr, yc[j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int co = ct * 16 + 4 * fg_t;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = trow[j];
                if ((j < 2 || third) && t < T2) {
                    const float* gt = gate + (t >= a.seg_len ? CB_G : 0) + co;   // (two segments at most)
                    half4v hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = (half_t)fmed3(yc[j][r] * gt[r], -65504.0f, 65504.0f);
                    global_store8_untracked(xb + (int64_t)t * a.ldx + L.cin + co, hv);   // (beside tracked loads a tracked store costs an s_waitcnt vmcnt(0) in the stage loop)
                }
                if ((j < 2 || third) && trow[j] - fr_t < T2) MV_VM_LOADS(1);   // (a store the wave issued: the layer entry counts it)
            }
        }
        if (more) {
            load_wl(Ln);                       // the k = 3 weights of this layer are consumed (twelve requests)
        }
    }
}

bool cam_dense_block_supported(int T2, int c_in, int c_out, int bottleneck, int growth, int dil, int seg_len) {
    return (c_out - c_in) / CB_G <= CB_MAX_LAYERS && bottleneck == CB_BN && growth == CB_G && T2 >= 1 && T2 <= CB_ROWS && c_in >= 128 && c_in % 32 == 0 && c_out <= CB_MAX_CIN + CB_G &&
           dil >= 1 && dil <= 2 && seg_len > 0 && (T2 + seg_len - 1) / seg_len <= CB_MAX_SEG;
}

int cam_dense_block_launch(half_t* x, int64_t ldx, int B, int T2, const MvCamLayerDesc* layers_dev, int nlayers, int dil, int seg_len,
                           hipStream_t stream) {
    MV_REQUIRE(x != nullptr && layers_dev != nullptr && B > 0 && nlayers > 0 && nlayers <= CB_MAX_LAYERS, "cam_dense_block: bad argument");
    MV_REQUIRE(T2 >= 1 && T2 <= CB_ROWS && dil >= 1 && dil <= 2 && seg_len > 0 && (T2 + seg_len - 1) / seg_len <= CB_MAX_SEG,
               "cam_dense_block: unsupported geometry");
    MV_REQUIRE((ldx % 8) == 0, "cam_dense_block: rows must be 16-byte aligned");
    static DeviceOnce smem_set;   // (per device: the attribute belongs to the current device's code object)
    int smem_set_slot;
    if (device_once_pending(smem_set, &smem_set_slot)) {
        if (MV_SET_MAX_SMEM(cam_dense_block_kernel, CB_LDS_BYTES) != hipSuccess) return fail(MV_ERR_HIP, "cam_dense_block: cannot reserve LDS");
        device_once_done(smem_set, smem_set_slot);
    }
    CamBlockArgs a;
    a.x = x;
    a.ldx = ldx;
    a.layers = layers_dev;
    a.nlayers = nlayers;
    a.T2 = T2;
    a.dil = dil;
    a.seg_len = seg_len;
    MV_LAUNCH(cam_dense_block_kernel, ((unsigned)B, 1, 1), (CB_THREADS, 1, 1), CB_LDS_BYTES, stream, a);
    return check_launch("cam_dense_block_kernel");
}

}  // namespace mv
