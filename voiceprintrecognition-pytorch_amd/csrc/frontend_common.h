// Shared pieces of the two front-end kernels (fbank.hip, melspec.hip): complex helpers on float2 vectors, the in-register
// 16-point FFT, the matrix-pipe mel stage primitives and the host-side planner of the banded mel walk.
#pragma once
#include "common.h"

#include <vector>

namespace mv {

// Complex values {re, im} as a plain struct: every operation is scalar fp32 VALU work (v_add / v_mul / v_fma at full rate).  The float2-vector spelling
// (packed v_pk_* ops: half the instructions, identical time -- a packed op issues at half rate; r03c, r12d) was this file's A/B arm through round 5.
struct alignas(8) cplx {
    float re, im;
    __device__ __forceinline__ float& operator[](int i) { return i == 0 ? re : im; }
    __device__ __forceinline__ float operator[](int i) const { return i == 0 ? re : im; }
};
__device__ __forceinline__ cplx cmake(float r, float i) { return cplx{r, i}; }
__device__ __forceinline__ cplx operator+(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx operator-(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx operator-(cplx a) { return cplx{-a.re, -a.im}; }
__device__ __forceinline__ cplx& operator+=(cplx& a, cplx b) { a = a + b; return a; }
__device__ __forceinline__ cplx cswap(cplx a) { return cplx{a.im, a.re}; }
__device__ __forceinline__ cplx cscale(cplx a, float s) { return cplx{a.re * s, a.im * s}; }
__device__ __forceinline__ cplx cconj(cplx a) { return cplx{a.re, -a.im}; }
__device__ __forceinline__ cplx mul_mi(cplx a) { return cplx{a.im, -a.re}; }  // a * (-i)
// a * (c - i s) = (re c + im s) + i (im c - re s)
__device__ __forceinline__ cplx cmul_conjtw(cplx a, float c, float s) { return cplx{fmaf(a.im, s, a.re * c), fmaf(-a.re, s, a.im * c)}; }

// 8-byte load of two consecutive floats as a complex value; elementwise product (window taps on {x[2n], x[2n+1]})
__device__ __forceinline__ cplx cload(const float* p) {
    const float2v v = *reinterpret_cast<const float2v*>(p);
    return cmake(v[0], v[1]);
}
__device__ __forceinline__ cplx cmul_elem(cplx a, float w0, float w1) { return cmake(a[0] * w0, a[1] * w1); }

// multiply by W16^M = exp(-2 pi i M / 16), M compile-time
template <int M>
__device__ __forceinline__ cplx mul_w16(cplx a) {
    constexpr int m = M & 15;
    if constexpr (m == 0) return a;
    if constexpr (m == 4) return mul_mi(a);
    if constexpr (m == 8) return -a;
    if constexpr (m == 12) return -mul_mi(a);
    constexpr float C[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                             -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
                             0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
    constexpr float S[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                             -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
    if constexpr (m == 2) return cscale(a + mul_mi(a), C[2]);     // (1 - i) / sqrt 2
    if constexpr (m == 6) return cscale(mul_mi(a) - a, C[2]);     // (-1 - i) / sqrt 2
    return cmul_conjtw(a, C[m], S[m]);
}

__device__ __forceinline__ void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3) {
    const cplx t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = t1 + t3;  // t1 - i (a1 - a3)
    a3 = t1 - t3;  // t1 + i (a1 - a3)
}

// forward 16-point DFT, natural order in and out:  X[k] = sum_n x[n] exp(-2 pi i n k / 16)
__device__ __forceinline__ void fft16(cplx (&x)[16]) {
    // n = 4*n1 + n2, k = k1 + 4*k2
    cplx y[4][4];  // [n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        cplx a0 = x[n2], a1 = x[4 + n2], a2 = x[8 + n2], a3 = x[12 + n2];
        dft4(a0, a1, a2, a3);
        y[n2][0] = a0;
        y[n2][1] = a1;
        y[n2][2] = a2;
        y[n2][3] = a3;
    }
    y[1][1] = mul_w16<1>(y[1][1]);
    y[1][2] = mul_w16<2>(y[1][2]);
    y[1][3] = mul_w16<3>(y[1][3]);
    y[2][1] = mul_w16<2>(y[2][1]);
    y[2][2] = mul_w16<4>(y[2][2]);
    y[2][3] = mul_w16<6>(y[2][3]);
    y[3][1] = mul_w16<3>(y[3][1]);
    y[3][2] = mul_w16<6>(y[3][2]);
    y[3][3] = mul_w16<9>(y[3][3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        cplx a0 = y[0][k1], a1 = y[1][k1], a2 = y[2][k1], a3 = y[3][k1];
        dft4(a0, a1, a2, a3);
        x[k1] = a0;
        x[k1 + 4] = a1;
        x[k1 + 8] = a2;
        x[k1 + 12] = a3;
    }
}

__device__ __forceinline__ float fb_log2(float x) { return log2_fast(x); }                       // v_log_f32, normal inputs only
__device__ __forceinline__ float4v fb_mfma4(float a, float b, float4v c) { return mfma_4x4x1(a, b, c); }  // 16 x (4 x 4 x 1) outer products

// One ds_read_b64 per element: pairs of them would be merged into ds_read2_b64, which moves 128 B per LDS clock where
// ds_read_b64 moves 256 (MI355X_MICROARCH.md, LDS table); a volatile access is left alone by the merger.
__device__ __forceinline__ cplx lds_read_single(const cplx* p) {
    const float2v v = lds_load_unmerged(reinterpret_cast<const float2v*>(p));
    return cmake(v[0], v[1]);
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108;

// ---- banded mel stage on v_mfma_f32_4x4x1 -------------------------------------------------------------------------------
// One MFMA pass = 16 blocks x (4 frames x 4 adjacent filters); every block walks `steps` consecutive bins of the power row
// from its own start (a multiple of 4, so operands are 16-byte LDS reads; start + steps <= row_len keeps the walk inside
// the row).  A pass with <= 8 (<= 4) filter groups gives each group 2 (4) adjacent blocks that split its bin range: wide
// high-frequency triangles then cost a quarter of the steps and the kernel adds the partial sums across lanes.
constexpr int MEL_MAX_PASSES = 2;

struct MelPlan {
    int passes;
    int pass_steps[MEL_MAX_PASSES];      // bins walked per pass (multiple of 4)
    int pass_split[MEL_MAX_PASSES];      // 1, 2 or 4 adjacent blocks share one filter group
    int pass_gbase[MEL_MAX_PASSES];      // first filter group (4 filters) of the pass
    int pass_start[MEL_MAX_PASSES][16];  // first bin of each block
};

// banks[filter][bin] (bins < row_len) -> plan + weights in MFMA B-operand order [step / 4][lane][step % 4], passes back to
// back.  Returns false when the filters need more than MEL_MAX_PASSES passes.
inline bool build_mel_plan(const std::vector<std::vector<float>>& banks, int row_len, MelPlan* plan, std::vector<float>* melb) {
    const int nfilt = (int)banks.size();
    const int nbin = nfilt ? (int)banks[0].size() : 0;
    const int groups = (nfilt + 3) / 4;
    plan->passes = 0;
    std::vector<int> clo(groups, 0), cnum(groups, 0);  // first 4-bin chunk and number of chunks of each group
    for (int g = 0; g < groups; ++g) {
        int lo = nbin, hi = -1;
        for (int m = 4 * g; m < 4 * g + 4 && m < nfilt; ++m)
            for (int k = 0; k < nbin; ++k)
                if (banks[m][k] > 0.0f) {
                    lo = k < lo ? k : lo;
                    hi = k > hi ? k : hi;
                }
        if (hi >= 0) {
            clo[g] = lo >> 2;
            cnum[g] = (hi >> 2) - (lo >> 2) + 1;
        }
    }
    int total_steps = 0;
    std::vector<int> seg_first[MEL_MAX_PASSES];  // first chunk of each block's segment
    for (int p = 0, g0 = 0; p < MEL_MAX_PASSES; ++p) {
        plan->pass_steps[p] = 0;
        plan->pass_split[p] = 1;
        plan->pass_gbase[p] = g0;
        for (int blk = 0; blk < 16; ++blk) plan->pass_start[p][blk] = 0;
        if (g0 >= groups) continue;
        const int left = groups - g0;
        // the last pass spreads its few groups over all 16 blocks; a full pass takes the next 16 groups
        const int split = left <= 4 ? 4 : (left <= 8 ? 2 : 1);
        const int ng = left < 16 / split ? left : 16 / split;
        int seg_chunks = 1;
        for (int g = g0; g < g0 + ng; ++g) {
            const int per = (cnum[g] + split - 1) / split;
            seg_chunks = per > seg_chunks ? per : seg_chunks;
        }
        const int steps = 4 * seg_chunks;
        plan->pass_steps[p] = steps;
        plan->pass_split[p] = split;
        seg_first[p].assign(16, -1);
        for (int blk = 0; blk < ng * split; ++blk) {
            const int g = g0 + blk / split, sidx = blk % split;
            const int per = (cnum[g] + split - 1) / split;
            const int first = clo[g] + sidx * per;  // chunks [first, first + per) of the group
            const int last = clo[g] + cnum[g];
            if (first >= last) continue;            // nothing left for this block: all-zero weights
            seg_first[p][blk] = first;
            int st = 4 * first;
            if (st + steps > row_len) st = row_len - steps;
            plan->pass_start[p][blk] = st;
        }
        total_steps += steps;
        g0 += ng;
        plan->passes = p + 1;
        if (p + 1 == MEL_MAX_PASSES && g0 < groups) return false;
    }
    melb->assign((size_t)total_steps * 64, 0.0f);  // [step / 4][lane][step % 4]
    for (int p = 0, off = 0; p < plan->passes; off += plan->pass_steps[p], ++p) {
        const int split = plan->pass_split[p];
        for (int ln = 0; ln < 64; ++ln) {
            const int blk = ln >> 2;
            if (seg_first[p][blk] < 0) continue;
            const int g = plan->pass_gbase[p] + blk / split;
            const int m = 4 * g + (ln & 3);
            if (m >= nfilt) continue;
            const int per = (cnum[g] + split - 1) / split;
            const int k_lo = 4 * seg_first[p][blk], k_hi = 4 * (seg_first[p][blk] + per);  // bins owned by this block
            for (int sidx = 0; sidx < plan->pass_steps[p]; ++sidx) {
                const int k = plan->pass_start[p][blk] + sidx;
                if (k >= k_lo && k < k_hi && k < nbin)
                    (*melb)[((size_t)((off + sidx) >> 2) * 64 + ln) * 4 + (sidx & 3)] = banks[m][k];
            }
        }
    }
    return true;
}

}  // namespace mv
