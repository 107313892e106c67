// MelSpectrogram for power-of-two n_fft <= 1024 (the reference README's run: n_fft 1024, hop 320, 64 mels -- README_en.md:263-269;
// torchaudio.transforms.MelSpectrogram(**method_args), mvector/data_utils/featurizer.py:41-42) as ONE launch per batch, in the
// scheme of melspec_tile_kernel (melspec.hip, n_fft = 400): one workgroup per utterance, a frame on 16 lanes, real FFT in
// registers with one LDS transpose, banded HTK mel on v_mfma_f32_4x4x1, the feature tile waiting in LDS for the time mean and
// written once.  Before this kernel every n_fft other than 400 ran the dense-DFT kernels (O(n_fft^2): 1.05 MFLOP per frame at 1024).
//
// Transform: the frame (n_fft windowed samples, zero-extended to 1024: bin k of the n_fft-point transform is bin k * 1024 / n_fft of
// the 1024-point one) is packed as 512 complex values z[m] = x[2m] + i x[2m+1] and transformed as 512 = 32 x 16:
//     m = 16 n1 + l,  k = k1 + 32 k2:   Z[k1 + 32 k2] = sum_l W16^(l k2) . W512^(l k1) . Y[k1, l],   Y[k1, l] = sum_n1 W32^(n1 k1) z[16 n1 + l]
//   * lane l of the frame's 16 lanes holds z[16 n1 + l], n1 = 0..31 (a 128-byte run of samples per n1 and frame); Y: 32-point FFT in
//     registers (two fft16 + a radix-2 step), twiddle from an LDS table;
//   * transpose through LDS in two halves of 16 rows (the slot of fbank_tile_kernel): lane l then owns k1 = l AND k1 = 32 - l
//     (lane 0: k1 = 0 and 16) and runs two 16-point FFTs over the frame's lanes;
//   * real-input split X[k] = (Z[k] + conj Z[512-k]) / 2 - i W1024^k (Z[k] - conj Z[512-k]) / 2: the partner of bin k1 + 32 k2 is
//     (32 - k1) + 32 (15 - k2) -- in the SAME lane by the choice of the pair, so the split needs no lane exchange (lane 0 pairs its
//     bins among themselves);
//   * |X|^2 -> power rows in the transpose slot -> banded mel (weights streamed from L1 in MFMA operand order, run-time step counts:
//     the plan of frontend_common.h for whatever filter bank the arguments give) -> tile -> time mean, mask, single write.
//
// Occupancy (round 3, r09a): the first form (4 waves, next quad's samples prefetched into a second register set) held ~400 registers = one
// wave per SIMD and ran the README geometry in 104-108 us.  Of those registers ~80 were loop-invariant ADDRESSES: the tables sat behind the
// 66 KB of wave slots, beyond the 16-bit immediate of an LDS instruction, so each of the 80 table reads of a quad got its own address
// register; another ~50 were the split twiddles W1024^k and the power-row addresses, hoisted out of the frame loop.  Tables first in LDS,
// W1024^k from an LDS table, opaque per-quad store offsets, no divergent branch around the power stores: 232 registers without the
// prefetch = 8 waves per workgroup, two per SIMD, 70 us (n_fft 512: 97 -> 63 us, 256: 173 -> 108 us).  12 waves (168-register cap)
// spill 36 registers: 87 us.  MV_MELFFT_WAVES = 4 | 8 | 12 selects the form (profiles/r09a_melspec_pow2_waves_ab.log).
// hipcc-flags: -fno-slp-vectorize -fno-signed-zeros
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "frontend_common.h"
#include "kernels.h"
#include "melfft.h"

namespace mv {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a template argument (compile-time twiddles)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load from a 4-byte aligned address

// multiply by W32^M = exp(-2 pi i M / 32), M compile-time (0..15)
template <int M>
__device__ __forceinline__ cplx mul_w32(cplx a) {
    if constexpr ((M & 1) == 0) return mul_w16<M / 2>(a);
    constexpr float C[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f, 0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                             -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
    constexpr float S[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f, 0.83146961230254524f,
                             0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                             0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
    return cmul_conjtw(a, C[M & 15], S[M & 15]);
}

// one pair of the real-input split: za = Z[k], zb = Z[512 - k], w = W1024^k = wc - i ws  ->  |X[k]|^2, |X[512 - k]|^2
__device__ __forceinline__ void split_pair(cplx za, cplx zb, float wc, float ws, float& p_lo, float& p_hi) {
    const cplx e = cmake(za.re + zb.re, za.im - zb.im);   // Z[k] + conj Z[512 - k]
    const cplx d = cmake(za.re - zb.re, za.im + zb.im);   // Z[k] - conj Z[512 - k]
    const cplx t = cmul_conjtw(d, wc, ws);                // W1024^k d;  X[k] = (e - i t) / 2,  conj X[512 - k] = (e + i t) / 2
    const float ar = e.re + t.im, ai = e.im - t.re, br = e.re - t.im, bi = e.im + t.re;
    p_lo = 0.25f * (ar * ar + ai * ai);
    p_hi = 0.25f * (br * br + bi * bi);
}

// MF_WAVES = 8 waves per workgroup, the quad's samples loaded at the top of the quad (two waves per SIMD hide the latency for each other:
// README geometry 70 us; a second register set for a prefetched quad = one wave per SIMD 108 us, 12 waves with 36 spills 87 us: r09a)
__global__ __launch_bounds__(MF_WAVES * 64) void melspec_pow2_kernel(MelFftArgs a) {
    constexpr int THREADS = MF_WAVES * 64;
    MV_DYN_SMEM(smem);
    // (the tables come FIRST: their reads are base + compile-time offset, and an LDS offset is a 16-bit immediate -- behind the 66 KB of
    // wave slots every one of the 80 table reads of a quad had its own loop-invariant address register)
    float* lwin = reinterpret_cast<float*>(smem);                  // [1024] window (zero beyond n_fft)
    float* ltw = lwin + 1024;                                      // [32 k1][16 l][2]: cos, sin of 2 pi l k1 / 512
    float* lwk = ltw + 1024;                                       // [512][2]: cos, sin of 2 pi k / 1024 (the real-input split)
    float* xbuf = lwk + 1024;                                      // [MF_WAVES][MF_SLOT_FLOATS]: transpose rows, then power rows
    float* tile = xbuf + MF_WAVES * MF_SLOT_FLOATS;                // [tile_rows][n_mels]
    float* colsum = xbuf;                                          // [MF_WAVES][256] then mean[256], after the frame loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, fs = lane >> 4;
    const int b = blockIdx.x;
    const int T = a.T, nm = a.n_mels, n_fft = a.n_fft;
    const int sb = 1024 / n_fft;       // bin k of the n_fft-point transform = bin k * sb of the 1024-point one
    const int sbl = 31 - __builtin_clz(sb);  // sb = 1 << sbl (a run-time division per stored bin would cost more than the split itself)
    const int nrun = n_fft >> 5;       // 32-sample runs per frame (n1 < nrun carry samples)
    const float* x = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * T * nm;

    for (int i = tid; i < 1024; i += THREADS) {
        lwin[i] = i < n_fft ? a.window[i] : 0.0f;
        ltw[i] = a.tw512[i];
        lwk[i] = a.w1024[i];
    }
    for (int i = tid; i < MF_WAVES * MF_SLOT_FLOATS; i += THREADS) xbuf[i] = 0.0f;  // pads of the power rows meet zero weights: keep them finite
    __syncthreads();

    float* wslot = xbuf + wave * MF_SLOT_FLOATS;
    cplx* tw_write = reinterpret_cast<cplx*>(wslot) + lane;                     // row r at + r * MF_ROW: (frame fs, l) at position 16 fs + l
    const int rowa = l16, rowb = l16 == 0 ? 0 : 16 - l16;                        // rows of this lane's two k1 in the two halves
    const cplx* ra_read = reinterpret_cast<const cplx*>(wslot) + rowa * MF_ROW + 16 * fs;
    const cplx* rb_read = reinterpret_cast<const cplx*>(wslot) + rowb * MF_ROW + 16 * fs;
    float* prow = wslot + fs * MF_PSTR;
    // W1024^k of this lane's bins k = l16 + 32 k2 (lane 0's second set: 16 + 32 k2) come from the LDS table at + 64 k2 floats.  (They --
    // and the sixteen power-row addresses below -- are the same in every quad: computed from per-lane constants the compiler hoists
    // all of them out of the frame loop, ~80 registers that either spill or cost the second wave per SIMD; a table read and an
    // address the compiler cannot see through per quad keep them out of the loop-carried set.)
    const float* wk_own = lwk + 2 * l16;
    const bool keep = (l16 & (sb - 1)) == 0;   // this lane's bins exist in the n_fft-point transform
    const float* arow = wslot + (lane & 3) * MF_PSTR;
    const int blk = lane >> 2;
    const int split0 = a.plan.pass_split[0], split1 = a.plan.pass_split[1];
    const int G0 = a.plan.pass_steps[0] >> 2, G1 = a.plan.passes > 1 ? a.plan.pass_steps[1] >> 2 : 0;
    const float* ap0 = arow + a.plan.pass_start[0][blk];
    const float* ap1 = arow + a.plan.pass_start[1][blk];
    const float* mb0 = a.melb + lane * 4;
    const float* mb1 = a.melb + (size_t)G0 * 256 + lane * 4;
    const int m0 = 4 * (a.plan.pass_gbase[0] + blk / split0) + (lane & 3);
    const int m1 = 4 * (a.plan.pass_gbase[1] + blk / split1) + (lane & 3);
    const bool own0 = m0 < nm && (blk & (split0 - 1)) == 0;
    const bool own1 = G1 > 0 && m1 < nm && (blk & (split1 - 1)) == 0;
    float csum0 = 0.0f, csum1 = 0.0f;
    const int tile_rows = a.tile_rows;  // multiple of 4
    const int nquads = (T + 3) >> 2;

    // samples of one quad into (ev, od) = z[16 n1 + l] for even / odd n1, unwindowed
    auto load_quad = [&](int q, cplx (&ev)[16], cplx (&od)[16]) __attribute__((always_inline)) {
        // ---- samples: lane l of frame fs takes x[32 n1 + 2 l], x[32 n1 + 2 l + 1]; reflect padding of torch.stft(center=True) at the
        // utterance edges (one scalar decision for the wave's four frames), zeros beyond n_fft ----
        const int f_raw = q * 4 + fs;
        const int f = f_raw < T ? f_raw : T - 1;
        const int64_t start = (int64_t)f * a.hop - a.pad;
        const int qs = MV_UNIFORM(q);
        const int f_last = qs * 4 + 3 < T ? qs * 4 + 3 : T - 1;
        const bool interior = (int64_t)qs * 4 * a.hop - a.pad >= 0 && (int64_t)f_last * a.hop - a.pad + n_fft <= a.L;
        if (interior) {       // (one branch around all loads: inside the unrolled loop the compiler turns it into two loads + selects each)
#pragma unroll
            for (int h = 0; h < 16; ++h) {
                ev[h] = od[h] = cmake(0.0f, 0.0f);
                if (2 * h < nrun) {  // uniform
                    const float2v sv = *reinterpret_cast<const float2u*>(x + start + 64 * h + 2 * l16);
                    ev[h] = cmake(sv[0], sv[1]);
                }
                if (2 * h + 1 < nrun) {
                    const float2v sv = *reinterpret_cast<const float2u*>(x + start + 64 * h + 32 + 2 * l16);
                    od[h] = cmake(sv[0], sv[1]);
                }
            }
        } else {
            auto edge = [&](int64_t i) {
                if (a.pad > 0) {
                    if (i < 0) i = -i;
                    if (i >= a.L) i = 2 * (a.L - 1) - i;
                }
                return (i >= 0 && i < a.L) ? x[i] : 0.0f;
            };
#pragma unroll 1
            for (int h = 0; h < 16; ++h) {   // rolled: the utterance's first and last frames only
                cplx e = cmake(0.0f, 0.0f), o = cmake(0.0f, 0.0f);
                if (2 * h < nrun) e = cmake(edge(start + 64 * h + 2 * l16), edge(start + 64 * h + 2 * l16 + 1));
                if (2 * h + 1 < nrun) o = cmake(edge(start + 64 * h + 32 + 2 * l16), edge(start + 64 * h + 32 + 2 * l16 + 1));
                // rolled loop, register arrays: select the element by comparison (16 selects per value, edge frames only)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j == h) {
                        ev[j] = e;
                        od[j] = o;
                    }
                }
            }
        }
    };
    // the next quad's samples are requested as soon as this quad's are windowed (two register sets, loop unrolled by two): their
    // latency runs under the transform (one wave per SIMD: nobody else hides it)
    auto process_quad = [&](int q, cplx (&ev)[16], cplx (&od)[16]) __attribute__((always_inline)) {
        // window (zero beyond n_fft: the frame is zero-extended to 1024 samples)
#pragma unroll
        for (int h = 0; h < 16; ++h) {
            const float2v we = lds_load_unmerged(reinterpret_cast<const float2v*>(lwin + 64 * h + 2 * l16));
            const float2v wo = lds_load_unmerged(reinterpret_cast<const float2v*>(lwin + 64 * h + 32 + 2 * l16));
            ev[h] = cmul_elem(ev[h], we[0], we[1]);
            od[h] = cmul_elem(od[h], wo[0], wo[1]);
        }
        // ---- Y[k1], k1 = 0..31: fft32 = two fft16 + radix-2 ----
        fft16(ev);
        fft16(od);
        od[1] = mul_w32<1>(od[1]);   od[2] = mul_w32<2>(od[2]);   od[3] = mul_w32<3>(od[3]);   od[4] = mul_w32<4>(od[4]);
        od[5] = mul_w32<5>(od[5]);   od[6] = mul_w32<6>(od[6]);   od[7] = mul_w32<7>(od[7]);   od[8] = mul_w32<8>(od[8]);
        od[9] = mul_w32<9>(od[9]);   od[10] = mul_w32<10>(od[10]); od[11] = mul_w32<11>(od[11]); od[12] = mul_w32<12>(od[12]);
        od[13] = mul_w32<13>(od[13]); od[14] = mul_w32<14>(od[14]); od[15] = mul_w32<15>(od[15]);
        // first half k1 = 0..15 (E + W O), twiddled by W512^(l k1), through the slot; then the second half k1 = 16..31 (E - W O)
        cplx za[16], zb[16];
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const float2v tw = lds_load_unmerged(reinterpret_cast<const float2v*>(ltw + 2 * (k1 * 16 + l16)));
            tw_write[k1 * MF_ROW] = cmul_conjtw(ev[k1] + od[k1], tw[0], tw[1]);
            od[k1] = ev[k1] - od[k1];   // the second half's value: ev is dead from here on (32 registers less across the transpose)
        }
        MV_WAVE_FENCE();
#pragma unroll
        for (int l = 0; l < 16; ++l) za[l] = lds_read_single(ra_read + l);
        MV_WAVE_FENCE();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const float2v tw = lds_load_unmerged(reinterpret_cast<const float2v*>(ltw + 2 * ((16 + k1) * 16 + l16)));
            tw_write[k1 * MF_ROW] = cmul_conjtw(od[k1], tw[0], tw[1]);
        }
        MV_WAVE_FENCE();
#pragma unroll
        for (int l = 0; l < 16; ++l) zb[l] = lds_read_single(rb_read + l);
        MV_WAVE_FENCE();  // the slot now takes the power rows
        fft16(za);  // za[k2] = Z[l16 + 32 k2]
        fft16(zb);  // zb[k2] = Z[32 - l16 + 32 k2]   (lane 0: Z[16 + 32 k2])
        // ---- real-input split + power ----
        const bool lane0 = l16 == 0;
        // bin k = l16 + 32 k2 lives at k >> sbl (exact for the lanes that keep it); the other lanes (n_fft < 1024: their bins do not exist in
        // the n_fft-point transform) store into the row's pad, which meets zero mel weights -- no divergent branch around 32 stores
        int o_lo = keep ? l16 >> sbl : 516, o_hi = keep ? (512 >> sbl) - (l16 >> sbl) : 517;
        const int o_step = keep ? 32 >> sbl : 0;
        MV_OPAQUE(o_lo);
        MV_OPAQUE(o_hi);
        static_for<16>([&](auto ic) {
            constexpr int k2 = decltype(ic)::value;
            // k = l16 + 32 k2: partner zb[15 - k2]; lane 0 (k = 32 k2): partner za[16 - k2] (k2 = 0: itself -> X[0], X[512])
            const cplx pa = za[k2];
            const cplx other = zb[15 - k2], self = za[(16 - k2) & 15];
            const cplx pb = cmake(lane0 ? self.re : other.re, lane0 ? self.im : other.im);
            const float2v wk = lds_load_unmerged(reinterpret_cast<const float2v*>(wk_own + 64 * k2));  // (cos, sin) of 2 pi k / 1024
            float p_lo, p_hi;
            split_pair(pa, pb, wk[0], wk[1], p_lo, p_hi);
            prow[o_lo + k2 * o_step] = p_lo;
            prow[o_hi - k2 * o_step] = p_hi;
        });
        if (lane0 && (16 & (sb - 1)) == 0) {  // lane 0's second set: k = 16 + 32 k2 pairs with 16 + 32 (15 - k2) inside the set
            static_for<8>([&](auto ic) {
                constexpr int k2 = decltype(ic)::value;
                const float2v wk = lds_load_unmerged(reinterpret_cast<const float2v*>(lwk + 2 * (16 + 32 * k2)));
                float p_lo, p_hi;
                split_pair(zb[k2], zb[15 - k2], wk[0], wk[1], p_lo, p_hi);
                const int k = 16 + 32 * k2;
                prow[k >> sbl] = p_lo;
                prow[(512 - k) >> sbl] = p_hi;
            });
        }
        MV_WAVE_FENCE();
        // ---- banded mel on the matrix pipe (weights in MFMA operand order, streamed through L1) ----
        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        float4v acc0[4], acc1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc0[c] = acc1[c] = zero4;
        for (int g = 0; g < G0; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);
            const float4v mv4 = *reinterpret_cast<const float4v*>(mb0 + (size_t)g * 256);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc0[c] = fb_mfma4(av[c], mv4[c], acc0[c]);
        }
        for (int g = 0; g < G1; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);
            const float4v mv4 = *reinterpret_cast<const float4v*>(mb1 + (size_t)g * 256);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc1[c] = fb_mfma4(av[c], mv4[c], acc1[c]);
        }
        float4v r0 = (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
        float4v r1 = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
        MV_WAVE_FENCE();  // the power rows are consumed: the next quad's transpose may overwrite them
        if (split0 >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r0[r] += dpp_mov<DPP_ROW_SHL4>(0.0f, r0[r]);
        }
        if (split0 == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r0[r] += dpp_mov<DPP_ROW_SHL8>(0.0f, r0[r]);
        }
        if (split1 >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[r] += dpp_mov<DPP_ROW_SHL4>(0.0f, r1[r]);
        }
        if (split1 == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[r] += dpp_mov<DPP_ROW_SHL8>(0.0f, r1[r]);
        }
        const int frames_here = (q * 4 + 4 <= T) ? 4 : T - q * 4;
        if (frames_here < 4) {
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                if (r >= frames_here) {
                    r0[r] = 0.0f;
                    r1[r] = 0.0f;
                }
            }
        }
        csum0 += (r0[0] + r0[1]) + (r0[2] + r0[3]);
        csum1 += (r1[0] + r1[1]) + (r1[2] + r1[3]);
        const int row0 = q * 4 * nm;
        if (q * 4 < tile_rows) {  // uniform: tile_rows is a multiple of 4
            auto d0 = MV_AS_LDS(float, tile + row0 + m0);
            auto d1 = MV_AS_LDS(float, tile + row0 + m1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < frames_here) {
                    if (own0) d0[r * nm] = r0[r];
                    if (own1) d1[r * nm] = r1[r];
                }
            }
        } else {
            auto d0 = MV_AS_GLOBAL(float, orow + row0 + m0);
            auto d1 = MV_AS_GLOBAL(float, orow + row0 + m1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < frames_here) {
                    if (own0) d0[r * nm] = r0[r];
                    if (own1) d1[r * nm] = r1[r];
                }
            }
        }
    };
    {
        cplx ea[16], oa[16];
        for (int q = wave; q < nquads; q += MF_WAVES) {
            load_quad(q, ea, oa);
            process_quad(q, ea, oa);
        }
    }

    // ---- per-utterance time mean over ALL frames (featurizer.py:79), mask, single write of the rows held in LDS ----
    __syncthreads();
    if (own0) colsum[wave * 256 + m0] = csum0;
    if (own1) colsum[wave * 256 + m1] = csum1;
    __syncthreads();
    float* mean = colsum + MF_WAVES * 256;
    if (tid < 256) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < MF_WAVES; ++w) s += colsum[w * 256 + tid];
        mean[tid] = (a.cmn && tid < nm) ? s / (float)T : 0.0f;
    }
    __syncthreads();
    int mask_len = T;
    if (a.lens_ratio != nullptr) mask_len = (int)rintf(a.lens_ratio[b] * (float)T);
    const int qn = nm >> 2;  // n_mels % 4 == 0 for this kernel
    const int rows_per_pass = THREADS / qn;
    const int r0 = tid / qn, cg = tid - r0 * qn;
    if (r0 < rows_per_pass) {
        const float4v m4 = *reinterpret_cast<const float4v*>(mean + 4 * cg);
        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        int t = r0;
        for (; t < T && t < tile_rows; t += rows_per_pass) {  // rows held in LDS: written to HBM once
            const float4v raw = *(reinterpret_cast<const float4v*>(tile + t * nm) + cg);
            *(reinterpret_cast<float4v*>(orow + (int64_t)t * nm) + cg) = t < mask_len ? raw - m4 : zero4;
        }
        for (; t < T; t += rows_per_pass) {                   // rows that went through global memory
            float4v* gp = reinterpret_cast<float4v*>(orow + (int64_t)t * nm) + cg;
            const float4v raw = *gp;
            *gp = t < mask_len ? raw - m4 : zero4;
        }
    }
}

size_t melfft_fixed_lds_bytes() { return ((size_t)MF_WAVES * MF_SLOT_FLOATS + 3072) * sizeof(float); }

int melfft_launch(const MelFftArgs& a, size_t smem, hipStream_t stream) {
    static DeviceOnce attr_set;   // (per device: the attribute belongs to the current device's code object)
    int attr_set_slot;
    if (device_once_pending(attr_set, &attr_set_slot)) {
        if (MV_SET_MAX_SMEM(melspec_pow2_kernel, 160 * 1024) != hipSuccess)
            return fail(MV_ERR_HIP, "melspec_pow2_kernel: cannot reserve dynamic LDS");
        device_once_done(attr_set, attr_set_slot);
    }
    MV_LAUNCH(melspec_pow2_kernel, ((unsigned)a.B, 1, 1), (MF_WAVES * 64, 1, 1), smem, stream, a);
    return check_launch("melspec_pow2_kernel");
}

}  // namespace mv
