// MelSpectrogram front-end for gfx950: power STFT (centre / reflect padding, periodic Hann window) -> HTK mel filterbank
// (no log) -> per-utterance time-mean subtraction -> length mask.
//
// Replaces torchaudio.transforms.MelSpectrogram(**method_args) (mvector/data_utils/featurizer.py:41-42) and the CMN / mask
// passes of AudioFeaturizer.forward (featurizer.py:77-90); arithmetic as in oracle/frontend.py::mel_spectrogram.
//
// n_fft is arbitrary here (400 by default, 1024 in the README run, not a power of two in general), so the transform is an
// exact-fp32 DFT on the f32 MFMA (v_mfma_f32_16x16x4_f32) instead of a radix FFT:
//   stft_power_kernel   rows = frames, cols = frequency bins, K = n_fft; the frame matrix is never materialised -- A
//                       fragments are gathered straight from the waveform (reflect indexing at the edges) and multiplied by
//                       the window on the fly; two accumulators per tile (cos / sin tables) give re, im; the epilogue writes
//                       |X|^power.  One workgroup = 16 frames x up to 256 bins (each wave 64 bins, the A fragment is reused
//                       for 8 MFMA chains).
//   mel projection      exact-fp32 GEMM (linear.hip) against the transposed filterbank.
//   cmn_mask_kernel     column means over ALL frames (padded ones included, featurizer.py:79), subtract, zero the frames
//                       t >= round_half_even(ratio * T).
// hipcc-flags: -fno-slp-vectorize -fno-signed-zeros
//   (see fbank.hip: scalar chains stay scalar instead of being re-packed into v_pk_* ops behind register shuffles)
#include <vector>

#include "frontend_common.h"
#include "kernels.h"
#include "melfft.h"

namespace mv {

struct StftArgs {
    const float* wav;
    int64_t wav_stride;
    int64_t L;
    const float* window;  // [n_fft] (win_length window centred in n_fft, zero elsewhere)
    const float* dcos;    // [nbin_pad][kpad]
    const float* dsin;
    float* P;             // [B*T][nbin_pad]
    int B, T, n_fft, kpad, hop, pad, nbin, nbin_pad;
    float power;   // exponent of |X|
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

// pad / pad_mode (torchaudio.functional.spectrogram: F.pad(waveform, (pad, pad)) with zeros, then torch.stft(center=True, pad_mode=...) extends by
// n_fft / 2 on either side): the extended signal, written once; the transform kernels then see a centre = False problem
__global__ __launch_bounds__(256) void melspec_extend_kernel(const float* wav, int64_t wav_stride, int64_t L, float* dst, int64_t dst_stride, int64_t L2,
                                                             int pad, int centre, int mode) {
    const int b = blockIdx.y;
    const int64_t Lp = L + 2 * (int64_t)pad;
    const float* src = wav + (int64_t)b * wav_stride;
    float* row = dst + (int64_t)b * dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < (int64_t)(blockIdx.x + 1) * 1024 && i < dst_stride; i += 256) {
        float v = 0.0f;
        if (i < L2) {
            int64_t u = i - centre;   // position in the zero-padded signal of Lp samples
            bool zero = false;
            if (u < 0 || u >= Lp) {
                if (mode == MV_STFT_PAD_REFLECT) {
                    u = u < 0 ? -u : 2 * (Lp - 1) - u;
                } else if (mode == MV_STFT_PAD_REPLICATE) {
                    u = u < 0 ? 0 : Lp - 1;
                } else if (mode == MV_STFT_PAD_CIRCULAR) {
                    u = u < 0 ? u + Lp : u - Lp;
                } else {
                    zero = true;
                }
            }
            const int64_t k = u - pad;
            if (!zero && k >= 0 && k < L) v = src[k];
        }
        row[i] = v;
    }
}

__device__ __forceinline__ int reflect_index(int64_t i, int64_t L) {
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return (int)i;
}

// Real input: X[k] = sum_n xw[n] e^{-2 pi i k n / N} is folded about n = N/2,
//   Re X[k] = xw[0] + sum_{0 < n < N/2} (xw[n] + xw[N-n]) cos(2 pi k n / N) + xw[N/2] cos(pi k)       (N even)
//   Im X[k] =       - sum_{0 < n < N/2} (xw[n] - xw[N-n]) sin(2 pi k n / N)
// which halves the MFMA work (K = N/2 + 1 instead of N).  One WAVE owns 16 frames x NJT bin tiles (7: the 201 bins of the
// default n_fft = 400 are two groups), so a gathered A fragment feeds 2 * NJT MFMA chains and the four waves of a workgroup
// stream the same cos / sin rows through the L1 (the first version re-read 819 KB of tables from L2 for every 16 frames).
template <int NJT>
__global__ __launch_bounds__(256) void stft_power_kernel(StftArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int64_t nframes = (int64_t)a.B * a.T;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    if (row0 >= nframes) return;  // no barriers in this kernel
    const int64_t row = row0 + fr < nframes ? row0 + fr : nframes - 1;  // clamp; masked at the store
    const int b = (int)(row / a.T);
    const int t = (int)(row - (int64_t)b * a.T);
    const float* x = a.wav + (int64_t)b * a.wav_stride;
    const int64_t start = (int64_t)t * a.hop - a.pad;
    const bool interior = start >= 0 && start + a.n_fft <= a.L;
    const int N = a.n_fft, kfold = N / 2 + 1;
    auto xw = [&](int n) {  // windowed sample n of this lane's frame
        const int64_t idx = interior ? start + n : reflect_index(start + n, a.L);
        return x[idx] * a.window[n];
    };

    for (int bin0 = blockIdx.y * NJT * 16; bin0 < a.nbin; bin0 += gridDim.y * NJT * 16) {
        float4v re[NJT], im[NJT];
#pragma unroll
        for (int j = 0; j < NJT; ++j) re[j] = im[j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        // this lane's table row for tile j is (bin0 + j*16 + fr); rows past the padded table are clamped (masked at the store)
        int brow = bin0 + fr;
        for (int k0 = 0; k0 < a.kpad; k0 += 16) {
            const int k = k0 + 4 * g;
            float ae[4], ao[4];
            if (interior && N >= 64) {
                // four consecutive samples and their four mirror partners are two 16-byte loads each (samples, window)
                // instead of sixteen scattered 4-byte loads: the gather, not the MFMA, bounded the first version.
                // For k == 0 the mirror block is shifted by one (n = 0 has no partner; N - 0 lies outside the frame).
                const int sh = k == 0 ? 1 : 0;
                const float4v x1 = *reinterpret_cast<const float4u*>(x + start + k);
                const float4v w1 = *reinterpret_cast<const float4u*>(a.window + k);
                const float4v x2 = *reinterpret_cast<const float4u*>(x + start + N - k - 3 - sh);
                const float4v w2 = *reinterpret_cast<const float4u*>(a.window + N - k - 3 - sh);
                const float4v p1 = x1 * w1, p2 = x2 * w2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = k + e;
                    // partner N - n sits at element 3 - e (+1 when the block is shifted) of the mirror block
                    const float m = sh ? (e == 1 ? p2[3] : (e == 2 ? p2[2] : p2[1])) : p2[3 - e];
                    const bool has = n > 0 && 2 * n != N && n < kfold;
                    ae[e] = n < kfold ? p1[e] + (has ? m : 0.0f) : 0.0f;
                    ao[e] = has ? p1[e] - m : 0.0f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = k + e;
                    float ve = 0.0f, vo = 0.0f;
                    if (n < kfold) {
                        const float x1 = xw(n);
                        ve = x1;
                        if (n > 0 && 2 * n != N) {
                            const float x2 = xw(N - n);
                            ve = x1 + x2;
                            vo = x1 - x2;
                        }
                    }
                    ae[e] = ve;
                    ao[e] = vo;
                }
            }
#pragma unroll
            for (int j = 0; j < NJT; ++j) {
                int bin = brow + j * 16;
                bin = bin < a.nbin_pad ? bin : a.nbin_pad - 1;
                const float4v c4 = *reinterpret_cast<const float4v*>(a.dcos + (int64_t)bin * a.kpad + k);
                const float4v s4 = *reinterpret_cast<const float4v*>(a.dsin + (int64_t)bin * a.kpad + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], c4[e], re[j], 0, 0, 0);
                    im[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao[e], s4[e], im[j], 0, 0, 0);
                }
            }
        }
        // lane holds frames row0 + 4g + r, bin = bin0 + j*16 + fr
#pragma unroll
        for (int j = 0; j < NJT; ++j) {
            const int bin = bin0 + j * 16 + fr;
            if (bin >= a.nbin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t orow = row0 + 4 * g + r;
                if (orow < nframes) {
                    const float p = re[j][r] * re[j][r] + im[j][r] * im[j][r];
                    a.P[orow * a.nbin_pad + bin] = a.power == 2.0f ? p : (a.power == 1.0f ? sqrtf(p) : powf(p, 0.5f * a.power));   // uniform
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// melspec_tile_kernel: n_fft = 400 (the torchaudio default every MelSpectrogram configuration of the reference runs
// with) as a real FFT, fused with the HTK mel stage, the time-mean subtraction and the length mask in ONE launch.
//
// 400 = 16 x 25.  With n = 16 n1 + n2 and k = k1 + 25 k2:
//     X[k1 + 25 k2] = sum_n2 W16^(n2 k2) . W400^(n2 k1) . Y[k1, n2],      Y[k1, n2] = sum_n1 W25^(n1 k1) x[16 n1 + n2]
//   * a frame sits on 16 lanes, lane n2 holds the 25 REAL samples x[16 n1 + n2] (64 contiguous bytes per 16 lanes and n1);
//   * Y: real-input 25-point DFT in registers as 5 x 5 (n1 = 5a + b, k1 = c + 5d): five real 5-point DFTs over a, twiddles
//     W25^(bc) for c = 1, 2 only, then one real (c = 0) and two complex (c = 1, 2) 5-point DFTs over b; the inputs are real, so
//     Y[25 - k1] = conj Y[k1] and k1 = 0..12 is all that is needed: c = 1, 2 with all five d give k1 = {1,6,11,16,21},
//     {2,7,12,17,22}, i.e. Y[9], Y[4], Y[8], Y[3] by conjugation -- 188 lane-ops instead of 650 for the dense product;
//   * twiddle by W400^(n2 k1) (per-lane constants), ONE LDS transpose in the layout of fbank_tile_kernel (row k1 = [frame][n2]
//     + pad), then lane k1 < 13 runs the 16-point FFT over n2: its 16 outputs are the bins k1 + 25 k2, of which k2 >= 8 are
//     the conjugates of the bins 400 - k -- every bin 0..200 comes out exactly once (lane 0: k2 = 0..8), no post-processing;
//   * |X|^2 -> power rows in LDS -> banded HTK mel on v_mfma_f32_4x4x1 with the weights in registers (12 + 20 steps for 128
//     mels) -> raw power features (no log, featurizer.py:76-79);
//   * features of the first `tile_rows` frames wait in LDS for the time mean (featurizer.py:79) and are written once; the
//     frames that do not fit (27 of 241 for 3 s x 128 mels) take the write / re-read / rewrite route through global memory.
// Any other n_fft runs the dense-DFT kernels above.
constexpr int MST_WAVES = 8;
constexpr int MST_ROW = 65;                       // complex elements per transpose row: [4 frames][16 n2] + 1 pad
constexpr int MST_SLOT_FLOATS = 13 * MST_ROW * 2 + 2;  // 1692 floats per wave (16-byte multiple: the mel operands are 16-byte reads)
constexpr int MST_PSTR = 228;                     // floats between the power rows of the wave's four frames (36 banks apart)

struct MelTileArgs {
    const float* wav;
    int64_t wav_stride;
    int64_t L;
    const float* lens_ratio;
    float* out;               // [B, T, n_mels]
    const float* window;      // [400]
    const float* tw400;       // [13 k1][16 n2][2] cos, sin of 2 pi n2 k1 / 400
    const float* melb;        // MFMA B-operand order (frontend_common.h)
    int B, T, hop, pad, n_mels, cmn, tile_rows;
    MelPlan plan;
};

// real 5-point DFT: X[d] = sum_a v[a] W5^(a d); returns X[0] (real) and X[1], X[2] (X[5-d] = conj X[d])
__device__ __forceinline__ void rdft5(float v0, float v1, float v2, float v3, float v4, float& x0, cplx& x1, cplx& x2) {
    constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;    // sin(2 pi / 5), sin(4 pi / 5)
    const float s1 = v1 + v4, e1 = v1 - v4, s2 = v2 + v3, e2 = v2 - v3;
    x0 = v0 + (s1 + s2);
    x1 = cmake(fmaf(C2, s2, fmaf(C1, s1, v0)), -fmaf(S2, e2, S1 * e1));     // cos(4 pi d/5), sin(4 pi d/5) at d = 1
    x2 = cmake(fmaf(C1, s2, fmaf(C2, s1, v0)), -fmaf(-S1, e2, S2 * e1));    // d = 2: cos(8 pi/5) = C1, sin(8 pi/5) = -S1
}

// complex 5-point DFT, all five outputs
__device__ __forceinline__ void cdft5(cplx z0, cplx z1, cplx z2, cplx z3, cplx z4, cplx (&x)[5]) {
    constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;
    constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
    const cplx s1 = z1 + z4, e1 = z1 - z4, s2 = z2 + z3, e2 = z2 - z3;
    x[0] = z0 + (s1 + s2);
    const cplx t1 = z0 + cscale(s1, C1) + cscale(s2, C2);
    const cplx t2 = z0 + cscale(s1, C2) + cscale(s2, C1);
    const cplx u1 = cscale(e1, S1) + cscale(e2, S2);
    const cplx u2 = cscale(e1, S2) - cscale(e2, S1);
    // X[d] = t_d - i u_d, X[5-d] = t_d + i u_d;  -i u = {u.im, -u.re}
    const cplx m1 = mul_mi(u1), m2 = mul_mi(u2);
    x[1] = t1 + m1;
    x[4] = t1 - m1;
    x[2] = t2 + m2;
    x[3] = t2 - m2;
}

// z * W25^m (m compile-time)
template <int M>
__device__ __forceinline__ cplx mul_w25(cplx z) {
    constexpr double PI2_25 = 6.283185307179586476925 / 25.0;
    // constexpr cos/sin are not available in device code: table of the 8 angles the 5 x 5 split uses (m = b c, b = 1..4, c = 1, 2)
    constexpr float C[9] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899666f,
                            0.0f, 0.06279051952931337f, 0.0f, -0.42577929156507272f};
    constexpr float S[9] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f,
                            0.0f, 0.99802672842827156f, 0.0f, 0.90482705246601958f};
    static_assert(M == 1 || M == 2 || M == 3 || M == 4 || M == 6 || M == 8, "angle not in the table");
    (void)PI2_25;
    return cmul_conjtw(z, C[M], S[M]);
}

template <int G0, int G1>
__global__ __launch_bounds__(MST_WAVES * 64) void melspec_tile_kernel(MelTileArgs a) {
    constexpr int THREADS = MST_WAVES * 64;
    MV_DYN_SMEM(smem);
    float* xbuf = reinterpret_cast<float*>(smem);                  // [MST_WAVES][MST_SLOT_FLOATS]
    float* tile = xbuf + MST_WAVES * MST_SLOT_FLOATS;              // [tile_rows][n_mels]
    float* colsum = xbuf;                                          // [MST_WAVES][256] then mean[256], after the frame loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, fs = lane >> 4;
    const int b = blockIdx.x;
    const int T = a.T, nm = a.n_mels;
    const float* x = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * T * nm;

    // ---- per-lane constants ----
    float cwin[25];
#pragma unroll
    for (int n1 = 0; n1 < 25; ++n1) cwin[n1] = a.window[16 * n1 + l16];
    float2v ctw[13];
#pragma unroll
    for (int k1 = 1; k1 < 13; ++k1) ctw[k1] = *reinterpret_cast<const float2v*>(a.tw400 + 2 * (k1 * 16 + l16));
    float4v mb0[G0], mb1[G1 > 0 ? G1 : 1];
#pragma unroll
    for (int g = 0; g < G0; ++g) mb0[g] = *reinterpret_cast<const float4v*>(a.melb + (size_t)g * 256 + lane * 4);
#pragma unroll
    for (int g = 0; g < G1; ++g) mb1[g] = *reinterpret_cast<const float4v*>(a.melb + (size_t)(G0 + g) * 256 + lane * 4);

    for (int i = tid; i < MST_WAVES * MST_SLOT_FLOATS; i += THREADS) xbuf[i] = 0.0f;  // pads of the power rows meet zero weights: keep them finite
    __syncthreads();
    float* wslot = xbuf + wave * MST_SLOT_FLOATS;
    cplx* tw_write = reinterpret_cast<cplx*>(wslot) + lane;                              // (k1, frame fs, n2 = l16) at + k1 * MST_ROW
    const int krow = l16 < 13 ? l16 : 12;                                                 // lanes 13..15 idle through stage 2
    const cplx* tw_read = reinterpret_cast<const cplx*>(wslot) + krow * MST_ROW + 16 * fs; // (k1 = krow, fs, n2) at + n2
    float* prow = wslot + fs * MST_PSTR;
    float* p_lo = prow + krow;            // bin k1 + 25 k2, k2 = 0..7
    float* p_hi = prow + 25 - krow;       // bin 400 - k = (25 - k1) + 25 (15 - k2), k2 = 8..15 (lane 0: k2 = 8 only -> bin 200)
    const bool act = l16 < 13;
    const float* arow = wslot + (lane & 3) * MST_PSTR;
    const float* ap0 = arow + a.plan.pass_start[0][lane >> 2];
    const float* ap1 = arow + a.plan.pass_start[1][lane >> 2];
    const int blk = lane >> 2;
    const int split0 = a.plan.pass_split[0], split1 = a.plan.pass_split[1];
    const int m0 = 4 * (a.plan.pass_gbase[0] + blk / split0) + (lane & 3);
    const int m1 = 4 * (a.plan.pass_gbase[1] + blk / split1) + (lane & 3);
    const bool own0 = m0 < nm && (blk & (split0 - 1)) == 0;
    const bool own1 = G1 > 0 && m1 < nm && (blk & (split1 - 1)) == 0;
    float csum0 = 0.0f, csum1 = 0.0f;
    const int tile_rows = a.tile_rows;  // multiple of 4

    const int nquads = (T + 3) >> 2;
    // samples of one quad (lane n2 of a frame: x[16 n1 + n2]); first / last frames: reflect padding of torch.stft(center=True),
    // or zeros beyond the signal without it
    auto load_quad = [&](int q, float (&raw)[25]) {
        const int f_raw = q * 4 + fs;
        const int f = f_raw < T ? f_raw : T - 1;
        const int64_t start = (int64_t)f * a.hop - a.pad;
        // one decision for the wave's four frames (consecutive frames: first and last decide), on a scalar: a per-lane
        // branch would emit both paths behind exec masks and make the edge path wait for every interior load it overwrites
        const int qs = MV_UNIFORM(q);
        const int f_last = qs * 4 + 3 < T ? qs * 4 + 3 : T - 1;
        const bool interior = (int64_t)qs * 4 * a.hop - a.pad >= 0 && (int64_t)f_last * a.hop - a.pad + 400 <= a.L;
        if (interior) {
            const float* fp = x + start + l16;
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) raw[n1] = fp[16 * n1];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) {
                int64_t i = start + 16 * n1 + l16;
                if (a.pad > 0) {
                    if (i < 0) i = -i;
                    if (i >= a.L) i = 2 * (a.L - 1) - i;
                }
                raw[n1] = (i >= 0 && i < a.L) ? x[i] : 0.0f;
            }
        }
    };
    // the next quad's samples are requested as soon as this quad's are windowed (two register sets, loop unrolled by two):
    // their latency runs under the transform instead of stalling the top of every iteration
    auto process_quad = [&](int q, float (&raw)[25], float (&raw_next)[25]) __attribute__((always_inline)) {
        float v[25];
#pragma unroll
        for (int n1 = 0; n1 < 25; ++n1) v[n1] = raw[n1] * cwin[n1];
        if (q + MST_WAVES < nquads) load_quad(q + MST_WAVES, raw_next);
        // ---- Y[k1], k1 = 0..12: 5 x 5 real DFT (n1 = 5a + b) ----
        float a0[5];
        cplx a1[5], a2[5];
#pragma unroll
        for (int bb = 0; bb < 5; ++bb) rdft5(v[bb], v[5 + bb], v[10 + bb], v[15 + bb], v[20 + bb], a0[bb], a1[bb], a2[bb]);
        a1[1] = mul_w25<1>(a1[1]); a1[2] = mul_w25<2>(a1[2]); a1[3] = mul_w25<3>(a1[3]); a1[4] = mul_w25<4>(a1[4]);
        a2[1] = mul_w25<2>(a2[1]); a2[2] = mul_w25<4>(a2[2]); a2[3] = mul_w25<6>(a2[3]); a2[4] = mul_w25<8>(a2[4]);
        cplx y[13];
        {
            float y0;
            cplx y5, y10;
            rdft5(a0[0], a0[1], a0[2], a0[3], a0[4], y0, y5, y10);   // c = 0: k1 = 0, 5, 10
            y[0] = cmake(y0, 0.0f);
            y[5] = y5;
            y[10] = y10;
            cplx o[5];
            cdft5(a1[0], a1[1], a1[2], a1[3], a1[4], o);             // c = 1: k1 = 1, 6, 11, 16, 21
            y[1] = o[0]; y[6] = o[1]; y[11] = o[2];
            y[9] = cconj(o[3]);                          // Y[9] = conj Y[16]
            y[4] = cconj(o[4]);                          // Y[4] = conj Y[21]
            cdft5(a2[0], a2[1], a2[2], a2[3], a2[4], o);             // c = 2: k1 = 2, 7, 12, 17, 22
            y[2] = o[0]; y[7] = o[1]; y[12] = o[2];
            y[8] = cconj(o[3]);                          // Y[8] = conj Y[17]
            y[3] = cconj(o[4]);                          // Y[3] = conj Y[22]
        }
#pragma unroll
        for (int k1 = 1; k1 < 13; ++k1) y[k1] = cmul_conjtw(y[k1], ctw[k1][0], ctw[k1][1]);
        // ---- transpose: lane n2 -> lane k1 ----
#pragma unroll
        for (int k1 = 0; k1 < 13; ++k1) tw_write[k1 * MST_ROW] = y[k1];
        MV_WAVE_FENCE();
        cplx z[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_read_single(tw_read + n2);
        MV_WAVE_FENCE();
        fft16(z);  // z[k2] = X[k1 + 25 k2]
        float pw[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pw[k2] = z[k2][0] * z[k2][0] + z[k2][1] * z[k2][1];
        if (act) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) p_lo[25 * k2] = pw[k2];
            p_hi[25 * 7] = pw[8];
            if (l16 != 0) {  // lane 0: k2 = 9..15 repeat its bins 175..25
#pragma unroll
                for (int k2 = 9; k2 < 16; ++k2) p_hi[25 * (15 - k2)] = pw[k2];
            }
        }
        MV_WAVE_FENCE();
        // ---- banded mel on the matrix pipe ----
        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        float4v acc0[4], acc1[4];
#pragma unroll
        for (int g = 0; g < G0; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc0[c] = fb_mfma4(av[c], mb0[g][c], g == 0 ? zero4 : acc0[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc1[c] = zero4;
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc1[c] = fb_mfma4(av[c], mb1[g][c], g == 0 ? zero4 : acc1[c]);
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
    float ra[25], rb[25];
    if (wave < nquads) load_quad(wave, ra);
    for (int q = wave; q < nquads; q += 2 * MST_WAVES) {
        process_quad(q, ra, rb);
        if (q + MST_WAVES < nquads) process_quad(q + MST_WAVES, rb, ra);
    }

    // ---- per-utterance time mean over ALL frames (featurizer.py:79), mask, single write of the rows held in LDS ----
    __syncthreads();
    if (own0) colsum[wave * 256 + m0] = csum0;
    if (own1) colsum[wave * 256 + m1] = csum1;
    __syncthreads();
    float* mean = colsum + MST_WAVES * 256;
    if (tid < 256) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < MST_WAVES; ++w) s += colsum[w * 256 + tid];
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

// out[b, t, c] -= mean_t out[b, :, c]; frames t >= mask_len zeroed.  One workgroup per utterance.
__global__ __launch_bounds__(256) void cmn_mask_kernel(float* out, int T, int C, const float* lens_ratio, int cmn) {
    __shared__ float part[4][256];
    __shared__ float mean[256];
    const int b = blockIdx.x;
    float* o = out + (int64_t)b * T * C;
    const int tid = threadIdx.x;
    int mask_len = T;
    if (lens_ratio != nullptr) mask_len = (int)rintf(lens_ratio[b] * (float)T);
    for (int c0 = 0; c0 < C; c0 += 64) {
        // 64 columns x 4 time phases
        const int c = c0 + (tid & 63);
        const int ph = tid >> 6;
        float s = 0.0f;
        if (c < C)
            for (int t = ph; t < T; t += 4) s += o[(int64_t)t * C + c];
        part[ph][tid & 63] = s;
        __syncthreads();
        if (tid < 64) mean[tid] = cmn ? (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) / (float)T : 0.0f;
        __syncthreads();
        if (c < C) {
            const float m = mean[tid & 63];
            for (int t = ph; t < T; t += 4) {
                const float v = o[(int64_t)t * C + c] - m;
                o[(int64_t)t * C + c] = t < mask_len ? v : 0.0f;
            }
        }
        __syncthreads();
    }
}

}  // namespace mv

struct MvMelSpec {
    MvMelSpecCfg cfg;
    int nbin, nbin_pad, kpad, pad;
    bool pre_pad = false;   // cfg.pad > 0 or a centre padding other than reflect: melspec_extend_kernel writes the extended signal to the workspace first
    float* d_window = nullptr;
    float* d_cos = nullptr;
    float* d_sin = nullptr;
    float* d_fbT = nullptr;  // [n_mels][nbin_pad]
    // melspec_tile_kernel (n_fft = 400 and a mel plan that matches an instantiation)
    bool tile_kernel = false;
    float* d_tw400 = nullptr;
    float* d_melb = nullptr;
    mv::MelPlan plan;
    // melspec_pow2_kernel (power-of-two n_fft <= 1024, melfft.hip)
    bool pow2_kernel = false;
    float* d_tw512 = nullptr;
    float* d_w1024 = nullptr;
};

// instantiated mel geometry: 128 HTK filters over 0 .. 8 kHz on the 201 bins of n_fft = 400 = two passes of 16 filter groups
// walking 12 and 20 bins
constexpr int MST_G0 = 3, MST_G1 = 5;

namespace {

template <typename T>
int upload_vec(const std::vector<T>& v, T** dptr) {
    MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(dptr), v.size() * sizeof(T)));
    MV_HIP_OK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MV_OK;
}

}  // namespace

extern "C" {

void mv_melspec_default_cfg(MvMelSpecCfg* cfg) {
    cfg->sample_rate = 16000;
    cfg->n_fft = 400;
    cfg->win_length = 400;
    cfg->hop_length = 200;
    cfg->f_min = 0.0f;
    cfg->f_max = 8000.0f;
    cfg->n_mels = 128;
    cfg->power = 2.0f;
    cfg->center = 1;
    cfg->subtract_time_mean = 1;
    cfg->mel_scale = MV_MEL_HTK;
    cfg->norm = MV_MEL_NORM_NONE;
    cfg->normalized = MV_STFT_NORM_NONE;
    cfg->window = nullptr;
    cfg->pad = 0;
    cfg->pad_mode = MV_STFT_PAD_REFLECT;
}

int mv_melspec_create(const MvMelSpecCfg* cfg, MvMelSpec** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_melspec_create: null argument");
    MV_REQUIRE(cfg->n_fft >= 4 && cfg->n_fft <= 8192, "mv_melspec_create: n_fft out of range");
    MV_REQUIRE(cfg->win_length >= 1 && cfg->win_length <= cfg->n_fft, "mv_melspec_create: win_length must be in [1, n_fft]");
    MV_REQUIRE(cfg->hop_length >= 1, "mv_melspec_create: hop_length must be positive");
    MV_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= 256, "mv_melspec_create: n_mels must be in [1, 256]");
    MV_REQUIRE(cfg->f_min <= cfg->f_max, "mv_melspec_create: Require f_min <= f_max (torchaudio's MelScale raises the same)");
    MV_REQUIRE(cfg->power > 0.0f && cfg->power < 64.0f, "mv_melspec_create: power must be a positive exponent (power=None, the complex spectrogram, has no mel scale)");
    MV_REQUIRE(cfg->mel_scale == MV_MEL_HTK || cfg->mel_scale == MV_MEL_SLANEY, "mv_melspec_create: unknown mel_scale");
    MV_REQUIRE(cfg->norm == MV_MEL_NORM_NONE || cfg->norm == MV_MEL_NORM_SLANEY, "mv_melspec_create: unknown norm");
    MV_REQUIRE(cfg->normalized >= MV_STFT_NORM_NONE && cfg->normalized <= MV_STFT_NORM_FRAME_LENGTH, "mv_melspec_create: unknown normalized mode");
    MV_REQUIRE(cfg->pad >= 0 && cfg->pad < (1 << 24), "mv_melspec_create: pad must be a non-negative sample count");
    MV_REQUIRE(cfg->pad_mode >= MV_STFT_PAD_REFLECT && cfg->pad_mode <= MV_STFT_PAD_CIRCULAR, "mv_melspec_create: unknown pad_mode");
    MvMelSpec* h = new MvMelSpec();
    h->cfg = *cfg;
    h->cfg.window = nullptr;   // (the caller's host array is read below and not kept)
    const int n_fft = cfg->n_fft;
    h->nbin = n_fft / 2 + 1;
    h->nbin_pad = (int)mv::round_up(h->nbin, 16);
    h->kpad = (int)mv::round_up(n_fft / 2 + 1, 16);  // folded transform length, padded to the MFMA K block
    h->pad = cfg->center ? n_fft / 2 : 0;
    h->pre_pad = cfg->pad > 0 || (cfg->center && cfg->pad_mode != MV_STFT_PAD_REFLECT);
    const double pi = 3.14159265358979323846;
    // the window (periodic Hann of win_length unless the caller passed window_fn's values), centred in n_fft (torch.stft pads it on both sides)
    std::vector<float> window(n_fft, 0.0f);
    const int left = (n_fft - cfg->win_length) / 2;
    for (int i = 0; i < cfg->win_length; ++i)
        window[left + i] = cfg->window != nullptr ? cfg->window[i] : (float)(0.5 - 0.5 * cos(2.0 * pi * i / cfg->win_length));
    // normalized = "window": spec / sqrt(sum window^2); "frame_length": torch.stft(normalized=True) = spec / sqrt(n_fft).  Both scale the complex
    // spectrum, i.e. the window (torchaudio.functional.spectrogram)
    if (cfg->normalized != MV_STFT_NORM_NONE) {
        float ss = 0.0f;   // fp32 like window.pow(2.).sum().sqrt()
        for (int i = 0; i < cfg->win_length; ++i) ss += window[left + i] * window[left + i];
        const float div = cfg->normalized == MV_STFT_NORM_WINDOW ? sqrtf(ss) : sqrtf((float)n_fft);
        MV_REQUIRE(div > 0.0f, "mv_melspec_create: normalized with an all-zero window");
        for (int i = 0; i < cfg->win_length; ++i) window[left + i] /= div;
    }
    std::vector<float> dcos((size_t)h->nbin_pad * h->kpad, 0.0f), dsin((size_t)h->nbin_pad * h->kpad, 0.0f);
    for (int k = 0; k < h->nbin; ++k)
        for (int n = 0; n <= n_fft / 2; ++n) {
            const int64_t m = ((int64_t)k * n) % n_fft;  // exact argument reduction
            const double ang = 2.0 * pi * (double)m / n_fft;
            dcos[(size_t)k * h->kpad + n] = (float)cos(ang);
            dsin[(size_t)k * h->kpad + n] = (float)(-sin(ang));
        }
    // mel filterbank, triangles in Hz (torchaudio.functional.melscale_fbanks): HTK or Slaney mel points, optional Slaney area normalisation
    std::vector<float> fbT((size_t)cfg->n_mels * h->nbin_pad, 0.0f);
    {
        const int n_mels = cfg->n_mels;
        const bool slaney = cfg->mel_scale == MV_MEL_SLANEY;
        const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
        auto hz_to_mel = [&](double f) {
            if (!slaney) return 2595.0 * log10(1.0 + f / 700.0);
            return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
        };
        const double m_min = hz_to_mel(cfg->f_min), m_max = hz_to_mel(cfg->f_max);
        std::vector<float> f_pts(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) {
            // torch.linspace in fp32, then the mel -> Hz map in fp32
            const float m = (float)(m_min + (m_max - m_min) * i / (n_mels + 1));
            if (!slaney) f_pts[i] = 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
            else f_pts[i] = m >= (float)min_log_mel ? (float)min_log_hz * expf((float)logstep * (m - (float)min_log_mel)) : (float)f_sp * m;
        }
        for (int k = 0; k < h->nbin; ++k) {
            const float f = (float)((double)(cfg->sample_rate / 2) * k / (h->nbin - 1));
            for (int j = 0; j < n_mels; ++j) {
                const float down = (f - f_pts[j]) / (f_pts[j + 1] - f_pts[j]);
                const float up = (f_pts[j + 2] - f) / (f_pts[j + 2] - f_pts[j + 1]);
                const float w = fminf(down, up);
                const float enorm = cfg->norm == MV_MEL_NORM_SLANEY ? 2.0f / (f_pts[j + 2] - f_pts[j]) : 1.0f;
                fbT[(size_t)j * h->nbin_pad + k] = w > 0.0f ? w * enorm : 0.0f;
            }
        }
    }
    int rc;
    if ((rc = upload_vec(window, &h->d_window)) || (rc = upload_vec(dcos, &h->d_cos)) || (rc = upload_vec(dsin, &h->d_sin)) ||
        (rc = upload_vec(fbT, &h->d_fbT))) {
        mv_melspec_destroy(h);
        return rc;
    }
    // ---- FFT path (melspec_tile_kernel) when the geometry matches ----
    if (n_fft == 400 && cfg->power == 2.0f && (cfg->n_mels & 3) == 0 && cfg->n_mels <= 128) {
        std::vector<std::vector<float>> banks(cfg->n_mels, std::vector<float>(h->nbin, 0.0f));
        for (int j = 0; j < cfg->n_mels; ++j)
            for (int k = 0; k < h->nbin; ++k) banks[j][k] = fbT[(size_t)j * h->nbin_pad + k];
        std::vector<float> melb;
        const bool ok = mv::build_mel_plan(banks, 208, &h->plan, &melb);  // rows of 201 bins, readable up to 208 (MST_PSTR = 228)
        if (ok && h->plan.pass_steps[0] == 4 * MST_G0 && h->plan.pass_steps[1] == 4 * MST_G1) {
            std::vector<float> tw(13 * 16 * 2);
            for (int k1 = 0; k1 < 13; ++k1)
                for (int n2 = 0; n2 < 16; ++n2) {
                    tw[2 * (k1 * 16 + n2)] = (float)cos(2.0 * pi * (n2 * k1) / 400.0);
                    tw[2 * (k1 * 16 + n2) + 1] = (float)sin(2.0 * pi * (n2 * k1) / 400.0);
                }
            if ((rc = upload_vec(tw, &h->d_tw400)) || (rc = upload_vec(melb, &h->d_melb))) {
                mv_melspec_destroy(h);
                return rc;
            }
            h->tile_kernel = true;
            if (h->tile_kernel && MV_SET_MAX_SMEM((mv::melspec_tile_kernel<MST_G0, MST_G1>), 160 * 1024) != hipSuccess) {
                mv_melspec_destroy(h);
                return mv::fail(MV_ERR_HIP, "mv_melspec_create: cannot reserve dynamic LDS for melspec_tile_kernel");
            }
        }
    }
    // ---- FFT path for power-of-two transforms (melspec_pow2_kernel: the README's n_fft 1024 / hop 320 / 64 mels, and 128 ... 512) ----
    if (!h->tile_kernel && (n_fft == 128 || n_fft == 256 || n_fft == 512 || n_fft == 1024) && cfg->power == 2.0f && (cfg->n_mels & 3) == 0 &&
        cfg->n_mels <= 128) {
        std::vector<std::vector<float>> banks(cfg->n_mels, std::vector<float>(h->nbin, 0.0f));
        for (int j = 0; j < cfg->n_mels; ++j)
            for (int k = 0; k < h->nbin; ++k) banks[j][k] = fbT[(size_t)j * h->nbin_pad + k];
        std::vector<float> melb;
        if (mv::build_mel_plan(banks, mv::MF_PSTR, &h->plan, &melb)) {
            std::vector<float> tw(32 * 16 * 2), w1(512 * 2);
            for (int k1 = 0; k1 < 32; ++k1)
                for (int l = 0; l < 16; ++l) {
                    tw[2 * (k1 * 16 + l)] = (float)cos(2.0 * pi * (l * k1) / 512.0);
                    tw[2 * (k1 * 16 + l) + 1] = (float)sin(2.0 * pi * (l * k1) / 512.0);
                }
            for (int k = 0; k < 512; ++k) {
                w1[2 * k] = (float)cos(2.0 * pi * k / 1024.0);
                w1[2 * k + 1] = (float)sin(2.0 * pi * k / 1024.0);
            }
            if ((rc = upload_vec(tw, &h->d_tw512)) || (rc = upload_vec(w1, &h->d_w1024)) || (rc = upload_vec(melb, &h->d_melb))) {
                mv_melspec_destroy(h);
                return rc;
            }
            h->pow2_kernel = true;
        }
    }
    *out = h;
    return MV_OK;
}

int mv_melspec_info(const MvMelSpec* h, int32_t* tile_kernel) {
    MV_REQUIRE(h != nullptr && tile_kernel != nullptr, "mv_melspec_info: null argument");
    *tile_kernel = h->tile_kernel ? 1 : (h->pow2_kernel ? 2 : 0);  // 1 = melspec_tile_kernel (n_fft 400), 2 = melspec_pow2_kernel
    return MV_OK;
}

int mv_melspec_destroy(MvMelSpec* h) {
    if (h == nullptr) return MV_OK;
    hipFree(h->d_window);
    hipFree(h->d_cos);
    hipFree(h->d_sin);
    hipFree(h->d_fbT);
    hipFree(h->d_tw400);
    hipFree(h->d_melb);
    hipFree(h->d_tw512);
    hipFree(h->d_w1024);
    delete h;
    return MV_OK;
}

int mv_melspec_num_frames(const MvMelSpec* h, int64_t num_samples, int64_t* num_frames) {
    MV_REQUIRE(h != nullptr && num_frames != nullptr, "mv_melspec_num_frames: null argument");
    num_samples += 2 * (int64_t)h->cfg.pad;
    if (h->cfg.center) {
        *num_frames = 1 + num_samples / h->cfg.hop_length;
    } else {
        *num_frames = num_samples < h->cfg.n_fft ? 0 : 1 + (num_samples - h->cfg.n_fft) / h->cfg.hop_length;
    }
    return MV_OK;
}

// the extended signal of a pre_pad handle: [B][stride] floats in front of the other workspace sections (a multiple of 256 bytes)
static int64_t melspec_extended_len(const MvMelSpec* h, int64_t L) { return L + 2 * (int64_t)h->cfg.pad + 2 * (int64_t)h->pad; }
static int64_t melspec_extended_stride(const MvMelSpec* h, int64_t L) { return mv::round_up(melspec_extended_len(h, L), (int64_t)4); }
static size_t melspec_extended_bytes(const MvMelSpec* h, int32_t B, int64_t L) {
    return h->pre_pad ? (size_t)mv::round_up((int64_t)B * melspec_extended_stride(h, L) * (int64_t)sizeof(float), (int64_t)256) : 0;
}

size_t mv_melspec_workspace_bytes(const MvMelSpec* h, int32_t B, int64_t L) {
    if (h == nullptr || B <= 0) return 0;
    int64_t T = 0;
    mv_melspec_num_frames(h, L, &T);
    return melspec_extended_bytes(h, B, L) + (size_t)B * (size_t)T * h->nbin_pad * sizeof(float);
}

int mv_melspec_forward(const MvMelSpec* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                       const float* lens_ratio, float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    MV_REQUIRE(h != nullptr, "mv_melspec_forward: null handle");
    MV_REQUIRE(B >= 0 && L >= 0 && wav_stride >= L, "mv_melspec_forward: bad batch geometry");
    int64_t T = 0;
    mv_melspec_num_frames(h, L, &T);
    if (B == 0 || T == 0) return MV_OK;
    MV_REQUIRE(wav != nullptr && out != nullptr && workspace != nullptr, "mv_melspec_forward: null buffer");
    int centre_pad = h->pad;   // what the transform kernels still have to reflect themselves
    if (h->pre_pad) {
        const int64_t Lp = L + 2 * (int64_t)h->cfg.pad;
        if (h->cfg.center && h->cfg.pad_mode == MV_STFT_PAD_REFLECT) MV_REQUIRE(Lp > h->pad, "mv_melspec_forward: reflect padding needs more than n_fft/2 samples (torch.stft raises too)");
        if (h->cfg.center && h->cfg.pad_mode == MV_STFT_PAD_CIRCULAR) MV_REQUIRE(Lp >= h->pad, "mv_melspec_forward: circular padding needs at least n_fft/2 samples (torch.stft raises too)");
        const size_t ext = melspec_extended_bytes(h, B, L);
        MV_REQUIRE(workspace_bytes >= ext && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "mv_melspec_forward: workspace too small for the extended signal (mv_melspec_workspace_bytes)");
        MV_REQUIRE(B <= 65535, "mv_melspec_forward: pad / pad_mode take at most 65535 rows per call");
        const int64_t L2 = melspec_extended_len(h, L), stride2 = melspec_extended_stride(h, L);
        float* dst = static_cast<float*>(workspace);
        MV_LAUNCH(mv::melspec_extend_kernel, ((unsigned)mv::ceil_div(L2, (int64_t)1024), (unsigned)B, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), wav, wav_stride, L,
                  dst, stride2, L2, h->cfg.pad, h->pad, h->cfg.pad_mode);
        int rc = mv::check_launch("melspec_extend_kernel");
        if (rc != MV_OK) return rc;
        wav = dst;
        wav_stride = stride2;
        L = L2;
        centre_pad = 0;
        workspace = static_cast<char*>(workspace) + ext;
        workspace_bytes -= ext;
    } else if (h->cfg.center) {
        MV_REQUIRE(L > h->pad, "mv_melspec_forward: reflect padding needs more than n_fft/2 samples (torch.stft raises too)");
    }
    if (h->tile_kernel && (int64_t)T * h->cfg.n_mels < ((int64_t)1 << 31)) {
        mv::MelTileArgs t;
        t.wav = wav; t.wav_stride = wav_stride; t.L = L; t.lens_ratio = lens_ratio; t.out = out;
        t.window = h->d_window; t.tw400 = h->d_tw400; t.melb = h->d_melb;
        t.B = B; t.T = (int)T; t.hop = h->cfg.hop_length; t.pad = centre_pad; t.n_mels = h->cfg.n_mels; t.cmn = h->cfg.subtract_time_mean;
        t.plan = h->plan;
        // feature rows that fit next to the wave slots stay in LDS until the time mean is known; the rest go through global memory
        const size_t slots = (size_t)mv::MST_WAVES * mv::MST_SLOT_FLOATS * sizeof(float);
        int64_t rows = (int64_t)((160 * 1024 - slots) / ((size_t)h->cfg.n_mels * sizeof(float))) & ~(int64_t)3;
        const int64_t need = (T + 3) & ~(int64_t)3;
        t.tile_rows = (int)(rows < need ? rows : need);
        const size_t smem = slots + (size_t)t.tile_rows * h->cfg.n_mels * sizeof(float);
        const int prof = mv::prof_begin(MV_PROF_FBANK, (double)B * (4.0 * (double)L + 4.0 * (double)T * h->cfg.n_mels), static_cast<hipStream_t>(stream));
        MV_LAUNCH((mv::melspec_tile_kernel<MST_G0, MST_G1>), ((unsigned)B, 1, 1), (mv::MST_WAVES * 64, 1, 1), smem, static_cast<hipStream_t>(stream), t);
        mv::prof_end(prof, static_cast<hipStream_t>(stream));
        return mv::check_launch("melspec_tile_kernel");
    }
    if (h->pow2_kernel && (int64_t)T * h->cfg.n_mels < ((int64_t)1 << 31)) {
        mv::MelFftArgs t;
        t.wav = wav; t.wav_stride = wav_stride; t.L = L; t.lens_ratio = lens_ratio; t.out = out;
        t.window = h->d_window; t.tw512 = h->d_tw512; t.w1024 = h->d_w1024; t.melb = h->d_melb;
        t.B = B; t.T = (int)T; t.n_fft = h->cfg.n_fft; t.hop = h->cfg.hop_length; t.pad = centre_pad; t.n_mels = h->cfg.n_mels;
        t.cmn = h->cfg.subtract_time_mean;
        t.plan = h->plan;
        const size_t fixed = mv::melfft_fixed_lds_bytes();
        int64_t rows = (int64_t)((160 * 1024 - fixed) / ((size_t)h->cfg.n_mels * sizeof(float))) & ~(int64_t)3;
        const int64_t need = (T + 3) & ~(int64_t)3;
        t.tile_rows = (int)(rows < need ? rows : need);
        const size_t smem = fixed + (size_t)t.tile_rows * h->cfg.n_mels * sizeof(float);
        const int prof = mv::prof_begin(MV_PROF_FBANK, (double)B * (4.0 * (double)L + 4.0 * (double)T * h->cfg.n_mels), static_cast<hipStream_t>(stream));
        const int rc = mv::melfft_launch(t, smem, static_cast<hipStream_t>(stream));
        mv::prof_end(prof, static_cast<hipStream_t>(stream));
        return rc;
    }
    MV_REQUIRE(workspace_bytes >= (size_t)B * (size_t)T * h->nbin_pad * sizeof(float), "mv_melspec_forward: workspace too small");
    MV_REQUIRE((int64_t)B * T < ((int64_t)1 << 31), "mv_melspec_forward: too many frames");
    hipStream_t st = static_cast<hipStream_t>(stream);
    mv::StftArgs a;
    a.wav = wav;
    a.wav_stride = wav_stride;
    a.L = L;
    a.window = h->d_window;
    a.dcos = h->d_cos;
    a.dsin = h->d_sin;
    a.P = static_cast<float*>(workspace);
    a.B = B;
    a.T = (int)T;
    a.n_fft = h->cfg.n_fft;
    a.kpad = h->kpad;
    a.hop = h->cfg.hop_length;
    a.pad = centre_pad;
    a.nbin = h->nbin;
    a.nbin_pad = h->nbin_pad;
    a.power = h->cfg.power;
    const int64_t nframes = (int64_t)B * T;
    const unsigned gx = (unsigned)mv::ceil_div(nframes, 64);  // 4 waves x 16 frames
    // 7 bin tiles per wave (56 accumulator registers for re + im: four waves per SIMD); the default n_fft = 400 has 13 tiles
    const unsigned gy = (unsigned)mv::ceil_div(h->nbin, 7 * 16);
    MV_LAUNCH(mv::stft_power_kernel<7>, (gx, gy, 1), (256, 1, 1), 0, st, a);
    int rc = mv::check_launch("stft_power_kernel");
    if (rc != MV_OK) return rc;
    rc = mv::linear_f32_launch(a.P, h->nbin_pad, h->d_fbT, h->nbin_pad, nullptr, MV_ACT_NONE, out, h->cfg.n_mels, (int)nframes,
                               h->nbin, h->cfg.n_mels, 0, st);
    if (rc != MV_OK) return rc;
    if (h->cfg.subtract_time_mean || lens_ratio != nullptr) {
        MV_LAUNCH(mv::cmn_mask_kernel, ((unsigned)B, 1, 1), (256, 1, 1), 0, st, out, (int)T, h->cfg.n_mels, lens_ratio,
                  h->cfg.subtract_time_mean);
        rc = mv::check_launch("cmn_mask_kernel");
    }
    return rc;
}

}  // extern "C"
