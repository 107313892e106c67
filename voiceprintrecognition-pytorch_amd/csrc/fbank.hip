// Batched Kaldi log-mel filterbank + time-mean subtraction + length mask for gfx950.
//
// Replaces the per-utterance Python loop of KaldiFbank.forward (mvector/data_utils/featurizer.py:
// 119-132, torchaudio.compliance.kaldi.fbank) and the CMN / mask passes of AudioFeaturizer.forward
// (featurizer.py:77-90).  Arithmetic follows oracle/frontend.py::kaldi_fbank.
//
// Mapping (wave64): one frame per 16 lanes, four frames per wave.  The 512-point real FFT of a
// frame is a 256-point complex FFT of z[n] = x[2n] + i x[2n+1], computed as 16 x 16:
//   stage 1  lane n2 holds z[16*n1 + n2], n1 = 0..15 (its samples are the float2 at 32*n1 + 2*n2,
//            so a 16-lane group reads 128 contiguous bytes) and runs a radix-16 butterfly in registers;
//   twiddle  W256^(n2*k1), per-lane constants kept in registers;
//   transpose through a padded per-wave LDS tile (conflict-free both ways);
//   stage 2  lane k1 runs the second radix-16 butterfly -> Z[k1 + 16*k2];
//   real post-processing with the partner bin Z[256 - k] (LDS exchange), power spectrum to LDS;
//   sparse triangular mel filters: lane n2 owns filters n2 + 16*i and walks only their non-zero
//   bins (501 weights in total instead of a dense 257 x 80 product), log, store.
// One workgroup owns one utterance, so the per-utterance time mean is a workgroup reduction and the
// second pass (subtract mean, apply the length mask) re-reads rows this CU has just written (L2 hits).
#include "common.h"

#include <cstdlib>
#include <vector>

namespace mv {

struct cplx {
    float re, im;
};

__device__ __forceinline__ cplx cmake(float r, float i) {
    cplx c;
    c.re = r;
    c.im = i;
    return c;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.re - b.re, a.im - b.im); }
// a * (c - i s)
__device__ __forceinline__ cplx cmul_conjtw(cplx a, float c, float s) {
    return cmake(a.re * c + a.im * s, a.im * c - a.re * s);
}

// multiply by W16^M = exp(-2 pi i M / 16), M compile-time
template <int M>
__device__ __forceinline__ cplx mul_w16(cplx a) {
    constexpr int m = M & 15;
    if constexpr (m == 0) return a;
    if constexpr (m == 4) return cmake(a.im, -a.re);
    if constexpr (m == 8) return cmake(-a.re, -a.im);
    if constexpr (m == 12) return cmake(-a.im, a.re);
    constexpr float C[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                             -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
                             0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
    constexpr float S[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                             -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
    return cmul_conjtw(a, C[m], S[m]);
}

__device__ __forceinline__ void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3) {
    cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cmake(t1.re + t3.im, t1.im - t3.re);  // t1 - i t3
    a3 = cmake(t1.re - t3.im, t1.im + t3.re);  // t1 + i t3
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

__device__ __forceinline__ void fb_glds16(const void* gsrc, char* lds_wave_base) {
#ifdef MV_EMU
    memcpy(lds_wave_base + (emu::flat_tid() & 63) * 16, gsrc, 16);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

__device__ __forceinline__ void fb_wait_loads() {
#ifndef MV_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

constexpr int FB_NFFT = 512;
constexpr int FB_MAX_WAVES = 16;          // waves per workgroup: 8, 12 or 16 (template parameter WAVES)
constexpr int FB_TSTRIDE = 17;            // padded row of the 16x16 transpose tile (complex elements)
constexpr int FB_SLOT_CPLX = 16 * FB_TSTRIDE;  // 272 complex = 2176 B per frame slot (>= 256 complex)
constexpr int FB_MAX_ROUNDS = 8;          // filters per lane: num_mel_bins <= 128

struct FbankTables {
    const float* window;    // [512] window, zero beyond the frame length
    const float* tw256;     // [16 k1][16 n2][2] cos, sin of 2 pi n2 k1 / 256 (lane-contiguous)
    const float* tw512;     // [256][2] cos, sin of 2 pi k / 512
    const float* melw;      // [sum_i width_i][16] filter weights, round-major
    const int* mel_start;   // [rounds*16] first FFT bin of each filter
    int melw_elems;
    int rounds;
    int round_width[FB_MAX_ROUNDS];
    int round_off[FB_MAX_ROUNDS];
};

struct FbankArgs {
    const float* wav;
    int64_t wav_stride;
    const float* lens_ratio;
    float* out;
    int B, T;
    int win, shift, nbins;
    float preemph;
    float inv_win;
    int remove_dc, use_power, use_log, cmn;
    int vec2_ok;
    int load_mode;  // 0 scalar, 1 float2, 2 LDS-DMA prefetch
    int64_t L;
    FbankTables tab;
};

// ROUNDS = ceil(num_mel_bins / 16): filters per lane (compile-time so the per-lane state stays in registers)
// MODE: 0 scalar global loads, 1 float2 global loads, 2 = the quad's samples are prefetched by LDS-DMA one iteration ahead
template <int ROUNDS, int MODE, int FB_WAVES>
__global__ __launch_bounds__(FB_WAVES * 64) void fbank_kernel(FbankArgs a) {
    constexpr int FB_THREADS = FB_WAVES * 64;
    MV_DYN_SMEM(smem);
    // carve (every offset but the last is a compile-time constant, so LDS accesses are base + immediate):
    // per-wave exchange tiles | tw512 | window | stage twiddles | column sums + mean | mel_start | mel weights
    cplx* xbuf = reinterpret_cast<cplx*>(smem);                                   // [FB_WAVES*4][FB_SLOT_CPLX]
    float* tw512 = reinterpret_cast<float*>(xbuf + FB_WAVES * 4 * FB_SLOT_CPLX);  // [512]
    float* lwin = tw512 + 512;                                                    // [512] window taps
    float* ltw = lwin + 512;                                                      // [16][16][2] stage twiddles
    float* colsum = ltw + 512;                                                    // [FB_WAVES][128] then mean[128]
    int* mstart = reinterpret_cast<int*>(colsum + (FB_WAVES + 1) * 128);          // [FB_MAX_ROUNDS*16]
    float* melw = reinterpret_cast<float*>(mstart + FB_MAX_ROUNDS * 16);          // [melw_elems]
    float* sring = melw + a.tab.melw_elems;                                       // MODE 2: [FB_WAVES][2][1024] samples

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l16 = lane & 15;   // n2 in stage 1, k1 in stage 2, filter lane in the mel stage
    const int fs = lane >> 4;    // frame slot inside the wave
    const int b = blockIdx.x;
    const int T = a.T;
    const int nbins = a.nbins;

    for (int i = tid; i < 512; i += FB_THREADS) {
        tw512[i] = a.tab.tw512[i];
        lwin[i] = a.tab.window[i];
        ltw[i] = a.tab.tw256[i];
    }
    for (int i = tid; i < a.tab.melw_elems; i += FB_THREADS) melw[i] = a.tab.melw[i];
    for (int i = tid; i < a.tab.rounds * 16; i += FB_THREADS) mstart[i] = a.tab.mel_start[i];
    for (int i = tid; i < FB_WAVES * 128; i += FB_THREADS) colsum[i] = 0.0f;

    __syncthreads();

    cplx* slot = xbuf + (wave * 4 + fs) * FB_SLOT_CPLX;
    float* pslot = reinterpret_cast<float*>(slot);
    const float* wrow = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * T * nbins;

    float csum[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) csum[i] = 0.0f;

    const int nquads = (T + 3) >> 2;
    // MODE 2: the 3*shift + win (<= 1024) samples of a quad of frames are contiguous; the wave copies them global -> LDS with
    // four 1 KiB DMA transfers, one iteration ahead of their use (each sample is fetched once instead of 2.5 times).
    float* swave = sring + wave * 2048;
    auto prefetch_quad = [&](int q, int buf) {
        const int64_t first = (int64_t)q * 4 * a.shift;
        const int64_t last_ok = a.L - 4 - first;  // largest in-row start of a 4-float transfer, relative to `first`
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t off = i * 256 + lane * 4;
            off = off < last_ok ? off : last_ok;  // clamped lanes land on samples no valid frame of this quad reads
            fb_glds16(wrow + first + off, reinterpret_cast<char*>(swave + buf * 1024 + i * 256));
        }
    };
    if (MODE == 2 && wave < nquads) prefetch_quad(wave, 0);
    int ring = 0;
    for (int q = wave; q < nquads; q += FB_WAVES) {
        const int f_raw = q * 4 + fs;
        const bool fvalid = f_raw < T;
        const int f = fvalid ? f_raw : T - 1;
        const float* fp;
        if (MODE == 2) {
            fb_wait_loads();   // this quad's transfers (issued one iteration ago) have landed
            MV_WAVE_FENCE();
            fp = swave + ring * 1024 + (f - q * 4) * a.shift;
            if (q + FB_WAVES < nquads) prefetch_quad(q + FB_WAVES, ring ^ 1);
            ring ^= 1;
        } else {
            fp = wrow + (int64_t)f * a.shift;
        }

        // ---- load the frame: lane holds samples 32*n1 + 2*l16 (+1).  Branch-free: indices beyond the frame are
        // clamped to a valid address and the value is zeroed by a select, so all 16 loads are in flight together ----
        float e0[16], e1[16];
        float s = 0.0f;
        const int n1_full = a.win >> 5;  // groups of 32 samples that lie entirely inside the frame (uniform)
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            float v0 = 0.0f, v1 = 0.0f;
            if (n1 < n1_full) {  // plain base + constant offset: nothing per-lane to keep alive across iterations
                if (MODE >= 1) {
                    const float2v v = *reinterpret_cast<const float2v*>(fp + 32 * n1 + 2 * l16);
                    v0 = v[0];
                    v1 = v[1];
                } else {
                    v0 = fp[32 * n1 + 2 * l16];
                    v1 = fp[32 * n1 + 2 * l16 + 1];
                }
            }
            e0[n1] = v0;
            e1[n1] = v1;
        }
        {   // the one group that straddles the end of the frame (none when win is a multiple of 32): clamp + select
            const int idx = 32 * n1_full + 2 * l16;
            float p0 = 0.0f, p1 = 0.0f;
            if (32 * n1_full < a.win) {
                p0 = fp[idx < a.win ? idx : a.win - 1];
                p1 = fp[idx + 1 < a.win ? idx + 1 : a.win - 1];
                p0 = idx < a.win ? p0 : 0.0f;
                p1 = idx + 1 < a.win ? p1 : 0.0f;
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                e0[n1] = n1 == n1_full ? p0 : e0[n1];
                e1[n1] = n1 == n1_full ? p1 : e1[n1];
                s += e0[n1] + e1[n1];
            }
        }
        // ---- remove DC (frame mean over the `win` samples) ----
        if (a.remove_dc) {
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            const float mu = s * a.inv_win;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int idx = 32 * n1 + 2 * l16;
                e0[n1] = (idx < a.win) ? e0[n1] - mu : 0.0f;
                e1[n1] = (idx + 1 < a.win) ? e1[n1] - mu : 0.0f;
            }
        }
        // ---- pre-emphasis y[j] = d[j] - c d[j-1], d[-1] := d[0]; then window ----
        cplx z[16];
        {
            const int src = (lane & 48) | ((l16 + 15) & 15);  // previous lane of the 16-lane group (rotating)
            float rprev = 0.0f;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float r = __shfl(e1[n1], src);
                float prev;
                if (l16 == 0)
                    prev = (n1 == 0) ? e0[0] : rprev;
                else
                    prev = r;
                rprev = r;
                const float y0 = e0[n1] - a.preemph * prev;
                const float y1 = e1[n1] - a.preemph * e0[n1];
                const float2v w2 = *reinterpret_cast<const float2v*>(lwin + 32 * n1 + 2 * l16);
                z[n1] = cmake(y0 * w2[0], y1 * w2[1]);
            }
        }
        // ---- stage 1: radix-16 over n1, twiddle by W256^(n2*k1) ----
        fft16(z);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            const float2v tw = *reinterpret_cast<const float2v*>(ltw + 2 * (k1 * 16 + l16));
            z[k1] = cmul_conjtw(z[k1], tw[0], tw[1]);
        }
        // ---- transpose: write [k1][n2], read [k1 = lane][n2] ----
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) slot[k1 * FB_TSTRIDE + l16] = z[k1];
        MV_WAVE_FENCE();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = slot[l16 * FB_TSTRIDE + n2];
        MV_WAVE_FENCE();
        // ---- stage 2: radix-16 over n2 -> Z[l16 + 16*k2] ----
        fft16(z);
        // ---- real-input post-processing: X[k] from Z[k] and Z[256-k] ----
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) slot[l16 + 16 * k2] = z[k2];
        if (l16 == 0) slot[256] = z[0];  // Z[256] == Z[0]: the partner index 256 - k then needs no wrap-around
        MV_WAVE_FENCE();
        float pw[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = l16 + 16 * k2;
            const cplx zp = slot[256 - k];
            const float c = tw512[2 * k], sn = tw512[2 * k + 1];
            const float ar = z[k2].re + zp.re, ai = z[k2].im - zp.im;
            const float br = z[k2].re - zp.re, bi = z[k2].im + zp.im;
            const float xr = 0.5f * (ar + c * bi - sn * br);
            const float xi = 0.5f * (ai - c * br - sn * bi);
            const float p = xr * xr + xi * xi;
            pw[k2] = a.use_power ? p : sqrtf(p);
        }
        MV_WAVE_FENCE();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pslot[l16 + 16 * k2] = pw[k2];
        MV_WAVE_FENCE();
        // ---- sparse mel filters: lane l16 owns filters l16 + 16*i ----
        float vals[ROUNDS];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            {
                const int m = l16 + 16 * i;
                const int st = mstart[m];
                const float* wr = melw + a.tab.round_off[i] * 16 + l16;
                float acc = 0.0f;
                const int width = a.tab.round_width[i];  // multiple of 4 (zero-weight padding)
                for (int j = 0; j < width; j += 4) {
                    float wv[4], pv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int kk = st + j + u;
                        kk = kk > 255 ? 255 : kk;
                        wv[u] = wr[(j + u) * 16];
                        pv[u] = pslot[kk];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc += wv[u] * pv[u];
                }
                float val = acc;
                if (a.use_log) val = logf(fmaxf(acc, 1.1920928955078125e-07f));
                vals[i] = val;
                if (fvalid && m < nbins) csum[i] += val;
            }
        }
        // ---- the quad's four rows are contiguous in the output: stage them in LDS, store 16 bytes per lane ----
        MV_WAVE_FENCE();  // every lane of the wave is done reading the power spectra
        {
            float* stage = reinterpret_cast<float*>(xbuf + (wave * 4) * FB_SLOT_CPLX);  // [4][nbins]
#pragma unroll
            for (int i = 0; i < ROUNDS; ++i) {
                const int m = l16 + 16 * i;
                if (m < nbins) stage[fs * nbins + m] = vals[i];
            }
            MV_WAVE_FENCE();
            const int frames_here = (q * 4 + 4 <= T) ? 4 : T - q * 4;
            const int nfloat = frames_here * nbins;
            float* dst = orow + (int64_t)q * 4 * nbins;
            if ((nbins & 3) == 0) {
                for (int e = lane * 4; e < nfloat; e += 256)
                    *reinterpret_cast<float4v*>(dst + e) = *reinterpret_cast<const float4v*>(stage + e);
            } else {
                for (int e = lane; e < nfloat; e += 64) dst[e] = stage[e];
            }
        }
        MV_WAVE_FENCE();
    }

    if (!a.cmn && a.lens_ratio == nullptr) return;

    // ---- per-utterance time mean (featurizer.py:79): reduce over frame slots, then over waves ----
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        float v = csum[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (fs == 0) colsum[wave * 128 + l16 + 16 * i] = v;
    }
    __syncthreads();
    float* mean = colsum + FB_WAVES * 128;
    if (tid < 128) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < FB_WAVES; ++w) v += colsum[w * 128 + tid];
        mean[tid] = a.cmn ? v / (float)T : 0.0f;
    }
    __syncthreads();  // also orders this workgroup's global stores before the reads below
    // ---- second pass over the rows this workgroup wrote: subtract mean, apply the length mask ----
    int mask_len = T;
    if (a.lens_ratio != nullptr) mask_len = (int)rintf(a.lens_ratio[b] * (float)T);  // round half to even
    if ((nbins & 3) == 0) {
        // each thread keeps one float4 column group and walks the rows: no index arithmetic in the loop
        const int q = nbins >> 2;               // float4 groups per row
        const int rows_per_pass = FB_THREADS / q;
        const int r0 = tid / q, cg = tid - r0 * q;
        if (r0 < rows_per_pass) {
            const float4v m4 = *reinterpret_cast<const float4v*>(mean + 4 * cg);
            for (int t = r0; t < T; t += rows_per_pass) {
                float4v* p = reinterpret_cast<float4v*>(orow + (int64_t)t * nbins) + cg;
                const float4v v = *p - m4;
                *p = t < mask_len ? v : float4v{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    } else {
        const int total = T * nbins;
        for (int e = tid; e < total; e += FB_THREADS) {
            const int t = e / nbins;
            const int m = e - t * nbins;
            float v = orow[e] - mean[m];
            orow[e] = (t < mask_len) ? v : 0.0f;
        }
    }
}

}  // namespace mv

// ------------------------------------------------------------------------------------------ host side

struct MvFbank {
    MvFbankCfg cfg;
    int win, shift, nbins;
    float* d_window = nullptr;
    float* d_tw256 = nullptr;
    float* d_tw512 = nullptr;
    float* d_melw = nullptr;
    int* d_mel_start = nullptr;
    mv::FbankTables tab;
    size_t smem_bytes = 0;
    int waves = 8;   // workgroup size in waves; 8 measured fastest (16 spills).  Tuning knob: MV_FBANK_WAVES = 8 | 12 | 16
};

namespace {

// Kaldi mel banks, triangles in mel space (oracle/frontend.py::kaldi_mel_banks), fp32 like torchaudio.
std::vector<std::vector<float>> kaldi_mel_banks(int num_bins, int padded, float sample_freq, float low_freq,
                                                float high_freq) {
    const int num_fft_bins = padded / 2;
    const float nyquist = 0.5f * sample_freq;
    if (high_freq <= 0.0f) high_freq += nyquist;
    const float fft_bin_width = sample_freq / padded;
    const float mel_lo = 1127.0f * logf(1.0f + low_freq / 700.0f);
    const float mel_hi = 1127.0f * logf(1.0f + high_freq / 700.0f);
    const float delta = (mel_hi - mel_lo) / (num_bins + 1);
    std::vector<std::vector<float>> banks(num_bins, std::vector<float>(num_fft_bins, 0.0f));
    for (int b = 0; b < num_bins; ++b) {
        const float left = mel_lo + b * delta, center = mel_lo + (b + 1.0f) * delta, right = mel_lo + (b + 2.0f) * delta;
        for (int k = 0; k < num_fft_bins; ++k) {
            const float mel = 1127.0f * logf(1.0f + (fft_bin_width * k) / 700.0f);
            const float up = (mel - left) / (center - left);
            const float down = (right - mel) / (right - center);
            const float w = fminf(up, down);
            banks[b][k] = w > 0.0f ? w : 0.0f;
        }
    }
    return banks;
}

template <typename T>
int upload(const std::vector<T>& v, T** dptr) {
    MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(dptr), v.size() * sizeof(T)));
    MV_HIP_OK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MV_OK;
}

}  // namespace

template <int R, int W>
hipError_t fbank_set_smem_w(size_t bytes) {
    hipError_t e = MV_SET_MAX_SMEM((mv::fbank_kernel<R, 2, W>), bytes);
    if (e != hipSuccess) return e;
    e = MV_SET_MAX_SMEM((mv::fbank_kernel<R, 1, W>), bytes);
    if (e != hipSuccess) return e;
    return MV_SET_MAX_SMEM((mv::fbank_kernel<R, 0, W>), bytes);
}

template <int R>
hipError_t fbank_set_smem(size_t bytes, int waves) {
    if (waves == 8) return fbank_set_smem_w<R, 8>(bytes);
    if (waves == 12) return fbank_set_smem_w<R, 12>(bytes);
    return fbank_set_smem_w<R, 16>(bytes);
}

template <int R, int W>
void fbank_launch_w(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a) {
    if (a.load_mode == 2) {
        MV_LAUNCH((mv::fbank_kernel<R, 2, W>), (B, 1, 1), (W * 64, 1, 1), smem, st, a);
    } else if (a.load_mode == 1) {
        MV_LAUNCH((mv::fbank_kernel<R, 1, W>), (B, 1, 1), (W * 64, 1, 1), smem, st, a);
    } else {
        MV_LAUNCH((mv::fbank_kernel<R, 0, W>), (B, 1, 1), (W * 64, 1, 1), smem, st, a);
    }
}

template <int R>
void fbank_launch(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a, int waves) {
    if (waves == 8) return fbank_launch_w<R, 8>(B, smem, st, a);
    if (waves == 12) return fbank_launch_w<R, 12>(B, smem, st, a);
    return fbank_launch_w<R, 16>(B, smem, st, a);
}

extern "C" {

void mv_fbank_default_cfg(MvFbankCfg* cfg) {
    cfg->sample_frequency = 16000.0f;
    cfg->frame_length_ms = 25.0f;
    cfg->frame_shift_ms = 10.0f;
    cfg->num_mel_bins = 23;  // torchaudio default; the reference configs set 80
    cfg->low_freq = 20.0f;
    cfg->high_freq = 0.0f;
    cfg->preemphasis_coefficient = 0.97f;
    cfg->remove_dc_offset = 1;
    cfg->use_power = 1;
    cfg->use_log_fbank = 1;
    cfg->subtract_time_mean = 1;
}

int mv_fbank_create(const MvFbankCfg* cfg, MvFbank** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_fbank_create: null argument");
    const int win = (int)(cfg->sample_frequency * cfg->frame_length_ms * 0.001f);
    const int shift = (int)(cfg->sample_frequency * cfg->frame_shift_ms * 0.001f);
    MV_REQUIRE(shift >= 1, "mv_fbank_create: frame shift must be at least one sample");
    MV_REQUIRE(cfg->num_mel_bins >= 1 && cfg->num_mel_bins <= 16 * mv::FB_MAX_ROUNDS,
               "mv_fbank_create: num_mel_bins must be in [1, 128]");
    if (win <= mv::FB_NFFT / 2 || win > mv::FB_NFFT)
        return mv::fail(MV_ERR_UNSUPPORTED,
                        "mv_fbank_create: only frame lengths that pad to a 512-point FFT (257..512 samples, e.g. 25 ms "
                        "at 16 kHz) are implemented on gfx950");
    MvFbank* h = new MvFbank();
    h->cfg = *cfg;
    h->win = win;
    h->shift = shift;
    h->nbins = cfg->num_mel_bins;

    const double pi = 3.14159265358979323846;
    std::vector<float> window(512, 0.0f), tw256(512), tw512(512);
    for (int i = 0; i < win; ++i) {
        // povey: hann(win, periodic=False) ** 0.85
        const double hann = 0.5 - 0.5 * cos(2.0 * pi * i / (win - 1));
        window[i] = (float)pow(hann, 0.85);
    }
    for (int m = 0; m < 256; ++m) {
        const int k1 = m >> 4, n2 = m & 15;  // stage-1 -> stage-2 twiddle W256^(n2*k1), laid out [k1][n2]
        tw256[2 * m] = (float)cos(2.0 * pi * ((n2 * k1) & 255) / 256.0);
        tw256[2 * m + 1] = (float)sin(2.0 * pi * ((n2 * k1) & 255) / 256.0);
        tw512[2 * m] = (float)cos(2.0 * pi * m / 512.0);
        tw512[2 * m + 1] = (float)sin(2.0 * pi * m / 512.0);
    }
    auto banks = kaldi_mel_banks(h->nbins, mv::FB_NFFT, cfg->sample_frequency, cfg->low_freq, cfg->high_freq);
    const int rounds = (h->nbins + 15) / 16;
    std::vector<int> start(rounds * 16, 0), width(rounds * 16, 0);
    for (int m = 0; m < h->nbins; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < 256; ++k)
            if (banks[m][k] > 0.0f) {
                if (lo < 0) lo = k;
                hi = k;
            }
        if (lo >= 0) {
            start[m] = lo;
            width[m] = hi - lo + 1;
        }
    }
    mv::FbankTables& tab = h->tab;
    tab.rounds = rounds;
    int off = 0;
    for (int i = 0; i < mv::FB_MAX_ROUNDS; ++i) {
        tab.round_width[i] = 0;
        tab.round_off[i] = off;
        if (i < rounds) {
            int w = 0;
            for (int l = 0; l < 16; ++l) w = width[i * 16 + l] > w ? width[i * 16 + l] : w;
            w = (w + 3) & ~3;  // the kernel walks the weights four at a time
            tab.round_width[i] = w;
            off += w;
        }
    }
    std::vector<float> melw((size_t)(off > 0 ? off : 1) * 16, 0.0f);
    for (int i = 0; i < rounds; ++i)
        for (int l = 0; l < 16; ++l) {
            const int m = i * 16 + l;
            if (m >= h->nbins) continue;
            for (int j = 0; j < width[m]; ++j) melw[(size_t)(tab.round_off[i] + j) * 16 + l] = banks[m][start[m] + j];
        }
    tab.melw_elems = (int)melw.size();
    int rc;
    if ((rc = upload(window, &h->d_window)) || (rc = upload(tw256, &h->d_tw256)) || (rc = upload(tw512, &h->d_tw512)) ||
        (rc = upload(melw, &h->d_melw)) || (rc = upload(start, &h->d_mel_start))) {
        mv_fbank_destroy(h);
        return rc;
    }
    tab.window = h->d_window;
    tab.tw256 = h->d_tw256;
    tab.tw512 = h->d_tw512;
    tab.melw = h->d_melw;
    tab.mel_start = h->d_mel_start;
    if (const char* e = getenv("MV_FBANK_WAVES")) {
        const int w = atoi(e);
        if (w == 8 || w == 12 || w == 16) h->waves = w;
    }
    h->smem_bytes = (size_t)h->waves * 4 * mv::FB_SLOT_CPLX * sizeof(mv::cplx) + 3 * 512 * sizeof(float) +
                    (size_t)(h->waves + 1) * 128 * sizeof(float) + (size_t)mv::FB_MAX_ROUNDS * 16 * sizeof(int) +
                    melw.size() * sizeof(float) + (h->waves == 8 ? (size_t)h->waves * 2048 * sizeof(float) : 0);
    hipError_t se = hipSuccess;
    switch (rounds) {
        case 1: se = fbank_set_smem<1>(h->smem_bytes, h->waves); break;
        case 2: se = fbank_set_smem<2>(h->smem_bytes, h->waves); break;
        case 3: se = fbank_set_smem<3>(h->smem_bytes, h->waves); break;
        case 4: se = fbank_set_smem<4>(h->smem_bytes, h->waves); break;
        case 5: se = fbank_set_smem<5>(h->smem_bytes, h->waves); break;
        case 6: se = fbank_set_smem<6>(h->smem_bytes, h->waves); break;
        case 7: se = fbank_set_smem<7>(h->smem_bytes, h->waves); break;
        default: se = fbank_set_smem<8>(h->smem_bytes, h->waves); break;
    }
    if (se != hipSuccess) {
        mv_fbank_destroy(h);
        return mv::fail(MV_ERR_HIP, "mv_fbank_create: cannot reserve dynamic LDS for fbank_kernel");
    }
    *out = h;
    return MV_OK;
}

int mv_fbank_destroy(MvFbank* h) {
    if (h == nullptr) return MV_OK;
    hipFree(h->d_window);
    hipFree(h->d_tw256);
    hipFree(h->d_tw512);
    hipFree(h->d_melw);
    hipFree(h->d_mel_start);
    delete h;
    return MV_OK;
}

int mv_fbank_num_frames(const MvFbank* h, int64_t num_samples, int64_t* num_frames) {
    MV_REQUIRE(h != nullptr && num_frames != nullptr, "mv_fbank_num_frames: null argument");
    *num_frames = num_samples < h->win ? 0 : 1 + (num_samples - h->win) / h->shift;
    return MV_OK;
}

int mv_fbank_forward(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                     const float* lens_ratio, float* out, mv_stream_t stream) {
    MV_REQUIRE(h != nullptr, "mv_fbank_forward: null handle");
    MV_REQUIRE(B >= 0 && L >= 0 && wav_stride >= L, "mv_fbank_forward: bad batch geometry");
    int64_t T = 0;
    mv_fbank_num_frames(h, L, &T);
    if (B == 0 || T == 0) return MV_OK;  // empty output, like kaldi.fbank on a too-short input
    MV_REQUIRE(wav != nullptr && out != nullptr, "mv_fbank_forward: null buffer");
    MV_REQUIRE(T * h->nbins < (int64_t)1 << 31, "mv_fbank_forward: utterance too long for 32-bit row indexing");
    mv::FbankArgs a;
    a.wav = wav;
    a.wav_stride = wav_stride;
    a.lens_ratio = lens_ratio;
    a.out = out;
    a.B = B;
    a.T = (int)T;
    a.win = h->win;
    a.shift = h->shift;
    a.nbins = h->nbins;
    a.preemph = h->cfg.preemphasis_coefficient;
    a.inv_win = 1.0f / (float)h->win;
    a.remove_dc = h->cfg.remove_dc_offset;
    a.use_power = h->cfg.use_power;
    a.use_log = h->cfg.use_log_fbank;
    a.cmn = h->cfg.subtract_time_mean;
    a.vec2_ok = ((reinterpret_cast<uintptr_t>(wav) & 7) == 0 && (wav_stride & 1) == 0 && (h->shift & 1) == 0 && (h->win & 1) == 0) ? 1 : 0;
    a.L = L;
    // DMA prefetch needs 16-byte aligned 4-float transfers and a quad of frames that fits the 1024-sample ring slot
    const bool dma_ok = (reinterpret_cast<uintptr_t>(wav) & 15) == 0 && (wav_stride & 3) == 0 && (h->shift & 3) == 0 &&
                        (h->win & 1) == 0 && 3 * h->shift + h->win <= 1024 && L >= 4 && h->waves == 8;  // the sample ring is only carved for 8-wave workgroups
    a.load_mode = dma_ok ? 2 : (a.vec2_ok ? 1 : 0);
    a.tab = h->tab;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (h->tab.rounds) {
        case 1: fbank_launch<1>(B, h->smem_bytes, st, a, h->waves); break;
        case 2: fbank_launch<2>(B, h->smem_bytes, st, a, h->waves); break;
        case 3: fbank_launch<3>(B, h->smem_bytes, st, a, h->waves); break;
        case 4: fbank_launch<4>(B, h->smem_bytes, st, a, h->waves); break;
        case 5: fbank_launch<5>(B, h->smem_bytes, st, a, h->waves); break;
        case 6: fbank_launch<6>(B, h->smem_bytes, st, a, h->waves); break;
        case 7: fbank_launch<7>(B, h->smem_bytes, st, a, h->waves); break;
        default: fbank_launch<8>(B, h->smem_bytes, st, a, h->waves); break;
    }
    return mv::check_launch("fbank_kernel");
}

}  // extern "C"
