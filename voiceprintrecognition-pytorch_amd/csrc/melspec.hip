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
#include <vector>

#include "kernels.h"

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
    int power_is_two;
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

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
                    a.P[orow * a.nbin_pad + bin] = a.power_is_two ? p : sqrtf(p);
                }
            }
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
    float* d_window = nullptr;
    float* d_cos = nullptr;
    float* d_sin = nullptr;
    float* d_fbT = nullptr;  // [n_mels][nbin_pad]
};

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
}

int mv_melspec_create(const MvMelSpecCfg* cfg, MvMelSpec** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_melspec_create: null argument");
    MV_REQUIRE(cfg->n_fft >= 4 && cfg->n_fft <= 8192, "mv_melspec_create: n_fft out of range");
    MV_REQUIRE(cfg->win_length >= 1 && cfg->win_length <= cfg->n_fft, "mv_melspec_create: win_length must be in [1, n_fft]");
    MV_REQUIRE(cfg->hop_length >= 1, "mv_melspec_create: hop_length must be positive");
    MV_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= 256, "mv_melspec_create: n_mels must be in [1, 256]");
    MV_REQUIRE(cfg->power == 2.0f || cfg->power == 1.0f, "mv_melspec_create: only power 1 or 2 is implemented");
    MvMelSpec* h = new MvMelSpec();
    h->cfg = *cfg;
    const int n_fft = cfg->n_fft;
    h->nbin = n_fft / 2 + 1;
    h->nbin_pad = (int)mv::round_up(h->nbin, 16);
    h->kpad = (int)mv::round_up(n_fft / 2 + 1, 16);  // folded transform length, padded to the MFMA K block
    h->pad = cfg->center ? n_fft / 2 : 0;
    const double pi = 3.14159265358979323846;
    // periodic Hann of win_length, centred in n_fft (torch.stft pads the window on both sides)
    std::vector<float> window(n_fft, 0.0f);
    const int left = (n_fft - cfg->win_length) / 2;
    for (int i = 0; i < cfg->win_length; ++i) window[left + i] = (float)(0.5 - 0.5 * cos(2.0 * pi * i / cfg->win_length));
    std::vector<float> dcos((size_t)h->nbin_pad * h->kpad, 0.0f), dsin((size_t)h->nbin_pad * h->kpad, 0.0f);
    for (int k = 0; k < h->nbin; ++k)
        for (int n = 0; n <= n_fft / 2; ++n) {
            const int64_t m = ((int64_t)k * n) % n_fft;  // exact argument reduction
            const double ang = 2.0 * pi * (double)m / n_fft;
            dcos[(size_t)k * h->kpad + n] = (float)cos(ang);
            dsin[(size_t)k * h->kpad + n] = (float)(-sin(ang));
        }
    // HTK mel filterbank, triangles in Hz (torchaudio.functional.melscale_fbanks, norm=None)
    std::vector<float> fbT((size_t)cfg->n_mels * h->nbin_pad, 0.0f);
    {
        const int n_mels = cfg->n_mels;
        const double m_min = 2595.0 * log10(1.0 + cfg->f_min / 700.0);
        const double m_max = 2595.0 * log10(1.0 + cfg->f_max / 700.0);
        std::vector<float> f_pts(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) {
            // torch.linspace in fp32, then the mel -> Hz map in fp32
            const float m = (float)(m_min + (m_max - m_min) * i / (n_mels + 1));
            f_pts[i] = 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
        }
        for (int k = 0; k < h->nbin; ++k) {
            const float f = (float)((double)(cfg->sample_rate / 2) * k / (h->nbin - 1));
            for (int j = 0; j < n_mels; ++j) {
                const float down = (f - f_pts[j]) / (f_pts[j + 1] - f_pts[j]);
                const float up = (f_pts[j + 2] - f) / (f_pts[j + 2] - f_pts[j + 1]);
                const float w = fminf(down, up);
                fbT[(size_t)j * h->nbin_pad + k] = w > 0.0f ? w : 0.0f;
            }
        }
    }
    int rc;
    if ((rc = upload_vec(window, &h->d_window)) || (rc = upload_vec(dcos, &h->d_cos)) || (rc = upload_vec(dsin, &h->d_sin)) ||
        (rc = upload_vec(fbT, &h->d_fbT))) {
        mv_melspec_destroy(h);
        return rc;
    }
    *out = h;
    return MV_OK;
}

int mv_melspec_destroy(MvMelSpec* h) {
    if (h == nullptr) return MV_OK;
    hipFree(h->d_window);
    hipFree(h->d_cos);
    hipFree(h->d_sin);
    hipFree(h->d_fbT);
    delete h;
    return MV_OK;
}

int mv_melspec_num_frames(const MvMelSpec* h, int64_t num_samples, int64_t* num_frames) {
    MV_REQUIRE(h != nullptr && num_frames != nullptr, "mv_melspec_num_frames: null argument");
    if (h->cfg.center) {
        *num_frames = 1 + num_samples / h->cfg.hop_length;
    } else {
        *num_frames = num_samples < h->cfg.n_fft ? 0 : 1 + (num_samples - h->cfg.n_fft) / h->cfg.hop_length;
    }
    return MV_OK;
}

size_t mv_melspec_workspace_bytes(const MvMelSpec* h, int32_t B, int64_t L) {
    if (h == nullptr || B <= 0) return 0;
    int64_t T = 0;
    mv_melspec_num_frames(h, L, &T);
    return (size_t)B * (size_t)T * h->nbin_pad * sizeof(float);
}

int mv_melspec_forward(const MvMelSpec* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                       const float* lens_ratio, float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    MV_REQUIRE(h != nullptr, "mv_melspec_forward: null handle");
    MV_REQUIRE(B >= 0 && L >= 0 && wav_stride >= L, "mv_melspec_forward: bad batch geometry");
    int64_t T = 0;
    mv_melspec_num_frames(h, L, &T);
    if (B == 0 || T == 0) return MV_OK;
    MV_REQUIRE(wav != nullptr && out != nullptr && workspace != nullptr, "mv_melspec_forward: null buffer");
    if (h->cfg.center) MV_REQUIRE(L > h->pad, "mv_melspec_forward: reflect padding needs more than n_fft/2 samples (torch.stft raises too)");
    MV_REQUIRE(workspace_bytes >= mv_melspec_workspace_bytes(h, B, L), "mv_melspec_forward: workspace too small");
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
    a.pad = h->pad;
    a.nbin = h->nbin;
    a.nbin_pad = h->nbin_pad;
    a.power_is_two = h->cfg.power == 2.0f;
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
