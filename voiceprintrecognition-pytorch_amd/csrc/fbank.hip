// Batched Kaldi log-mel filterbank + time-mean subtraction + length mask for gfx950.
//
// Replaces the per-utterance Python loop of KaldiFbank.forward (mvector/data_utils/featurizer.py:
// 119-132, torchaudio.compliance.kaldi.fbank) and the CMN / mask passes of AudioFeaturizer.forward
// (featurizer.py:77-90).  Arithmetic follows oracle/frontend.py::kaldi_fbank.
//
// Mapping (wave64): one frame per 16 lanes (one DPP row), four frames per wave.  The 512-point real FFT of a
// frame is a 256-point complex FFT of z[n] = x[2n] + i x[2n+1], computed as 16 x 16:
//   load     lane n2 holds z[16*n1 + n2], n1 = 0..NG-1 (its samples are the float2 at 32*n1 + 2*n2, so a 16-lane
//            group reads 128 contiguous bytes); groups beyond the window are compile-time zeros (NG = 13 for 25 ms);
//   DC / pre-emphasis: the frame mean is a DPP row reduction, the previous sample comes from the neighbouring lane
//            by a DPP row rotation -- no LDS traffic;
//   stage 1  radix-16 butterfly in registers (zero inputs pruned), twiddle W256^(n2*k1) from an LDS table;
//   transpose through a padded per-wave LDS tile (conflict-free both ways);
//   stage 2  lane k1 runs the second radix-16 butterfly -> Z[k1 + 16*k2];
//   real post-processing with the partner bin Z[256 - k] (LDS exchange), power spectrum back to the frame's slot;
//   mel      on the matrix pipe, exact fp32: v_mfma_f32_4x4x1 runs 16 independent 4 x 4 outer products per issue
//            = 16 blocks x (4 frames x 4 adjacent filters) x 1 FFT bin.  Every block walks only the bins its four
//            triangles cover (the filterbank is banded: 80 filters need 28 + 44 steps instead of a 257 x 80 product),
//            the weights come from an LDS table in operand order, the result lands as lane = filter, register =
//            frame, so the log, the column sums for CMN and 256-byte row stores need no further shuffles.
// One workgroup owns one utterance, so the per-utterance time mean is a workgroup reduction and the
// second pass (subtract mean, apply the length mask) re-reads rows this CU has just written (L2 hits).
//
// hipcc-flags: -fno-slp-vectorize -fno-signed-zeros
//   (complex math is written on float2 vectors where packed ops pay; the SLP pass would additionally pair up scalar chains
//    -- DC sums, pre-emphasis, the paired post-processing -- at the price of two v_mov per packed op; without signed zeros
//    the zero-padded inputs of the first FFT stage fold away)
#include "frontend_common.h"

#include <cstdlib>
#include <vector>

namespace mv {

constexpr int FB_NFFT = 512;
constexpr int FB_TSTRIDE = 17;        // padded row of the 16x16 transpose tile (complex elements)
constexpr int FB_SLOT_FLOATS = 548;   // per-frame LDS slot: 16*17 complex = 544 floats, padded so the four frame rows of a
                                      // wave start 36 banks apart (conflict-free 16-byte operand reads of the mel stage)
constexpr int FB_MAX_PASSES = MEL_MAX_PASSES;  // 16 blocks x 4 filters per MFMA pass: num_mel_bins <= 128

struct FbankTables {
    const float* window;    // [512] window, zero beyond the frame length
    const float* window_half;  // [512] 0.5 * window (fbank_tile_kernel: Z comes out halved, so |2X|^2 / 4 needs no scaling)
    const float* tw256;     // [16 k1][16 n2][2] cos, sin of 2 pi n2 k1 / 256 (lane-contiguous)
    const float* tw512;     // [256][2] cos, sin of 2 pi k / 512
    const float* melb;      // [steps/4][64 lanes][4] mel weights in MFMA B-operand order, passes back to back
    int melb_elems;
    int passes;                           // (copied from the MelPlan of frontend_common.h)
    int pass_steps[FB_MAX_PASSES];        // bins walked per pass (multiple of 4)
    int pass_split[FB_MAX_PASSES];        // 1, 2 or 4 adjacent blocks share one filter group (each walks a part of its bins)
    int pass_gbase[FB_MAX_PASSES];        // first filter group (4 filters) of the pass
    int pass_start[FB_MAX_PASSES][16];    // first bin of each block (multiple of 4, start + steps <= 256)
};

struct FbankArgs {
    const float* wav;
    int64_t wav_stride;
    const float* lens_ratio;
    const int64_t* num_samples;  // optional [B]: true length of every row (<= L); rows are featurised independently
    float* out;
    int B, T;
    int win, shift, nbins;
    float preemph;
    float inv_win;
    int remove_dc, use_power, use_log, cmn;
    int64_t L;
    int64_t min_len;  // rows of fewer samples have no frames (the window, or min_duration when that is longer)
    int tile_rows;    // fbank_tile_kernel: the first tile_rows (multiple of 4) frames of the utterance's [T, nbins] block stay in LDS
                      // until the time mean is known; later frames take the write / re-read / rewrite route through global memory
    // fbank_tile_kernel: the time sum of an utterance has ONE order whatever the launch form, so a row's bits depend on its own length only,
    // never on the batch size: wave slot w (of 8) adds the sums of the quads (4 frames) q = w, w + 8, w + 16, ... in that order, the
    // utterance's sum is the sum of the eight slots in order.  chunked = 0: one workgroup per utterance does exactly that.  chunked = 1:
    // grid = B * nchunks workgroups, workgroup (b, c) takes the quads [c * chunk_quads, (c + 1) * chunk_quads) (chunk_quads a multiple of 8,
    // so its wave w meets slot w's quads), writes its raw rows and every quad's column sums to part[b][q][128] (caller workspace);
    // fbank_cmn_finish_kernel adds them up in the same order, subtracts the mean and applies the mask.
    int chunked, nchunks, chunk_quads;
    float* part;
    FbankTables tab;
};

// NG = groups of 32 samples that cover the window (fbank_kernel: 13 when 384 < win <= 416, e.g. 25 ms at 16 kHz; 16 = any window up to 512.
// fbank_tile_kernel: 10 / 12 / 13 / 15 for windows of 289..320 / 353..384 / 385..416 / 449..480 samples -- 20 / 24 / 25 / 30 ms at 16 kHz --, whose
// groups 0 .. NG - 2 are full at compile time; 16 for every other window);
// VEC2: rows and frames start on 8-byte boundaries, samples are fetched as float2
template <int NG, bool VEC2, int FB_WAVES>
__global__ __launch_bounds__(FB_WAVES * 64) void fbank_kernel(FbankArgs a) {
    constexpr int FB_THREADS = FB_WAVES * 64;
    MV_DYN_SMEM(smem);
    // carve (compile-time offsets): per-frame slots | tw512 | window | stage twiddles | mel weights; the column sums of
    // the CMN epilogue reuse the slot area
    float* xbuf = reinterpret_cast<float*>(smem);                   // [FB_WAVES*4][FB_SLOT_FLOATS]
    float* tw512 = xbuf + FB_WAVES * 4 * FB_SLOT_FLOATS;            // [512]
    float* lwin = tw512 + 512;                                      // [512] window taps
    float* ltw = lwin + 512;                                        // [16][16][2] stage twiddles
    float* melb = ltw + 512;                                        // [melb_elems]
    float* colsum = xbuf;                                           // [FB_WAVES][128] then mean[128], after the frame loop

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l16 = lane & 15;   // n2 in stage 1, k1 in stage 2
    const int fs = lane >> 4;    // frame slot inside the wave
    const int b = blockIdx.x;
    // Variable-length batch (num_samples given): utterance b has its own frame count Tb; its time mean runs over those
    // frames only and rows Tb..T-1 of its output block are zero -- exactly what the reference's evaluation path produces
    // by featurising every utterance alone and zero-padding the features (reader.py:102, collate_fn.py:11-19).
    const int Tout = a.T;  // rows of the output block
    int T = a.T;
    if (a.num_samples != nullptr) {
        const int64_t ns = a.num_samples[b];
        const int64_t tb = ns < a.min_len ? 0 : 1 + (ns - a.win) / a.shift;
        T = (int)(tb < a.T ? tb : a.T);
    }
    const int nbins = a.nbins;

    for (int i = tid; i < 512; i += FB_THREADS) {
        tw512[i] = a.tab.tw512[i];
        lwin[i] = a.tab.window[i];
        ltw[i] = a.tab.tw256[i];
    }
    for (int i = tid * 4; i < a.tab.melb_elems; i += FB_THREADS * 4)
        *reinterpret_cast<float4v*>(melb + i) = *reinterpret_cast<const float4v*>(a.tab.melb + i);
    __syncthreads();

    float* wslots = xbuf + wave * 4 * FB_SLOT_FLOATS;
    float* pslot = wslots + fs * FB_SLOT_FLOATS;
    cplx* slot = reinterpret_cast<cplx*>(pslot);
    const float* wrow = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * Tout * nbins;

    // mel stage: this lane is (block = lane/4, j = lane%4): A operand = power of frame j, D = filter 4*(16*pass + block) + j
    const float* arow = wslots + (lane & 3) * FB_SLOT_FLOATS;
    int mstart[FB_MAX_PASSES];
#pragma unroll
    for (int p = 0; p < FB_MAX_PASSES; ++p) mstart[p] = a.tab.pass_start[p][lane >> 2];
    float csum[FB_MAX_PASSES];
#pragma unroll
    for (int p = 0; p < FB_MAX_PASSES; ++p) csum[p] = 0.0f;

    const int nquads = (T + 3) >> 2;
    // load: lane holds samples 32*n1 + 2*l16 (+1) and, for the pre-emphasis, the sample before them (one more 4-byte load
    // that hits the lines just fetched; d[-1] := d[0]).  A group that reaches beyond the window has its indices clamped
    // (the row may end with the frame) and the surplus values zeroed.
    auto load_quad = [&](int q, cplx (&e)[NG], float (&eprev)[NG]) {
        const int f_raw = q * 4 + fs;
        const int f = f_raw < T ? f_raw : T - 1;  // surplus slots recompute the last frame; nothing of theirs is kept
        const float* fp = wrow + (int64_t)f * a.shift;
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const int idx = 32 * n1 + 2 * l16;
            // NG == 13 is only launched for 384 < win <= 416: groups 0..11 are full at compile time
            const bool full = NG == 13 ? n1 < 12 : 32 * n1 + 32 <= a.win;
            if (full) {
                if (VEC2) {
                    e[n1] = cload(fp + idx);
                } else {
                    e[n1] = cmake(fp[idx], fp[idx + 1]);
                }
                eprev[n1] = fp[n1 == 0 ? (idx > 0 ? idx - 1 : 0) : idx - 1];
            } else {
                const int i0 = idx < a.win ? idx : a.win - 1, i1 = idx + 1 < a.win ? idx + 1 : a.win - 1;
                cplx v;
                if (VEC2) {  // win is even: idx < win implies idx + 1 < win
                    v = cload(fp + (idx < a.win ? idx : a.win - 2));
                } else {
                    v = cmake(fp[i0], fp[i1]);
                }
                e[n1] = v;  // surplus values are zeroed by mask_tail() when the group is consumed
                eprev[n1] = fp[i0 > 0 ? i0 - 1 : 0];
            }
        }
    };

    auto mask_tail = [&](cplx (&e)[NG]) {  // kept apart from the loads: a select right behind a prefetch would wait for it
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const int idx = 32 * n1 + 2 * l16;
            const bool full = NG == 13 ? n1 < 12 : 32 * n1 + 32 <= a.win;
            if (!full) e[n1] = cmake(idx < a.win ? e[n1][0] : 0.0f, idx + 1 < a.win ? e[n1][1] : 0.0f);
        }
    };

    // PREFETCH (workgroups of <= 12 waves, 170 VGPRs): the next quad's samples are requested before this quad is
    // processed, so their latency hides under ~800 VALU instructions instead of stalling the top of every iteration
    constexpr bool PREFETCH = FB_WAVES <= 12;
    cplx enext[NG];
    float epnext[NG];
    if (PREFETCH && wave < nquads) load_quad(wave, enext, epnext);
    for (int q = wave; q < nquads; q += FB_WAVES) {
        cplx e[NG];
        float eprev[NG];
        if (PREFETCH) {
#pragma unroll
            for (int n1 = 0; n1 < NG; ++n1) {
                e[n1] = enext[n1];
                eprev[n1] = epnext[n1];
            }
            if (q + FB_WAVES < nquads) load_quad(q + FB_WAVES, enext, epnext);
        } else {
            load_quad(q, e, eprev);
        }
        mask_tail(e);
        // ---- DC removal (frame mean over the `win` samples), pre-emphasis y[j] = d[j] - c d[j-1] and window:
        //      (x[j] - mu) - c (x[j-1] - mu) = x[j] - c x[j-1] - (1 - c) mu.  Taps beyond the window meet a zero weight ----
        float dc = 0.0f;
        if (a.remove_dc) {
            cplx s2 = cmake(0.0f, 0.0f);
#pragma unroll
            for (int n1 = 0; n1 < NG; ++n1) s2 += e[n1];
            dc = row16_sum(s2[0] + s2[1]) * a.inv_win * (1.0f - a.preemph);
        }
        cplx z[16];
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const cplx prev = cmake(eprev[n1], e[n1][0]);
            const cplx y = e[n1] - cscale(prev, a.preemph) - cmake(dc, dc);
            const float2v w2 = *reinterpret_cast<const float2v*>(lwin + 32 * n1 + 2 * l16);
            z[n1] = cmul_elem(y, w2[0], w2[1]);
        }
#pragma unroll
        for (int n1 = NG; n1 < 16; ++n1) z[n1] = cmake(0.0f, 0.0f);
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
        // ---- real-input post-processing: X[k] from Z[k] and Z[256-k]; |X|^2 = (|2X|^2) / 4 ----
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) slot[l16 + 16 * k2] = z[k2];
        if (l16 == 0) slot[256] = z[0];  // Z[256] == Z[0]: the partner index 256 - k then needs no wrap-around
        MV_WAVE_FENCE();
        float pw[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = l16 + 16 * k2;
            const cplx zp = slot[256 - k];
            const float2v cs = *reinterpret_cast<const float2v*>(tw512 + 2 * k);
            const cplx za = z[k2] + cconj(zp);  // Z[k] + conj(Z[256-k])
            const cplx zb = z[k2] - cconj(zp);  // Z[k] - conj(Z[256-k])
            // 2 X[k] = za - i (c - i s) zb
            const cplx x2 = za + cmul_elem(cswap(zb), cs[0], -cs[0]) - cscale(zb, cs[1]);
            pw[k2] = 0.25f * (x2[0] * x2[0] + x2[1] * x2[1]);
        }
        if (!a.use_power) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) pw[k2] = sqrtf(pw[k2]);
        }
        MV_WAVE_FENCE();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pslot[l16 + 16 * k2] = pw[k2];
        MV_WAVE_FENCE();
        // ---- mel filters on the matrix pipe + log + row stores ----
        const int frames_here = (q * 4 + 4 <= T) ? 4 : T - q * 4;
        int moff = 0;
#pragma unroll
        for (int p = 0; p < FB_MAX_PASSES; ++p) {
            if (p < a.tab.passes) {
                float4v acc = float4v{0.0f, 0.0f, 0.0f, 0.0f}, acc2 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
                const int ngrp = a.tab.pass_steps[p] >> 2;
                const float* ap = arow + mstart[p];
                const float* bp = melb + moff * 64 + lane * 4;
                for (int g = 0; g < ngrp; ++g) {
                    const float4v av = *reinterpret_cast<const float4v*>(ap + 4 * g);
                    const float4v bv = *reinterpret_cast<const float4v*>(bp + g * 256);
                    acc = fb_mfma4(av[0], bv[0], acc);
                    acc2 = fb_mfma4(av[1], bv[1], acc2);
                    acc = fb_mfma4(av[2], bv[2], acc);
                    acc2 = fb_mfma4(av[3], bv[3], acc2);
                }
                acc += acc2;
                moff += a.tab.pass_steps[p];
                // blocks that share a filter group hold partial sums over disjoint bin ranges: add them up
                const int split = a.tab.pass_split[p];
                if (split >= 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], 4);
                }
                if (split == 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], 8);
                }
                const int blk = lane >> 2;
                const int m = 4 * (a.tab.pass_gbase[p] + blk / split) + (lane & 3);  // filter of this lane
                if (m < nbins && (blk & (split - 1)) == 0) {
                    float val[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)  // the clamp keeps the argument normal: one v_log_f32 (log2) and a scale
                        val[r] = a.use_log ? fb_log2(fmaxf(acc[r], 1.1920928955078125e-07f)) * 0.69314718055994531f : acc[r];
                    float* dst = orow + (int64_t)q * 4 * nbins + m;
                    if (frames_here == 4) {
                        csum[p] += (val[0] + val[1]) + (val[2] + val[3]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r * nbins] = val[r];
                    } else {
                        for (int r = 0; r < frames_here; ++r) {
                            csum[p] += val[r];
                            dst[r * nbins] = val[r];
                        }
                    }
                }
            }
        }
        MV_WAVE_FENCE();  // the power spectra have been consumed before the next quad's transposes overwrite them
    }

    if (!a.cmn && a.lens_ratio == nullptr && a.num_samples == nullptr) return;

    // ---- per-utterance time mean (featurizer.py:79): lane = filter already, reduce over waves ----
    __syncthreads();  // every wave has left the frame loop: the slot area becomes the reduction buffer
#pragma unroll
    for (int p = 0; p < FB_MAX_PASSES; ++p) {
        if (p < a.tab.passes) {
            const int split = a.tab.pass_split[p], blk = lane >> 2;
            const int m = 4 * (a.tab.pass_gbase[p] + blk / split) + (lane & 3);
            if (m < 128 && (blk & (split - 1)) == 0) colsum[wave * 128 + m] = csum[p];  // every filter has exactly one owner lane
        }
    }
    __syncthreads();
    float* mean = colsum + FB_WAVES * 128;
    if (tid < 128) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < FB_WAVES; ++w) v += colsum[w * 128 + tid];
        mean[tid] = (a.cmn && T > 0) ? v / (float)T : 0.0f;
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
            for (int t = r0; t < Tout; t += rows_per_pass) {  // rows T..Tout-1 of a shorter utterance become zeros
                float4v* p = reinterpret_cast<float4v*>(orow + (int64_t)t * nbins) + cg;
                const float4v v = *p - m4;
                *p = t < mask_len ? v : float4v{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    } else {
        const int total = Tout * nbins;
        for (int e = tid; e < total; e += FB_THREADS) {
            const int t = e / nbins;
            const int m = e - t * nbins;
            float v = orow[e] - mean[m];
            orow[e] = (t < mask_len) ? v : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fbank_tile_kernel: the same arithmetic with the issue slots and LDS cycles cut to what the algorithm needs.
//
// Measured on the kernel above (profiles/r01m, r01n, r02h): VALU 39 % busy, LDS 30 %, waves 53 % parked, 1.62 x the
// algorithmic HBM traffic (the CMN pass re-reads and rewrites the features).  Per wave iteration (4 frames) it issues
// ~1280 VALU lane-ops and ~690 LDS cycles; one LDS cycle is shared by the CU's four SIMDs, so an LDS cycle costs as
// much machine time as two VALU ops.  This kernel (same frame -> 16-lane mapping, 8 waves of <= 256 VGPRs):
//   * every per-lane constant lives in registers for the whole utterance: window taps, stage twiddles, post-processing
//     twiddles and the mel weights in MFMA operand order (no table reads in the loop);
//   * real-FFT post-processing in PAIRS (k, 256 - k): both bins come from the same two values Z[k], Z[256-k]; a lane
//     owns 8 pairs instead of 16 single bins.  The partner value sits in lane 16 - k1 of the same 16-lane row: two DPP
//     moves (row_mirror, then row_shr:1 whose lane 0 keeps `old` = its own register, exactly what k1 = 0 needs) -- no
//     LDS exchange.  The window is pre-scaled by 1/2, so |2X|^2 / 4 = |2 X_half|^2 needs no scaling;
//   * one LDS transpose per frame, in a layout shared by the wave's four frames: row k1 = 64 elements [frame][n2] + one
//     pad -> lane-linear conflict-free writes, conflict-free reads, one address register each way;
//   * the [T, nbins] log-mel block of the utterance stays in LDS (95 KB for 3 s x 80 bins) until the time mean is
//     known: features are written to HBM once (algorithmic traffic), not written, re-read and rewritten.  Longer
//     utterances fall back to the second pass over global memory inside the same kernel.
// Instantiated for the mel geometry of the reference configurations (80 bins, 16 kHz, 512-point FFT: passes of 7 and 3
// four-bin groups); any other geometry runs fbank_kernel above.
constexpr int FBT_WAVES = 8;
constexpr int FBT_ROW = 65;                      // complex elements per transpose row: [4 frames][16] + 1 pad
constexpr int FBT_SLOT_FLOATS = 16 * FBT_ROW * 2;  // 2080 floats = 8320 B per wave
constexpr int fbt_win_floats(int ng) { return 32 * ng; }   // window taps kept in LDS: the taps the NG sample groups touch
constexpr int FBT_PSTR = 292;                    // floats between the power rows of the wave's four frames (36 banks apart)

template <int NG, bool VEC2, int G0, int G1>
__global__ __launch_bounds__(FBT_WAVES * 64) void fbank_tile_kernel(FbankArgs a) {
    constexpr int THREADS = FBT_WAVES * 64;
    MV_DYN_SMEM(smem);
    float* xbuf = reinterpret_cast<float*>(smem);                  // [FBT_WAVES][FBT_SLOT_FLOATS]
    constexpr int WINF = fbt_win_floats(NG);
    float* lwin = xbuf + FBT_WAVES * FBT_SLOT_FLOATS;              // [32 NG] 0.5 * window
    float* ltw1 = lwin + WINF;                                     // [16 k1][16 n2][2] stage twiddles
    float* tile = ltw1 + 512;                                      // [tile_rows][nbins]
    float* colsum = xbuf;                                          // [FBT_WAVES][128] then mean[128], after the frame loop

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = MV_UNIFORM(tid >> 6);   // scalar: the quad / chunk bookkeeping below runs on the scalar unit
    const int l16 = lane & 15;
    const int fs = lane >> 4;
    const int b = a.chunked ? (int)blockIdx.x / a.nchunks : (int)blockIdx.x;
    const int chunk = a.chunked ? (int)blockIdx.x - b * a.nchunks : 0;
    const int Tout = a.T;
    int T = a.T;
    if (a.num_samples != nullptr) {
        const int64_t ns = a.num_samples[b];
        const int64_t tb = ns < a.min_len ? 0 : 1 + (ns - a.win) / a.shift;
        T = (int)(tb < a.T ? tb : a.T);
    }
    const int nbins = a.nbins;

    // ---- per-lane constants (registers) ----
    for (int i = tid; i < WINF; i += THREADS) lwin[i] = a.tab.window_half[i];
    const float* cwin = lwin + 2 * l16;  // 0.5 * window at samples 32 n1 + 2 l16 (+1): one 8-byte LDS read per group
    for (int i = tid; i < 512; i += THREADS) ltw1[i] = a.tab.tw256[i];
    const float* ctw1 = ltw1 + 2 * l16;  // W256^(l16 k1) as (cos, sin) at + 32 k1: one 8-byte LDS read per twiddle
    float2v ctw2[8];        // (cos, sin) of pi k / 256 at k = l16 + 16 j
#pragma unroll
    for (int j = 0; j < 8; ++j) ctw2[j] = *reinterpret_cast<const float2v*>(a.tab.tw512 + 2 * (l16 + 16 * j));
    float4v mb0[G0], mb1[G1];  // mel weights of this lane's (block, filter): 4 bins per group
#pragma unroll
    for (int g = 0; g < G0; ++g) mb0[g] = *reinterpret_cast<const float4v*>(a.tab.melb + (size_t)g * 256 + lane * 4);
#pragma unroll
    for (int g = 0; g < G1; ++g) mb1[g] = *reinterpret_cast<const float4v*>(a.tab.melb + (size_t)(G0 + g) * 256 + lane * 4);

    float* wslot = xbuf + wave * FBT_SLOT_FLOATS;
    cplx* tw_write = reinterpret_cast<cplx*>(wslot) + lane;                       // element (k1, frame fs, n2 = l16) at + k1 * FBT_ROW
    const cplx* tw_read = reinterpret_cast<const cplx*>(wslot) + l16 * FBT_ROW + 16 * fs;  // element (k1 = l16, fs, n2) at + n2
    float* prow = wslot + fs * FBT_PSTR;                                          // power row of this lane's frame
    float* p_own = prow + l16;                                                    // bin l16 + 16 j at + 16 j
    float* p_par = prow + (16 - l16);                                             // bin 256 - k = (16 - l16) + 16 (15 - j) at + 16 (15 - j)
    // lane 0 of a row: its j = 0 partner slot would be the Nyquist bin (zero mel weight); it stores bin 128 there instead
    float* p_par0 = l16 == 0 ? prow + 128 - 240 : p_par;
    const float* wrow = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * Tout * nbins;

    // mel stage: lane = (block = lane / 4, i = lane % 4): A operand = power of frame i, D = 4 frames x filter
    const float* arow = wslot + (lane & 3) * FBT_PSTR;
    const float* ap0 = arow + a.tab.pass_start[0][lane >> 2];
    const float* ap1 = arow + a.tab.pass_start[1][lane >> 2];
    const int blk = lane >> 2;
    const int split1 = a.tab.pass_split[1];
    const int m0 = 4 * (a.tab.pass_gbase[0] + blk) + (lane & 3);                // pass 0: one block per filter group
    const int m1 = 4 * (a.tab.pass_gbase[1] + blk / split1) + (lane & 3);
    const bool own0 = m0 < nbins, own1 = m1 < nbins && (blk & (split1 - 1)) == 0;
    float csum0 = 0.0f, csum1 = 0.0f;   // this wave's column sums (slot `wave` of FbankArgs' summation order)
    const int tile_rows = a.tile_rows;

    __syncthreads();  // window taps
    const int nquads_all = (T + 3) >> 2;
    const int qbeg = a.chunked ? chunk * a.chunk_quads : 0;                                   // this workgroup's quads: [qbeg, nquads)
    const int nquads = a.chunked && qbeg + a.chunk_quads < nquads_all ? qbeg + a.chunk_quads : nquads_all;
    float* part_b = a.chunked ? a.part + (int64_t)b * (a.nchunks * a.chunk_quads) * 128 : nullptr;
    // samples of one quad: lane holds {x[j-1], x[j], x[j+1]} at j = 32 n1 + 2 l16 -- the sample pair and, for the
    // pre-emphasis, the sample before it -- as ONE 12-byte load per group whose three result registers are consumed as
    // they are.  (Loading the pair and the previous sample as separate values made the compiler merge them into the same
    // 12-byte load and then copy the components out right behind it: a full-latency wait directly after the prefetch was
    // issued, which is why prefetching showed no gain in r03c.)
    auto load_quad = [&](int q, float3u (&r)[NG]) {
        const int f_raw = q * 4 + fs;
        const int f = f_raw < T ? f_raw : T - 1;  // surplus slots recompute the last frame; nothing of theirs is kept
        const float* fp = wrow + (int64_t)f * a.shift;
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const int idx = 32 * n1 + 2 * l16;
            const bool full = NG < 16 ? n1 < NG - 1 : 32 * n1 + 32 <= a.win;   // NG < 16 is launched for 32 (NG - 1) < win <= 32 NG only
            // first group: x[-1] := x[0] (replicate); groups that reach beyond the window: clamped (the row may end with the
            // frame), their surplus taps are zeroed when the group is consumed
            int j = idx;
            if (!full) j = idx < a.win ? idx : a.win - 2;
            // first group, lane 0: x[-1] := x[0] (replicate) -- it loads {x[0], x[1], x[2]} and picks its pair when the group is consumed
            // (one 12-byte load like every other lane: a lane-dependent {x[0], x[0], x[1]} assembled from two loads made the compiler copy
            // the components out right behind the prefetch, i.e. wait for it)
            const int jl = (n1 == 0 && j == 0) ? 1 : j;
            r[n1] = load_f32x3(fp + jl - 1);
        }
    };
    // The next quad's samples are requested as soon as this quad's have been consumed (window stage): their HBM latency runs
    // under the two FFTs, the post-processing and the mel stage instead of stalling the top of every iteration (two waves
    // per SIMD cannot hide it: PMC r03b, waves 61 % parked with the VALU 30 % busy).  Two register sets used alternately --
    // the loop below is unrolled by two -- so the prefetched values are consumed where the loads put them.
    auto process_quad = [&](int q, float3u (&r)[NG], float3u (&r_next)[NG]) __attribute__((always_inline)) {
        float x0[NG], x1[NG];
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const int idx = 32 * n1 + 2 * l16;
            const bool full = NG < 16 ? n1 < NG - 1 : 32 * n1 + 32 <= a.win;   // NG < 16 is launched for 32 (NG - 1) < win <= 32 NG only
            x0[n1] = full || idx < a.win ? r[n1][1] : 0.0f;
            x1[n1] = full || idx + 1 < a.win ? r[n1][2] : 0.0f;
        }
        if (l16 == 0) {  // (see load_quad: lane 0 of a frame row holds {x[0], x[1], x[2]} in its first group)
            x1[0] = r[0][1];
            x0[0] = r[0][0];
        }
        // ---- DC removal, pre-emphasis, window (see fbank_kernel) ----
        float dc = 0.0f;
        if (a.remove_dc) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int n1 = 0; n1 < NG; ++n1) {
                s0 += x0[n1];
                s1 += x1[n1];
            }
            dc = row16_sum(s0 + s1) * a.inv_win * (1.0f - a.preemph);
        }
        cplx z[16];
        const float npre = -a.preemph;
#pragma unroll
        for (int n1 = 0; n1 < NG; ++n1) {
            const float2v w2 = lds_load_unmerged(reinterpret_cast<const float2v*>(cwin + 32 * n1));
            const float y0 = fmaf(npre, r[n1][0], x0[n1]) - dc;   // taps beyond the window meet a zero weight
            const float y1 = fmaf(npre, x0[n1], x1[n1]) - dc;
            z[n1] = cmake(y0 * w2[0], y1 * w2[1]);
        }
#pragma unroll
        for (int n1 = NG; n1 < 16; ++n1) z[n1] = cmake(0.0f, 0.0f);
        if (q + FBT_WAVES < nquads) load_quad(q + FBT_WAVES, r_next);
        // ---- stage 1 + twiddle ----
        fft16(z);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            const float2v tw = lds_load_unmerged(reinterpret_cast<const float2v*>(ctw1 + 32 * k1));
            z[k1] = cmul_conjtw(z[k1], tw[0], tw[1]);
        }
        // ---- the one transpose ----
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tw_write[k1 * FBT_ROW] = z[k1];
        MV_WAVE_FENCE();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = lds_read_single(tw_read + n2);
        MV_WAVE_FENCE();
        // ---- stage 2 -> z[k2] = Z[l16 + 16 k2] (halved) ----
        fft16(z);
        // ---- paired real-input post-processing ----
        //   A = Z[k], B = Z[256-k], w = exp(-i pi k / 256):  za = A + conj B, zb = A - conj B, r = w zb,
        //   2X[k] = za - i r,  2X[256-k] = conj(za) - i conj(r)
        float pk[8], pp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx own_alt = z[(16 - j) & 15];  // lane 0 (k1 = 0): partner Z[16 (16 - j)] is its own register; j = 0: Z[256] = Z[0]
            const cplx src = z[15 - j];
            cplx bp;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float t = dpp_mov_all<DPP_ROW_MIRROR>(src[c]);
                bp[c] = dpp_mov<DPP_ROW_SHR1>(own_alt[c], t);
            }
            const cplx av = z[j];
            const float zar = av[0] + bp[0], zai = av[1] - bp[1];
            const float zbr = av[0] - bp[0], zbi = av[1] + bp[1];
            const float c = ctw2[j][0], s = ctw2[j][1];      // w = c - i s
            const float rr = zbr * c + zbi * s, ri = zbi * c - zbr * s;
            const float u1 = zar + ri, u2 = zai - rr;        // 2X[k]
            const float v1 = zar - ri, v2 = zai + rr;        // conj of 2X[256-k]
            pk[j] = u1 * u1 + u2 * u2;
            pp[j] = v1 * v1 + v2 * v2;
        }
        {   // bin 128 = conj(Z[128]) lives in lane 0's register 8; it takes the slot of lane 0's (unused) Nyquist output
            const float p128 = 4.0f * (z[8][0] * z[8][0] + z[8][1] * z[8][1]);  // |X[128]|^2 = |Z[128]|^2, and z is Z / 2
            pp[0] = l16 == 0 ? p128 : pp[0];
        }
        // (the transposed values were all read before the FFT: the power rows may overwrite them)
#pragma unroll
        for (int j = 0; j < 8; ++j) p_own[16 * j] = pk[j];
        p_par0[16 * 15] = pp[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) p_par[16 * (15 - j)] = pp[j];
        MV_WAVE_FENCE();
        // ---- mel filters on the matrix pipe (weights in registers) ----
        // four independent accumulator chains per pass (the matrix pipe never waits for its own result), each started
        // from the constant zero operand
        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        float4v acc0[4], acc1[4];
#pragma unroll
        for (int g = 0; g < G0; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap0 + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc0[c] = fb_mfma4(av[c], mb0[g][c], g == 0 ? zero4 : acc0[c]);
        }
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const float4v av = *reinterpret_cast<const float4v*>(ap1 + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc1[c] = fb_mfma4(av[c], mb1[g][c], g == 0 ? zero4 : acc1[c]);
        }
        const float4v a0 = (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
        float4v a1 = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
        MV_WAVE_FENCE();  // the power rows are consumed: the next quad's transpose may overwrite them
        // pass 1: `split1` adjacent blocks hold partial sums of one filter group over disjoint bin ranges; the first of
        // them collects the others (lanes 4 / 8 / 12 further up in the same 16-lane row)
        if (split1 >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] += dpp_mov<DPP_ROW_SHL4>(0.0f, a1[r]);
        }
        if (split1 == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] += dpp_mov<DPP_ROW_SHL8>(0.0f, a1[r]);
        }
        const int frames_here = (q * 4 + 4 <= T) ? 4 : T - q * 4;
        float v0[4], v1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[r] = fb_log2(fmaxf(a0[r], 1.1920928955078125e-07f)) * 0.69314718055994531f;
            v1[r] = fb_log2(fmaxf(a1[r], 1.1920928955078125e-07f)) * 0.69314718055994531f;
        }
        if (frames_here < 4) {  // last quad of the utterance: surplus frames are neither summed nor kept
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                if (r >= frames_here) {
                    v0[r] = 0.0f;
                    v1[r] = 0.0f;
                }
            }
        }
        const float qs0 = (v0[0] + v0[1]) + (v0[2] + v0[3]), qs1 = (v1[0] + v1[1]) + (v1[2] + v1[3]);
        csum0 += qs0;
        csum1 += qs1;
        if (a.chunked) {  // uniform: the quad's column sums for the finish pass (FbankArgs)
            if (own0) part_b[q * 128 + m0] = qs0;
            if (own1) part_b[q * 128 + m1] = qs1;
        }
        const int row0 = q * 4 * nbins;
        if (q * 4 < tile_rows) {  // uniform (tile_rows is a multiple of 4): LDS block [t][m]
            auto d0 = MV_AS_LDS(float, tile + row0 + m0);
            auto d1 = MV_AS_LDS(float, tile + row0 + m1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < frames_here) {
                    if (own0) d0[r * nbins] = v0[r];
                    if (own1) d1[r * nbins] = v1[r];
                }
            }
        } else {
            auto d0 = MV_AS_GLOBAL(float, orow + row0 + m0);
            auto d1 = MV_AS_GLOBAL(float, orow + row0 + m1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < frames_here) {
                    if (own0) d0[r * nbins] = v0[r];
                    if (own1) d1[r * nbins] = v1[r];
                }
            }
        }
    };
    float3u ra[NG], rb[NG];
    if (qbeg + wave < nquads) load_quad(qbeg + wave, ra);
    for (int q = qbeg + wave; q < nquads; q += 2 * FBT_WAVES) {
        process_quad(q, ra, rb);
        if (q + FBT_WAVES < nquads) process_quad(q + FBT_WAVES, rb, ra);
    }
    if (a.chunked) return;  // uniform: mean, mask and zero rows belong to fbank_cmn_finish_kernel

    const bool second_pass = a.cmn || a.lens_ratio != nullptr || a.num_samples != nullptr;
    if (!second_pass && tile_rows == 0) return;

    // ---- per-utterance time mean (featurizer.py:79) ----
    __syncthreads();  // every wave has left the frame loop: the slot area becomes the reduction buffer
    if (own0) colsum[wave * 128 + m0] = csum0;   // (lanes that own no filter summed values nobody reads)
    if (own1) colsum[wave * 128 + m1] = csum1;
    __syncthreads();
    float* mean = colsum + FBT_WAVES * 128;
    if (tid < 128) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < FBT_WAVES; ++w) v += colsum[w * 128 + tid];
        mean[tid] = (a.cmn && T > 0 && tid < nbins) ? v / (float)T : 0.0f;
    }
    __syncthreads();  // also orders this workgroup's stores (LDS block / global rows) before the reads below
    // ---- subtract the mean, apply the length mask, write the features (once, when the block sat in LDS) ----
    int mask_len = T;
    if (a.lens_ratio != nullptr) mask_len = (int)rintf(a.lens_ratio[b] * (float)T);  // round half to even
    const int qn = nbins >> 2;              // float4 groups per row (nbins % 4 == 0 for this kernel)
    const int rows_per_pass = THREADS / qn;
    const int r0 = tid / qn, cg = tid - r0 * qn;
    if (r0 < rows_per_pass) {
        const float4v m4 = *reinterpret_cast<const float4v*>(mean + 4 * cg);
        const float4v zero4 = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        int t = r0;
        for (; t < Tout && t < tile_rows; t += rows_per_pass) {  // rows held in LDS: written to HBM once
            const float4v raw = t < T ? *(reinterpret_cast<const float4v*>(tile + t * nbins) + cg) : zero4;
            *(reinterpret_cast<float4v*>(orow + (int64_t)t * nbins) + cg) = t < mask_len ? raw - m4 : zero4;
        }
        if (second_pass) {
            for (; t < Tout; t += rows_per_pass) {  // rows that went through global memory (rows T..Tout-1 of a shorter utterance become zeros)
                float4v* p = reinterpret_cast<float4v*>(orow + (int64_t)t * nbins) + cg;
                const float4v raw = t < T ? *p : zero4;
                *p = t < mask_len ? raw - m4 : zero4;
            }
        }
    }
}

// second launch of the chunked form: feat[b, t, :] = t < mask_len ? raw - mean : 0 (featurizer.py:79, 119-132) with the time sum formed
// exactly as the one-workgroup form forms it (FbankArgs): per wave slot the quad sums q = w, w + 8, ... in order, then the eight slots in
// order.  Workgroup = (utterance, block of FBF_ROWS rows); one thread per group of 4 bins.
constexpr int FBF_ROWS = 32;
__global__ __launch_bounds__(256) void fbank_cmn_finish_kernel(float* out, const float* part, const float* lens_ratio, int T, int nbins,
                                                               int part_quads, int row_blocks, int cmn) {
    __shared__ float slot[FBT_WAVES][128];
    __shared__ float mean[128];
    const int b = (int)blockIdx.x / row_blocks, rb = (int)blockIdx.x - b * row_blocks;
    const int tid = threadIdx.x;
    const int nquads = (T + 3) >> 2;
    const float* pb = part + (int64_t)b * part_quads * 128;
    for (int i = tid; i < FBT_WAVES * 128; i += 256) {  // thread = (slot w, bin m)
        const int w = i >> 7, m = i & 127;
        float run = 0.0f;
        if (cmn && m < nbins) {
            int q = w;
            for (; q + 7 * FBT_WAVES < nquads; q += 8 * FBT_WAVES) {   // eight loads in flight, added in order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pb[(q + u * FBT_WAVES) * 128 + m];
#pragma unroll
                for (int u = 0; u < 8; ++u) run += v[u];
            }
            for (; q < nquads; q += FBT_WAVES) run += pb[q * 128 + m];
        }
        slot[w][m] = run;
    }
    __syncthreads();
    if (tid < 128) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < FBT_WAVES; ++w) v += slot[w][tid];
        mean[tid] = (cmn && T > 0 && tid < nbins) ? v / (float)T : 0.0f;
    }
    __syncthreads();
    int mask_len = T;
    if (lens_ratio != nullptr) mask_len = (int)rintf(lens_ratio[b] * (float)T);  // round half to even
    const int qn = nbins >> 2;
    const int t1 = (rb + 1) * FBF_ROWS < T ? (rb + 1) * FBF_ROWS : T;
    const int64_t row0 = (int64_t)b * T;
    for (int i = rb * FBF_ROWS * qn + tid; i < t1 * qn; i += 256) {
        const int t = i / qn, cg = i - t * qn;
        const float4v m4 = *reinterpret_cast<const float4v*>(mean + 4 * cg);
        float4v* p = reinterpret_cast<float4v*>(out + (row0 + t) * nbins) + cg;
        *p = t < mask_len ? *p - m4 : float4v{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

// use_energy (torchaudio.compliance.kaldi.fbank: `_get_window`'s signal_log_energy, concatenated in front of -- htk_compat: behind -- the mel columns before
// subtract_mean): the mel kernels have written their [B, T, nbins] rows (time mean and mask applied) to the caller workspace; this pass writes the
// [B, T, nbins + 1] output: the mel columns copied, the energy column = log(max(sum of squares of the frame, eps)) -- of the frame after the DC removal
// (raw_energy) or of the pre-emphasised, windowed frame -- raised to log(energy_floor), minus its own time mean over the utterance's frames, zero from the
// mask length on.  One workgroup per utterance, thread i takes the frames i, i + 256, ...; a frame's sums run in four interleaved accumulators, the
// utterance's mean in one fixed order (per-thread sums in frame order, then a tree over the 256 threads): a row's bits do not depend on the batch.
// An option no shipped configuration sets: written for clarity, not for speed (a frame's samples are read twice by one thread).
struct FbankEnergyArgs {
    const float* wav;
    int64_t wav_stride;
    const float* lens_ratio;
    const int64_t* num_samples;
    const float* mel;      // [B, T, nbins]
    float* out;            // [B, T, nbins + 1]
    const float* window;   // [512]
    int T, win, shift, nbins;
    float preemph, inv_win;
    int remove_dc, raw_energy, cmn, energy_col, mel_col, has_floor;
    float log_floor;
    int64_t min_len;
};

__global__ __launch_bounds__(256) void fbank_energy_kernel(FbankEnergyArgs a) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Tout = a.T;
    int T = a.T;
    if (a.num_samples != nullptr) {   // (as in the mel kernels)
        const int64_t ns = a.num_samples[b];
        const int64_t tb = ns < a.min_len ? 0 : 1 + (ns - a.win) / a.shift;
        T = (int)(tb < a.T ? tb : a.T);
    }
    const int ld = a.nbins + 1;
    const float* wrow = a.wav + (int64_t)b * a.wav_stride;
    float* orow = a.out + (int64_t)b * Tout * ld;
    const float* mrow = a.mel + (int64_t)b * Tout * a.nbins;
    const float eps = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps (kaldi.fbank's EPSILON)
    float part = 0.0f;
    for (int t = tid; t < T; t += 256) {
        const float* x = wrow + (int64_t)t * a.shift;
        float mean = 0.0f;
        if (a.remove_dc) {
            float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int i = 0; i < a.win; ++i) s[i & 3] += x[i];
            mean = ((s[0] + s[1]) + (s[2] + s[3])) * a.inv_win;
        }
        float e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.raw_energy) {
            for (int i = 0; i < a.win; ++i) {
                const float c = x[i] - mean;
                e[i & 3] = fmaf(c, c, e[i & 3]);
            }
        } else {
            float prev = x[0] - mean;   // replicate padding on the left: frame[0] - preemph * frame[0]
            for (int i = 0; i < a.win; ++i) {
                const float c = x[i] - mean;
                const float y = (c - a.preemph * prev) * a.window[i];
                e[i & 3] = fmaf(y, y, e[i & 3]);
                prev = c;
            }
        }
        float le = logf(fmaxf((e[0] + e[1]) + (e[2] + e[3]), eps));
        if (a.has_floor) le = fmaxf(le, a.log_floor);
        orow[(int64_t)t * ld + a.energy_col] = le;
        part += le;
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float mean_e = (a.cmn && T > 0) ? red[0] / (float)T : 0.0f;
    int mask_len = T;
    if (a.lens_ratio != nullptr) mask_len = (int)rintf(a.lens_ratio[b] * (float)T);  // round half to even
    for (int t = tid; t < Tout; t += 256) {   // (a thread re-reads what it wrote itself)
        float* p = orow + (int64_t)t * ld + a.energy_col;
        *p = t < T && t < mask_len ? *p - mean_e : 0.0f;
    }
    for (int i = tid; i < Tout * a.nbins; i += 256) {
        const int t = i / a.nbins, m = i - t * a.nbins;
        orow[(int64_t)t * ld + a.mel_col + m] = mrow[i];
    }
}

// snip_edges = False (torchaudio.compliance.kaldi._get_strided): T = (n + shift / 2) / shift frames over the signal mirrored at both ends --
// frame f covers the samples f * shift - pad ... + win - 1 with pad = win / 2 - shift / 2, index -1 - j reads x[j], index n + j reads x[n - 1 - j].
// This pre-pass writes the mirrored row [0, (T - 1) * shift + win) to the caller workspace (one extra pass over the waveform for a non-default
// argument); the frames of that row are then plain snip_edges = True frames, so fbank_kernel / fbank_tile_kernel run unchanged on it.
// lens_out (variable-length batches): the length of row b's mirrored signal, 0 when it has no frames.
__global__ __launch_bounds__(256) void fbank_mirror_kernel(const float* wav, int64_t wav_stride, const int64_t* num_samples, int64_t L, float* dst,
                                                           int64_t dst_stride, int64_t Lp, int64_t* lens_out, int win, int shift, int64_t min_samples) {
    const int b = blockIdx.y;
    int64_t n = num_samples != nullptr ? num_samples[b] : L;
    n = n < 0 ? 0 : (n > L ? L : n);
    const int64_t pad = win / 2 - shift / 2;
    int64_t T = n < min_samples ? 0 : (n + shift / 2) / shift;
    // the mirror reaches one signal length to either side (torchaudio concatenates the reversed signal once): a row too short for its
    // frames (n < pad, or the last frame beyond 2 n) has no frames here -- the reference raises on such a row
    if (T > 0 && (pad > n || (T - 1) * shift - pad + win > 2 * n)) T = 0;
    const int64_t own = T > 0 ? (T - 1) * shift + win : 0;
    if (lens_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) lens_out[b] = own;
    const float* src = wav + (int64_t)b * wav_stride;
    float* d = dst + (int64_t)b * dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < (int64_t)(blockIdx.x + 1) * 1024 && i < dst_stride; i += 256) {
        float v = 0.0f;
        if (i < own) {
            int64_t j = i - pad;
            j = j < 0 ? -1 - j : (j >= n ? 2 * n - 1 - j : j);
            v = src[j];
        }
        d[i] = v;
    }
}

}  // namespace mv

// ------------------------------------------------------------------------------------------ host side

struct MvFbank {
    MvFbankCfg cfg;
    int win, shift, nbins;
    int padded = 512;          // kaldi's FFT size: the window rounded up to a power of two (<= 512)
    int64_t min_samples = 0;   // min_duration in samples
    float* d_window = nullptr;
    float* d_window_half = nullptr;
    float* d_tw256 = nullptr;
    float* d_tw512 = nullptr;
    float* d_melb = nullptr;
    mv::FbankTables tab;
    size_t smem_bytes = 0;
    int waves = 15;  // fbank_kernel: workgroup size in waves (15 waves x 5 quads = the 75 quads of a 3 s utterance); 12 / 8 when a long mel table leaves less LDS
    bool tile_kernel = false;  // the mel geometry matches an instantiation of fbank_tile_kernel (every other geometry: fbank_kernel)
};

namespace {

// Kaldi mel banks, triangles in mel space (oracle/frontend.py::kaldi_mel_banks = torchaudio.compliance.kaldi.get_mel_banks), fp32 like torchaudio.
// vtln_warp != 1: the filter edges pass through Kaldi's 3-piece linear VTLN warp (vtln_warp_freq) and the weights follow the warped branch's
// half-open comparisons.  Returns false for option values torchaudio asserts on.
bool kaldi_mel_banks(int num_bins, int padded, float sample_freq, float low_freq, float high_freq, float vtln_low, float vtln_high, float vtln_warp,
                     std::vector<std::vector<float>>* out) {
    const int num_fft_bins = padded / 2;
    const float nyquist = 0.5f * sample_freq;
    if (high_freq <= 0.0f) high_freq += nyquist;
    if (vtln_high < 0.0f) vtln_high += nyquist;
    const bool warp = vtln_warp != 1.0f;
    if (warp && !(low_freq < vtln_low && vtln_low < high_freq && 0.0f < vtln_high && vtln_high < high_freq && vtln_low < vtln_high)) return false;
    const float fft_bin_width = sample_freq / padded;
    const float mel_lo = 1127.0f * logf(1.0f + low_freq / 700.0f);
    const float mel_hi = 1127.0f * logf(1.0f + high_freq / 700.0f);
    const float delta = (mel_hi - mel_lo) / (num_bins + 1);
    // vtln_warp_mel_freq: mel -> Hz -> warped Hz -> mel
    const float l = vtln_low * fmaxf(1.0f, vtln_warp), h = vtln_high * fminf(1.0f, vtln_warp), scale = 1.0f / vtln_warp;
    if (warp && !(l > low_freq && h < high_freq)) return false;
    const float scale_left = (scale * l - low_freq) / (l - low_freq), scale_right = (high_freq - scale * h) / (high_freq - h);
    auto warp_mel = [&](float mel) {
        const float f = 700.0f * (expf(mel / 1127.0f) - 1.0f);
        float r;
        if (f < low_freq || f > high_freq) r = f;
        else if (f < l) r = low_freq + scale_left * (f - low_freq);
        else if (f < h) r = scale * f;
        else r = high_freq + scale_right * (f - high_freq);
        return 1127.0f * logf(1.0f + r / 700.0f);
    };
    out->assign(num_bins, std::vector<float>(num_fft_bins, 0.0f));
    for (int b = 0; b < num_bins; ++b) {
        float left = mel_lo + b * delta, center = mel_lo + (b + 1.0f) * delta, right = mel_lo + (b + 2.0f) * delta;
        if (warp) {
            left = warp_mel(left);
            center = warp_mel(center);
            right = warp_mel(right);
        }
        for (int k = 0; k < num_fft_bins; ++k) {
            const float mel = 1127.0f * logf(1.0f + (fft_bin_width * k) / 700.0f);
            const float up = (mel - left) / (center - left);
            const float down = (right - mel) / (right - center);
            float w;
            if (!warp) {
                w = fminf(up, down);
                w = w > 0.0f ? w : 0.0f;
            } else {   // warping can move the order of left, center, right anywhere
                w = (mel > left && mel <= center) ? up : ((mel > center && mel < right) ? down : 0.0f);
            }
            (*out)[b][k] = w;
        }
    }
    return true;
}

template <typename T>
int upload(const std::vector<T>& v, T** dptr) {
    MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(dptr), v.size() * sizeof(T)));
    MV_HIP_OK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MV_OK;
}

}  // namespace

template <int NG, bool V>
hipError_t fbank_set_smem_v(size_t bytes) {
    hipError_t e = MV_SET_MAX_SMEM((mv::fbank_kernel<NG, V, 8>), bytes);
    if (e == hipSuccess) e = MV_SET_MAX_SMEM((mv::fbank_kernel<NG, V, 12>), bytes);
    if (e == hipSuccess) e = MV_SET_MAX_SMEM((mv::fbank_kernel<NG, V, 15>), bytes);
    return e;
}

hipError_t fbank_set_smem(size_t bytes) {
    hipError_t e = fbank_set_smem_v<13, true>(bytes);
    if (e == hipSuccess) e = fbank_set_smem_v<13, false>(bytes);
    if (e == hipSuccess) e = fbank_set_smem_v<16, true>(bytes);
    if (e == hipSuccess) e = fbank_set_smem_v<16, false>(bytes);
    return e;
}

template <int NG, bool V>
void fbank_launch_v(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a, int waves) {
    if (waves == 15) {
        MV_LAUNCH((mv::fbank_kernel<NG, V, 15>), (B, 1, 1), (15 * 64, 1, 1), smem, st, a);
    } else if (waves == 12) {
        MV_LAUNCH((mv::fbank_kernel<NG, V, 12>), (B, 1, 1), (12 * 64, 1, 1), smem, st, a);
    } else {
        MV_LAUNCH((mv::fbank_kernel<NG, V, 8>), (B, 1, 1), (8 * 64, 1, 1), smem, st, a);
    }
}

void fbank_launch(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a, int waves, bool vec2) {
    const bool ng13 = a.win > 12 * 32 && a.win <= 13 * 32;
    if (ng13 && vec2) return fbank_launch_v<13, true>(B, smem, st, a, waves);
    if (ng13) return fbank_launch_v<13, false>(B, smem, st, a, waves);
    if (vec2) return fbank_launch_v<16, true>(B, smem, st, a, waves);
    return fbank_launch_v<16, false>(B, smem, st, a, waves);
}

// fbank_tile_kernel is instantiated for the mel geometry of the reference configurations: 80 bins at 16 kHz on a 512-point
// FFT = a pass of 16 filter groups (7 four-bin steps) and a pass of 4 groups, each split over 4 blocks (3 steps)
constexpr int FBT_G0 = 7, FBT_G1 = 3;

static bool fbank_tile_geometry_ok(const MvFbank* h) {
    const mv::FbankTables& t = h->tab;
    return t.passes == 2 && t.pass_steps[0] == 4 * FBT_G0 && t.pass_steps[1] == 4 * FBT_G1 && t.pass_split[0] == 1 &&
           (h->nbins & 3) == 0 && h->nbins <= 128 && h->win >= 4 && (h->win & 1) == 0 && h->cfg.use_power && h->cfg.use_log_fbank;  // log power spectra only
    // (windows of 385 .. 416 samples -- 25 ms at 16 kHz is 400 -- run the 13-group instantiation, every other even window the 16-group one, whose
    // LDS window table holds all 512 taps since round 5: in rounds 2 - 4 it held 448, and groups 14 / 15 read the twiddle table behind it -- the
    // wrong features of 20 / 24 ms windows.  tests/test_gpu_parity.py::test_gpu_fbank_arguments runs both instantiations against the oracle.)
}

// sample groups of the instantiation a window runs: its own count when that is instantiated (20 / 24 / 25 / 30 ms at 16 kHz), else 16
static int fbank_tile_ng(int win) {
    const int ng = (win + 31) / 32;
    return (ng == 10 || ng == 12 || ng == 13 || ng == 15) ? ng : 16;
}

static size_t fbank_tile_fixed_lds_bytes(int win) {
    return ((size_t)mv::FBT_WAVES * mv::FBT_SLOT_FLOATS + mv::fbt_win_floats(fbank_tile_ng(win)) + 512) * sizeof(float);
}

template <int NG, bool V>
static hipError_t fbank_tile_set_smem() {
    return MV_SET_MAX_SMEM((mv::fbank_tile_kernel<NG, V, FBT_G0, FBT_G1>), 160 * 1024);
}

template <int NG>
static void fbank_tile_launch_ng(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a, bool vec2) {
    if (vec2) {
        MV_LAUNCH((mv::fbank_tile_kernel<NG, true, FBT_G0, FBT_G1>), (B, 1, 1), (mv::FBT_WAVES * 64, 1, 1), smem, st, a);
    } else {
        MV_LAUNCH((mv::fbank_tile_kernel<NG, false, FBT_G0, FBT_G1>), (B, 1, 1), (mv::FBT_WAVES * 64, 1, 1), smem, st, a);
    }
}

static void fbank_tile_launch(int B, size_t smem, hipStream_t st, const mv::FbankArgs& a, bool vec2) {
    switch (fbank_tile_ng(a.win)) {
        case 10: return fbank_tile_launch_ng<10>(B, smem, st, a, vec2);
        case 12: return fbank_tile_launch_ng<12>(B, smem, st, a, vec2);
        case 13: return fbank_tile_launch_ng<13>(B, smem, st, a, vec2);
        case 15: return fbank_tile_launch_ng<15>(B, smem, st, a, vec2);
        default: return fbank_tile_launch_ng<16>(B, smem, st, a, vec2);
    }
}

template <int NG>
static hipError_t fbank_tile_set_smem_ng() {
    hipError_t e = fbank_tile_set_smem<NG, true>();
    return e == hipSuccess ? fbank_tile_set_smem<NG, false>() : e;
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
    cfg->window_type = MV_WINDOW_POVEY;
    cfg->blackman_coeff = 0.42f;
    cfg->snip_edges = 1;
    cfg->subtract_mean = 0;
    cfg->min_duration = 0.0f;
    cfg->min_samples = 0;
    cfg->kernel = MV_FBANK_KERNEL_AUTO;
    cfg->vtln_warp = 1.0f;
    cfg->vtln_low = 100.0f;
    cfg->vtln_high = -500.0f;
    cfg->use_energy = 0;
    cfg->raw_energy = 1;
    cfg->energy_floor = 1.0f;
    cfg->htk_compat = 0;
}

int mv_fbank_create(const MvFbankCfg* cfg, MvFbank** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_fbank_create: null argument");
    const int win = (int)(cfg->sample_frequency * cfg->frame_length_ms * 0.001f);
    const int shift = (int)(cfg->sample_frequency * cfg->frame_shift_ms * 0.001f);
    MV_REQUIRE(shift >= 1, "mv_fbank_create: frame shift must be at least one sample");
    MV_REQUIRE(cfg->num_mel_bins >= 4 && cfg->num_mel_bins <= 64 * mv::FB_MAX_PASSES,
               "mv_fbank_create: num_mel_bins must be in [4, 128] (torchaudio's get_mel_banks asserts num_bins > 3)");
    {   // torchaudio.compliance.kaldi.get_mel_banks asserts on the band the same way ("Bad values in options: low-freq ... and high-freq ... vs. nyquist ...")
        const float nyq = 0.5f * cfg->sample_frequency, hi = cfg->high_freq <= 0.0f ? cfg->high_freq + nyq : cfg->high_freq;
        MV_REQUIRE(cfg->low_freq >= 0.0f && cfg->low_freq < nyq && hi > 0.0f && hi <= nyq && cfg->low_freq < hi,
                   "mv_fbank_create: bad band (need 0 <= low_freq < nyquist, 0 < high_freq <= nyquist and low_freq < high_freq; high_freq <= 0 counts from nyquist)");
    }
    if (win < 2 || win > mv::FB_NFFT)
        return mv::fail(MV_ERR_UNSUPPORTED,
                        "mv_fbank_create: only frame lengths of 2 .. 512 samples (an FFT of up to 512 points, e.g. 25 ms at 16 kHz) are "
                        "implemented on gfx950");
    MV_REQUIRE(cfg->window_type >= MV_WINDOW_POVEY && cfg->window_type <= MV_WINDOW_BLACKMAN, "mv_fbank_create: unknown window_type");
    MV_REQUIRE(cfg->kernel >= MV_FBANK_KERNEL_AUTO && cfg->kernel <= MV_FBANK_KERNEL_TILE, "mv_fbank_create: unknown kernel selector");
    MV_REQUIRE(cfg->min_duration >= 0.0f, "mv_fbank_create: negative min_duration");
    MV_REQUIRE(cfg->preemphasis_coefficient >= 0.0f && cfg->preemphasis_coefficient <= 1.0f,
               "mv_fbank_create: preemphasis_coefficient must be in [0, 1] (torchaudio asserts the same)");
    MV_REQUIRE(cfg->vtln_warp > 0.0f, "mv_fbank_create: vtln_warp must be positive");
    MV_REQUIRE((cfg->use_energy == 0 || cfg->use_energy == 1) && (cfg->raw_energy == 0 || cfg->raw_energy == 1) && (cfg->htk_compat == 0 || cfg->htk_compat == 1) &&
                   cfg->energy_floor >= 0.0f, "mv_fbank_create: use_energy / raw_energy / htk_compat are 0 or 1, energy_floor is not negative");
    MvFbank* h = new MvFbank();
    h->cfg = *cfg;
    h->win = win;
    h->shift = shift;
    h->nbins = cfg->num_mel_bins;
    h->padded = 2;
    while (h->padded < win) h->padded *= 2;   // round_to_power_of_two=True (the only form implemented)
    MV_REQUIRE(cfg->min_samples >= 0, "mv_fbank_create: negative min_samples");
    // len < min_duration * sf  <=>  len < ceil(...); the caller's own double evaluation (min_samples) wins over the float32 field
    h->min_samples = cfg->min_samples > 0 ? cfg->min_samples : (int64_t)ceil((double)cfg->min_duration * (double)cfg->sample_frequency);

    const double pi = 3.14159265358979323846;
    std::vector<float> window(512, 0.0f), window_half(512, 0.0f), tw256(512), tw512(512);
    for (int i = 0; i < win; ++i) {
        // torchaudio.compliance.kaldi._feature_window_function (all symmetric, periodic=False)
        const double a = 2.0 * pi / (win - 1);
        double w = 1.0;                                                                       // rectangular
        if (cfg->window_type == MV_WINDOW_POVEY) w = pow(0.5 - 0.5 * cos(a * i), 0.85);       // hann ** 0.85
        if (cfg->window_type == MV_WINDOW_HANNING) w = 0.5 - 0.5 * cos(a * i);
        if (cfg->window_type == MV_WINDOW_HAMMING) w = 0.54 - 0.46 * cos(a * i);
        if (cfg->window_type == MV_WINDOW_BLACKMAN) w = cfg->blackman_coeff - 0.5 * cos(a * i) + (0.5 - cfg->blackman_coeff) * cos(2.0 * a * i);
        window[i] = (float)w;
        window_half[i] = 0.5f * window[i];  // exact: the halved spectrum squares to |X|^2 without a final scale
    }
    for (int m = 0; m < 256; ++m) {
        const int k1 = m >> 4, n2 = m & 15;  // stage-1 -> stage-2 twiddle W256^(n2*k1), laid out [k1][n2]
        tw256[2 * m] = (float)cos(2.0 * pi * ((n2 * k1) & 255) / 256.0);
        tw256[2 * m + 1] = (float)sin(2.0 * pi * ((n2 * k1) & 255) / 256.0);
        tw512[2 * m] = (float)cos(2.0 * pi * m / 512.0);
        tw512[2 * m + 1] = (float)sin(2.0 * pi * m / 512.0);
    }
    // Windows that round up to an FFT of P < 512 points (8 kHz: 25 ms = 200 samples, P = 256): the kernels still transform the frame zero-padded to
    // 512 points; bin k of the P-point transform of a zero-padded frame IS bin k * 512 / P of the 512-point one, so kaldi's filter weights (built
    // for P) are placed on those bins and the bins in between carry zero weight.
    std::vector<std::vector<float>> banks_p;
    if (!kaldi_mel_banks(h->nbins, h->padded, cfg->sample_frequency, cfg->low_freq, cfg->high_freq, cfg->vtln_low, cfg->vtln_high, cfg->vtln_warp, &banks_p)) {
        delete h;
        return mv::fail(MV_ERR_INVALID_ARGUMENT, "mv_fbank_create: bad VTLN options (need low_freq < vtln_low < vtln_high < high_freq, and the warped cut-offs inside the band)");
    }
    std::vector<std::vector<float>> banks(h->nbins, std::vector<float>(mv::FB_NFFT / 2, 0.0f));
    {
        const int stride = mv::FB_NFFT / h->padded;
        for (int m = 0; m < h->nbins; ++m)
            for (int k = 0; k < h->padded / 2; ++k) banks[m][k * stride] = banks_p[m][k];
    }
    // Mel stage tables (frontend_common.h::build_mel_plan): passes of 16 blocks x (4 frames x 4 adjacent filters), every block
    // walking only the bins its triangles cover, inside the 256 bins of a power row
    mv::FbankTables& tab = h->tab;
    mv::MelPlan plan;
    std::vector<float> melb;
    if (!mv::build_mel_plan(banks, 256, &plan, &melb)) {
        delete h;
        return mv::fail(MV_ERR_UNSUPPORTED, "mv_fbank_create: num_mel_bins needs more than two MFMA passes");
    }
    tab.passes = plan.passes;
    for (int p = 0; p < mv::FB_MAX_PASSES; ++p) {
        tab.pass_steps[p] = plan.pass_steps[p];
        tab.pass_split[p] = plan.pass_split[p];
        tab.pass_gbase[p] = plan.pass_gbase[p];
        for (int blk = 0; blk < 16; ++blk) tab.pass_start[p][blk] = plan.pass_start[p][blk];
    }
    tab.melb_elems = (int)melb.size();
    int rc;
    if ((rc = upload(window, &h->d_window)) || (rc = upload(window_half, &h->d_window_half)) || (rc = upload(tw256, &h->d_tw256)) ||
        (rc = upload(tw512, &h->d_tw512)) ||
        (rc = upload(melb, &h->d_melb))) {
        mv_fbank_destroy(h);
        return rc;
    }
    tab.window = h->d_window;
    tab.window_half = h->d_window_half;
    tab.tw256 = h->d_tw256;
    tab.tw512 = h->d_tw512;
    tab.melb = h->d_melb;
    auto lds_need = [&](int waves) { return ((size_t)waves * 4 * mv::FB_SLOT_FLOATS + 3 * 512 + melb.size()) * sizeof(float); };
    if (lds_need(h->waves) > 160 * 1024 && h->waves > 12) h->waves = 12;  // a long mel table leaves room for fewer frame slots
    if (lds_need(h->waves) > 160 * 1024) h->waves = 8;
    h->smem_bytes = lds_need(h->waves);
    if (h->smem_bytes > 160 * 1024) {
        mv_fbank_destroy(h);
        return mv::fail(MV_ERR_UNSUPPORTED, "mv_fbank_create: the mel table does not fit the LDS next to the frame slots");
    }
    if (fbank_set_smem(h->smem_bytes) != hipSuccess) {
        mv_fbank_destroy(h);
        return mv::fail(MV_ERR_HIP, "mv_fbank_create: cannot reserve dynamic LDS for fbank_kernel");
    }
    h->tile_kernel = fbank_tile_geometry_ok(h) && cfg->kernel != MV_FBANK_KERNEL_GENERIC;
    if (cfg->kernel == MV_FBANK_KERNEL_TILE && !h->tile_kernel) {
        mv_fbank_destroy(h);
        return mv::fail(MV_ERR_UNSUPPORTED, "mv_fbank_create: fbank_tile_kernel is instantiated for log power spectra on the mel geometry of 80 bins / 16 kHz / "
                                            "512-point FFT (mel passes of 28 + 12 bins) with an even window; this configuration runs fbank_kernel");
    }
    if (h->tile_kernel && (fbank_tile_set_smem_ng<10>() != hipSuccess || fbank_tile_set_smem_ng<12>() != hipSuccess || fbank_tile_set_smem_ng<13>() != hipSuccess ||
                           fbank_tile_set_smem_ng<15>() != hipSuccess || fbank_tile_set_smem_ng<16>() != hipSuccess)) {
        mv_fbank_destroy(h);
        return mv::fail(MV_ERR_HIP, "mv_fbank_create: cannot reserve dynamic LDS for fbank_tile_kernel");
    }
    *out = h;
    return MV_OK;
}

int mv_fbank_destroy(MvFbank* h) {
    if (h == nullptr) return MV_OK;
    hipFree(h->d_window);
    hipFree(h->d_window_half);
    hipFree(h->d_tw256);
    hipFree(h->d_tw512);
    hipFree(h->d_melb);
    delete h;
    return MV_OK;
}

int mv_fbank_num_frames(const MvFbank* h, int64_t num_samples, int64_t* num_frames) {
    MV_REQUIRE(h != nullptr && num_frames != nullptr, "mv_fbank_num_frames: null argument");
    if (num_samples < h->min_samples) {
        *num_frames = 0;
    } else if (h->cfg.snip_edges) {
        *num_frames = num_samples < h->win ? 0 : 1 + (num_samples - h->win) / h->shift;
    } else {
        *num_frames = (num_samples + h->shift / 2) / h->shift;
    }
    return MV_OK;
}

int mv_fbank_info(const MvFbank* h, int32_t* tile_kernel, int32_t* pass_steps) {
    MV_REQUIRE(h != nullptr && tile_kernel != nullptr && pass_steps != nullptr, "mv_fbank_info: null argument");
    *tile_kernel = h->tile_kernel ? 1 : 0;
    pass_steps[0] = h->tab.pass_steps[0];
    pass_steps[1] = h->tab.pass_steps[1];
    return MV_OK;
}

// Geometry of one forward.  `fit` = feature rows that fit next to the wave slots in LDS (a multiple of 4).  The several-workgroups form
// (chunk_form) is a matter of time only -- both forms sum an utterance's time mean in one order (FbankArgs) -- and pays whenever there are
// fewer utterances than CUs: chunks of a multiple of 8 quads so that (utterance, chunk) workgroups about fill the chip.
struct FbankPlan {
    int64_t T = 0, fit = 0, need = 0;
    int nch = 1, chunk_quads = 0;
    bool chunk_form = false;   // B * nch workgroups + the finish pass; needs the caller's workspace
    // snip_edges = 0: the mirrored rows [B][mirror_stride] and their lengths [B] come first in the workspace (mirror_bytes, a multiple of 256)
    int64_t mirror_len = 0, mirror_stride = 0;
    size_t mirror_bytes = 0;
    size_t chunk_bytes = 0;
    // use_energy: the mel kernels' [B, T, nbins] rows come last (mel_off, a multiple of 256)
    size_t mel_off = 0, mel_bytes = 0;
    size_t workspace_bytes = 0;
};

static FbankPlan fbank_plan_mel(const MvFbank* h, int32_t B, int64_t L);

static FbankPlan fbank_plan(const MvFbank* h, int32_t B, int64_t L) {
    FbankPlan p = fbank_plan_mel(h, B, L);
    if (h->cfg.use_energy && B > 0 && p.T > 0) {
        p.mel_off = (size_t)mv::round_up((int64_t)p.workspace_bytes, (int64_t)256);
        p.mel_bytes = (size_t)B * (size_t)p.T * (size_t)h->nbins * sizeof(float);
        p.workspace_bytes = p.mel_off + p.mel_bytes;
    }
    return p;
}

static FbankPlan fbank_plan_mel(const MvFbank* h, int32_t B, int64_t L) {
    FbankPlan p;
    mv_fbank_num_frames(h, L, &p.T);
    if (B <= 0 || p.T <= 0) return p;
    if (!h->cfg.snip_edges) {
        p.mirror_len = (p.T - 1) * h->shift + h->win;
        p.mirror_stride = mv::round_up(p.mirror_len, (int64_t)4);
        p.mirror_bytes = (size_t)mv::round_up((int64_t)B * p.mirror_stride * (int64_t)sizeof(float) + (int64_t)B * (int64_t)sizeof(int64_t), (int64_t)256);
        p.workspace_bytes = p.mirror_bytes;
    }
    if (!h->tile_kernel) return p;
    p.fit = (int64_t)((160 * 1024 - fbank_tile_fixed_lds_bytes(h->win)) / ((size_t)h->nbins * sizeof(float))) & ~(int64_t)3;
    p.need = (p.T + 3) & ~(int64_t)3;
    const int cus = mv::device_cu_count();
    const int nquads = (int)(p.need / 4);
    if (B < cus && nquads >= 2 * mv::FBT_WAVES) {
        const int want = (int)mv::ceil_div(cus, B);                                    // chunks per utterance that fill the chip once
        const int most = nquads / mv::FBT_WAVES;                                       // (every wave of a chunk gets a quad)
        const int nch0 = want < most ? want : most;
        p.chunk_quads = (int)mv::round_up(mv::ceil_div(nquads, nch0), mv::FBT_WAVES);
        p.nch = (int)mv::ceil_div(nquads, p.chunk_quads);
        p.chunk_form = p.nch > 1;
    }
    if (p.chunk_form) p.chunk_bytes = (size_t)B * p.nch * p.chunk_quads * 128 * sizeof(float);
    p.workspace_bytes += p.chunk_bytes;
    return p;
}

static int fbank_forward_impl(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const float* lens_ratio,
                              const int64_t* num_samples, float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream);

int mv_fbank_forward(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                     const float* lens_ratio, float* out, mv_stream_t stream) {
    return fbank_forward_impl(h, wav, B, L, wav_stride, lens_ratio, nullptr, out, nullptr, 0, stream);
}

int mv_fbank_workspace_bytes(const MvFbank* h, int32_t B, int64_t L, size_t* bytes) {
    MV_REQUIRE(h != nullptr && bytes != nullptr, "mv_fbank_workspace_bytes: null argument");
    *bytes = fbank_plan(h, B, L).workspace_bytes;
    return MV_OK;
}

int mv_fbank_forward_ws(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const float* lens_ratio,
                        float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    return fbank_forward_impl(h, wav, B, L, wav_stride, lens_ratio, nullptr, out, workspace, workspace_bytes, stream);
}

int mv_fbank_forward_varlen(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                            const int64_t* num_samples, float* out, mv_stream_t stream) {
    MV_REQUIRE(num_samples != nullptr, "mv_fbank_forward_varlen: null length array");
    return fbank_forward_impl(h, wav, B, L, wav_stride, nullptr, num_samples, out, nullptr, 0, stream);
}

int mv_fbank_forward_varlen_ws(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const int64_t* num_samples,
                               float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    MV_REQUIRE(num_samples != nullptr, "mv_fbank_forward_varlen_ws: null length array");
    return fbank_forward_impl(h, wav, B, L, wav_stride, nullptr, num_samples, out, workspace, workspace_bytes, stream);
}

static int fbank_forward_impl(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const float* lens_ratio,
                              const int64_t* num_samples, float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    MV_REQUIRE(h != nullptr, "mv_fbank_forward: null handle");
    MV_REQUIRE(B >= 0 && L >= 0 && wav_stride >= L, "mv_fbank_forward: bad batch geometry");
    const FbankPlan plan = fbank_plan(h, B, L);
    int64_t T = 0;
    mv_fbank_num_frames(h, L, &T);
    if (B == 0 || T == 0) return MV_OK;  // empty output, like kaldi.fbank on a too-short input
    MV_REQUIRE(wav != nullptr && out != nullptr, "mv_fbank_forward: null buffer");
    MV_REQUIRE(T * (h->nbins + 1) < (int64_t)1 << 31, "mv_fbank_forward: utterance too long for 32-bit row indexing");
    float* const final_out = out;
    if (h->cfg.use_energy) {   // the mel kernels write to the workspace, fbank_energy_kernel composes the [B, T, nbins + 1] output
        if (workspace == nullptr || workspace_bytes < plan.workspace_bytes || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
            return mv::fail(MV_ERR_WORKSPACE, "mv_fbank_forward: use_energy writes the mel columns to the caller workspace first "
                                            "(mv_fbank_workspace_bytes, 16-byte aligned; mv_fbank_forward_ws / mv_fbank_forward_varlen_ws)");
        out = reinterpret_cast<float*>(static_cast<char*>(workspace) + plan.mel_off);
        workspace_bytes = plan.mel_off;   // what the other sections may use
    }
    mv::FbankArgs a;
    a.wav = wav;
    a.wav_stride = wav_stride;
    a.lens_ratio = lens_ratio;
    a.num_samples = num_samples;
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
    // kaldi.fbank's own subtract_mean (column means over the utterance's frames) followed by the wrapper's time mean over the same frames
    // subtracts (a rounding of) zero the second time: one subtraction serves both
    a.cmn = h->cfg.subtract_time_mean || h->cfg.subtract_mean;
    a.L = L;
    a.min_len = h->min_samples > h->win ? h->min_samples : h->win;
    a.tile_rows = 0;
    a.chunked = 0;
    a.nchunks = 1;
    a.chunk_quads = 0;
    a.part = nullptr;
    a.tab = h->tab;
    void* chunk_ws = workspace;
    size_t chunk_ws_bytes = workspace_bytes;
    if (!h->cfg.snip_edges) {
        if (workspace == nullptr || workspace_bytes < plan.mirror_bytes || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
            return mv::fail(MV_ERR_WORKSPACE, "mv_fbank_forward: snip_edges=False writes the mirrored signal to the caller workspace "
                                            "(mv_fbank_workspace_bytes, 16-byte aligned; mv_fbank_forward_ws / mv_fbank_forward_varlen_ws)");
        if (num_samples == nullptr) {   // (torchaudio concatenates the reversed signal once to either side: shorter signals make it raise)
            const int64_t pad = h->win / 2 - h->shift / 2;
            if (pad > L || (T - 1) * h->shift - pad + h->win > 2 * L)
                return mv::fail(MV_ERR_INVALID_ARGUMENT, "mv_fbank_forward: snip_edges=False: the signal is too short to be mirrored over its frames");
        }
        float* mw = static_cast<float*>(workspace);
        int64_t* mlens = reinterpret_cast<int64_t*>(mw + (int64_t)B * plan.mirror_stride);
        const int64_t xb = mv::ceil_div(plan.mirror_stride, (int64_t)1024);
        MV_REQUIRE(B <= 65535, "mv_fbank_forward: snip_edges=False takes at most 65535 rows per call");
        MV_LAUNCH(mv::fbank_mirror_kernel, ((unsigned)xb, (unsigned)B, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), wav, wav_stride, num_samples, L, mw,
                  plan.mirror_stride, plan.mirror_len, num_samples != nullptr ? mlens : nullptr, h->win, h->shift, h->min_samples);
        a.wav = wav = mw;
        a.wav_stride = wav_stride = plan.mirror_stride;
        a.L = plan.mirror_len;
        a.min_len = h->win;
        if (num_samples != nullptr) a.num_samples = num_samples = mlens;
        chunk_ws = static_cast<char*>(workspace) + plan.mirror_bytes;
        chunk_ws_bytes = workspace_bytes - plan.mirror_bytes;
    }
    const bool vec2 = (reinterpret_cast<uintptr_t>(wav) & 7) == 0 && (wav_stride & 1) == 0 && (h->shift & 1) == 0 && (h->win & 1) == 0;
    const int prof = mv::prof_begin(MV_PROF_FBANK, (double)B * (4.0 * (double)L + 4.0 * (double)T * h->nbins), static_cast<hipStream_t>(stream));
    if (h->tile_kernel) {
        const size_t fixed = fbank_tile_fixed_lds_bytes(h->win);
        // The chunk form and the one-workgroup form give the same bits (FbankArgs), so taking it is a matter of time only.  It needs scratch for
        // the chunks' per-wave sums, which belongs to the CALL, not to the handle: without the caller's workspace (mv_fbank_forward, the
        // variable-length entry point) every utterance runs on one workgroup.
        if (plan.chunk_form && num_samples == nullptr && chunk_ws != nullptr && chunk_ws_bytes >= plan.chunk_bytes &&
            (reinterpret_cast<uintptr_t>(chunk_ws) & 15) == 0) {
            a.chunked = 1;
            a.nchunks = plan.nch;
            a.chunk_quads = plan.chunk_quads;
            a.part = static_cast<float*>(chunk_ws);
            fbank_tile_launch(B * plan.nch, fixed, static_cast<hipStream_t>(stream), a, vec2);
            if (a.cmn || lens_ratio != nullptr) {
                const int row_blocks = (int)mv::ceil_div(T, mv::FBF_ROWS);
                MV_LAUNCH(mv::fbank_cmn_finish_kernel, (B * row_blocks, 1, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), out, a.part, lens_ratio,
                          (int)T, h->nbins, plan.nch * plan.chunk_quads, row_blocks, a.cmn);
            }
        } else {
            // the first `fit` feature rows stay in LDS next to the wave slots until the time mean is known (292 of the 298 frames of a 3 s
            // utterance at 80 bins); the rest take the write / re-read / rewrite route through global memory
            a.tile_rows = (int)(plan.fit < plan.need ? plan.fit : plan.need);
            fbank_tile_launch(B, fixed + (size_t)a.tile_rows * h->nbins * sizeof(float), static_cast<hipStream_t>(stream), a, vec2);
        }
    } else {
        fbank_launch(B, h->smem_bytes, static_cast<hipStream_t>(stream), a, h->waves, vec2);
    }
    if (h->cfg.use_energy) {
        int rc = mv::check_launch(h->tile_kernel ? "fbank_tile_kernel" : "fbank_kernel");
        if (rc != MV_OK) return rc;
        mv::FbankEnergyArgs e;
        e.wav = a.wav;
        e.wav_stride = a.wav_stride;
        e.lens_ratio = lens_ratio;
        e.num_samples = a.num_samples;
        e.mel = out;
        e.out = final_out;
        e.window = h->tab.window;
        e.T = (int)T;
        e.win = h->win;
        e.shift = h->shift;
        e.nbins = h->nbins;
        e.preemph = a.preemph;
        e.inv_win = a.inv_win;
        e.remove_dc = a.remove_dc;
        e.raw_energy = h->cfg.raw_energy;
        e.cmn = a.cmn;
        e.energy_col = h->cfg.htk_compat ? h->nbins : 0;
        e.mel_col = h->cfg.htk_compat ? 0 : 1;
        e.has_floor = h->cfg.energy_floor != 0.0f;
        e.log_floor = e.has_floor ? (float)log((double)h->cfg.energy_floor) : 0.0f;
        e.min_len = a.min_len;
        MV_LAUNCH(mv::fbank_energy_kernel, ((unsigned)B, 1, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), e);
        mv::prof_end(prof, static_cast<hipStream_t>(stream));
        return mv::check_launch("fbank_energy_kernel");
    }
    mv::prof_end(prof, static_cast<hipStream_t>(stream));
    return mv::check_launch(h->tile_kernel ? "fbank_tile_kernel" : "fbank_kernel");
}

}  // extern "C"
