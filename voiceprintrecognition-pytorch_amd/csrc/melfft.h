// melspec_pow2_kernel (melfft.hip): MelSpectrogram for power-of-two n_fft <= 1024 in one launch; host side in melspec.hip
#pragma once
#include "frontend_common.h"

namespace mv {

// waves per workgroup: 8 = no prefetch, 232 registers, two per SIMD (measured against 4 = prefetch form, ~380 registers, one wave per SIMD, and
// 12 = 168-register cap, 36 spilled.  README geometry, 256 x 3 s (profiles/r09a_melspec_pow2_waves_ab.log): 70.1 / 107.6 / 87.1 us)
constexpr int MF_WAVES = 8;
constexpr int MF_ROW = 65;                          // complex elements per transpose row: [4 frames][16 lanes] + 1 pad
constexpr int MF_PSTR = 520;                        // floats between the power rows of the wave's four frames (513 bins + pad)
constexpr int MF_SLOT_FLOATS = 16 * MF_ROW * 2;     // 2080 floats per wave: 16 transpose rows = 4 power rows
static_assert(MF_SLOT_FLOATS == 4 * MF_PSTR, "the power rows reuse the transpose slot");

struct MelFftArgs {
    const float* wav;
    int64_t wav_stride;
    int64_t L;
    const float* lens_ratio;
    float* out;               // [B, T, n_mels]
    const float* window;      // [n_fft]
    const float* tw512;       // [32 k1][16 l][2]: cos, sin of 2 pi l k1 / 512
    const float* w1024;       // [512][2]: cos, sin of 2 pi k / 1024
    const float* melb;        // MFMA B-operand order (frontend_common.h), passes back to back
    int B, T, n_fft, hop, pad, n_mels, cmn, tile_rows;
    MelPlan plan;
};

size_t melfft_fixed_lds_bytes();
int melfft_launch(const MelFftArgs& a, size_t smem, hipStream_t stream);

}  // namespace mv
