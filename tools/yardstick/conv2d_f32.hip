// NOT PART OF THE PRODUCT LIBRARY.  The fp32 form of the ERes2Net conv layers (rounds 2-3): kept as the exact-fp32
// yard-stick the split-fp16 form (csrc/conv2ds.hip) is timed and checked against -- tools/bench_conv2d.py, tools/yardstick/check_conv2d.py.
// Built by tools/yardstick/build.py into tools/probe/libmvector_yardstick.so (this file + the product's objects).
//
// 2-D convolutions of the ERes2Net family (mvector/models/eres2net.py): 1x1 and 3x3 kernels, stride 1 or 2, zero padding
// k/2, no conv bias, BatchNorm directly behind every conv (folded into the packed weights + a per-channel bias), the
// clamped ReLU (Hardtanh 0..20, eres2net.py:12-15), the Res2Net "sp + spx[i]" input sum (eres2net.py:92), the residual
// add (eres2net.py:103-105) and the attentional feature fusion AFF (eres2net.py:32-52) as epilogue / loader modes.
//
// Precision: fp32 maps, fp32 weights, v_mfma_f32_16x16x4_f32.  Unlike the TDNN-style backbones this family does not
// tolerate 11-bit operands: ~50 clamped layers amplify a relative perturbation of 1e-6 at the input to 2e-4 at the
// embedding, and rounding either the weights or the activations to fp16 moves the embedding by 6-8 % (1 - cos 2e-3..5e-3
// against the 1e-4 bar; measured with the oracle).  The fp32 matrix pipe runs at 1/16 of the fp16 rate
// (157 vs 2500 TFLOP/s, MI355X_MICROARCH.md), so a hi + lo fp16 split of both operands (3 MFMA passes) would cut the matrix
// time 5.3 x -- but the matrix pipe is not what these layers wait for: with only ONE of every four fp32 MFMAs issued (wrong
// results, same loads / staging / stores: tools/probe_conv2d.py, profiles/r07e) the 54.9 M ERes2NetV2 runs 99.5 -> 77.1 ms
// (1.29 x) and the m32 model 36.9 -> 32.4 ms (1.14 x).  That is the upper bound of the split before its own cost (three VALU
// operations per activation element to form hi / lo); the family is bound by operand delivery -- fp32 activation traffic on the
// large maps, weight fetch latency on the small ones -- and stays on the fp32 pipe.
//
// Layout: feature maps are channel-last fp32 [B, H = frequency, W = time, C]; the channel counts of the model (13 ... 512)
// are padded to multiples of 16 when the weights are packed (zero rows / columns), so padded channels carry exact zeros.
//
// The output plane of an utterance is cut into segments of 16 consecutive time steps (row-major: the maps shrink to
// 10 x 38 in the last stage, so whole-row tiles would idle most lanes); one workgroup (4 waves) owns 8 consecutive
// segments and one tile of NB*16 output channels.  K loop over chunks of 16 input channels: the k x (15*stride + k)
// input patch behind each segment is staged in LDS once (zero padding, the input sum and the channel concatenation of AFF happen here) and every tap
// reads it back as MFMA B operands; the weights are small (<= 4.7 MB for the largest layer, L2 resident) and are read
// straight from global memory as A operands, each feeding the wave's two 16-step column blocks.
// K slot (step k4, lane group q) of a 16-channel group carries channel 4q + k4, so a lane's four K steps are ONE 16-byte
// read of consecutive channels on both sides (the sum over K does not care about the order).  D[channel 4q+r][step j]:
// a lane stores 4 consecutive channels (16 bytes) of one time step.
#include "kernels.h"
#include "conv2d_f32.h"

namespace mv {

typedef MvConv2dDesc Conv2dDesc;

constexpr int C2_SEGS = 8;      // 16-step segments per workgroup (two per wave)
constexpr int C2_RS = 16 + 4;   // LDS row stride in floats of the 3x3 kernel's patch (16 channels + 16 bytes: spreads the banks)


struct Conv2dArgs {
    const float* x;     // [B, H, W, ldx]
    const float* x2;    // optional second input, same spatial shape
    const float* w;     // packed [cout16][k*k][cin16], BatchNorm scale folded in
    const float* bias;  // [cout16]
    const float* res;   // epi 0: optional residual [B, Ho, Wo, ldres]; epi 2: first AFF operand
    const float* res2;  // epi 2: second AFF operand
    float* y;           // [B, Ho, Wo, ldy]
    int64_t ldx, ldx2, ldres, ldres2, ldy;
    int x2_mode;        // 0 none, 1 added to x, 2 concatenated behind the cin1 channels of x
    int cin1, cin16, cout16;
    int B, H, W, Ho, Wo, ks, stride, stride_w;  // stride = rows (H), stride_w = columns (W): the CAM++ head strides the frequency axis only
    int epi;            // 0: clamp(v [+ res], lo, hi); 1: SiLU; 2: AFF mix  res*(1+tanh v) + res2*(1-tanh v)
    float lo, hi;
};

__device__ __forceinline__ float4v mfma4(float a, float b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bias + epilogue mode + 16-byte stores of a wave's two segments (D[channel 4q+r][step j])
template <int NB>
__device__ __forceinline__ void conv2d_epilogue(const Conv2dArgs& a, float4v (&acc)[2][NB], const int (&ho_u)[2], const int (&wo_u)[2],
                                                int b, int co0, int j16, int q) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int wo = wo_u[u] + j16;
        if (ho_u[u] < 0 || wo >= a.Wo) continue;
        const int64_t pix = ((int64_t)b * a.Ho + ho_u[u]) * a.Wo + wo;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int co = co0 + m * 16 + q * 4;
            if (co0 + m * 16 >= a.cout16) break;
            const float4v bias = *reinterpret_cast<const float4v*>(a.bias + co);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[u][m][r] + bias[r];
            if (a.epi == 0) {
                if (a.res != nullptr) {
                    const float4v rv = *reinterpret_cast<const float4v*>(a.res + pix * a.ldres + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], a.lo), a.hi);
            } else if (a.epi == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-v[r]));
            } else {
                const float4v xa = *reinterpret_cast<const float4v*>(a.res + pix * a.ldres + co);
                const float4v ya = *reinterpret_cast<const float4v*>(a.res2 + pix * a.ldres2 + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = tanhf(v[r]);  // x_att = 1 + t;  out = x * x_att + y * (2 - x_att)
                    v[r] = xa[r] * (1.0f + t) + ya[r] * (1.0f - t);
                }
            }
            *reinterpret_cast<float4v*>(a.y + pix * a.ldy + co) = float4v{v[0], v[1], v[2], v[3]};
        }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void conv2d_kernel(Conv2dArgs a) {
    MV_DYN_SMEM(smem);
    float* patch = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, q = lane >> 4;
    const int nsegw = (a.Wo + 15) >> 4, nseg = a.Ho * nsegw;  // 16-step segments of one utterance, row-major
    const int stiles = (nseg + C2_SEGS - 1) / C2_SEGS;
    const int st = blockIdx.x % stiles, ct = blockIdx.x / stiles;
    const int b = blockIdx.y;
    const int sg0 = st * C2_SEGS, co0 = ct * NB * 16;
    constexpr int taps = 9;
    const int s = a.stride_w, sh = a.stride;   // column / row stride
    const int ncols = 15 * s + 3;              // input columns behind one segment
    const int seg_floats = 3 * ncols * C2_RS;
    const int per_seg = 3 * ncols * 4;          // 16-byte pieces of one segment's patch

    float4v acc[2][NB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[u][m] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    // this lane's weight rows (clamped inside the matrix for the uneven last channel tile: computed, never stored)
    const float* wrow[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int cb = co0 + m * 16 < a.cout16 ? co0 + m * 16 : a.cout16 - 16;
        wrow[m] = a.w + (int64_t)(cb + j16) * taps * a.cin16 + q * 4;
    }
    // this wave's two segments
    int ho_u[2], wo_u[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sg = sg0 + wave * 2 + u;
        ho_u[u] = sg < nseg ? sg / nsegw : -1;
        wo_u[u] = sg < nseg ? (sg - ho_u[u] * nsegw) * 16 : 0;
    }

    // staging role of this thread: lanes 32*slot .. 32*slot+31 fill the patch of segment sg0 + slot
    const int st_slot = tid >> 5, st_lane = tid & 31;
    const int st_sg = sg0 + st_slot;
    const bool st_ok = st_sg < nseg;
    const int st_ho = st_ok ? st_sg / nsegw : 0;
    const int st_h0 = st_ho * sh - 1;                             // input row of kh = 0
    const int st_w0 = (st_sg - st_ho * nsegw) * 16 * s - 1;       // input column of col = 0
    const int ncols_magic = 65536 / ncols + 1;                    // rc / ncols for rc < 99

    // 16 input channels per K chunk (see conv2d_lds_bytes): every tap is then one MFMA group of 4 K steps.  The A operands of tap
    // t+1 are requested before the MFMAs of tap t (those of tap 0 before the patch is staged), so the L2 latency of the
    // weight reads is paid once per chunk instead of once per tap.
    for (int c0 = 0; c0 < a.cin16; c0 += 16) {
        float4v an[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) an[m] = *reinterpret_cast<const float4v*>(wrow[m] + c0);
        {
            float* pseg = patch + st_slot * seg_floats;
            for (int r0 = st_lane; r0 < per_seg; r0 += 32) {
                const int ch = r0 & 3;
                const int rc = r0 >> 2;                              // kh * ncols + col, < 99
                const int kh = (rc * ncols_magic) >> 16;
                const int col = rc - kh * ncols;
                const int hi = st_h0 + kh;
                const int wi = st_w0 + col;
                float4v v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (st_ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) {
                    const int64_t pix = ((int64_t)b * a.H + hi) * a.W + wi;
                    const int cc = c0 + ch * 4;
                    if (a.x2_mode == 2 && cc >= a.cin1) {
                        v = *reinterpret_cast<const float4v*>(a.x2 + pix * a.ldx2 + (cc - a.cin1));
                    } else {
                        v = *reinterpret_cast<const float4v*>(a.x + pix * a.ldx + cc);
                        if (a.x2_mode == 1) v += *reinterpret_cast<const float4v*>(a.x2 + pix * a.ldx2 + cc);
                    }
                }
                *reinterpret_cast<float4v*>(pseg + rc * C2_RS + ch * 4) = v;
            }
        }
        __syncthreads();
        const float* pw = patch + (wave * 2) * seg_floats + q * 4;
        auto tap_step = [&](int tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            float4v af[NB], bf[2];
#pragma unroll
            for (int m = 0; m < NB; ++m) af[m] = an[m];
            if (tap + 1 < 9) {
#pragma unroll
                for (int m = 0; m < NB; ++m) an[m] = *reinterpret_cast<const float4v*>(wrow[m] + (int64_t)(tap + 1) * a.cin16 + c0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                bf[u] = *reinterpret_cast<const float4v*>(pw + u * seg_floats + (kh * ncols + j16 * s + kw) * C2_RS);
#pragma unroll
            for (int m = 0; m < NB; ++m)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
        };
        if constexpr (NB <= 2) {  // narrow tiles: all nine taps unrolled (the compiler then requests every tap's weights up front)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) tap_step(tap);
        } else {                  // wide tiles: a rolled loop keeps exactly one tap of weights in flight (registers)
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) tap_step(tap);
        }
        __syncthreads();
    }

    conv2d_epilogue<NB>(a, acc, ho_u, wo_u, b, co0, j16, q);
}

// 1x1 convolutions (conv1 / conv3 / shortcut / AFF: half of the family's FLOPs and most of its bytes) have no tap reuse, so
// nothing is gained by staging the input: every wave streams its two segments' B operands straight from global memory
// (a time step's 16 channels = 64 contiguous bytes over the 4 lane groups) -- no LDS, no barriers, waves fully independent.
template <int NB>
__global__ __launch_bounds__(256) void conv2d_1x1_kernel(Conv2dArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, q = lane >> 4;
    const int nsegw = (a.Wo + 15) >> 4, nseg = a.Ho * nsegw;
    const int stiles = (nseg + C2_SEGS - 1) / C2_SEGS;
    const int st = blockIdx.x % stiles, ct = blockIdx.x / stiles;
    const int b = blockIdx.y;
    const int sg0 = st * C2_SEGS, co0 = ct * NB * 16;
    const int s = a.stride_w, sh = a.stride;

    float4v acc[2][NB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[u][m] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    const float* wrow[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int cb = co0 + m * 16 < a.cout16 ? co0 + m * 16 : a.cout16 - 16;
        wrow[m] = a.w + (int64_t)(cb + j16) * a.cin16 + q * 4;
    }
    int ho_u[2], wo_u[2];
    const float* xp[2];   // this lane's input position in x (and x2): null = outside the map
    const float* x2p[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sg = sg0 + wave * 2 + u;
        ho_u[u] = sg < nseg ? sg / nsegw : -1;
        wo_u[u] = sg < nseg ? (sg - ho_u[u] * nsegw) * 16 : 0;
        const int wo = wo_u[u] + j16;
        const bool ok = ho_u[u] >= 0 && wo < a.Wo;
        const int64_t pix = ok ? ((int64_t)b * a.H + ho_u[u] * sh) * a.W + wo * s : 0;
        xp[u] = ok ? a.x + pix * a.ldx + q * 4 : nullptr;
        x2p[u] = ok && a.x2_mode != 0 ? a.x2 + pix * a.ldx2 + q * 4 : nullptr;
    }
    const int groups = a.cin16 >> 4;
    const int g1 = a.x2_mode == 2 ? a.cin1 >> 4 : groups;  // groups taken from x; the rest from x2 (AFF concat)
    auto load_b = [&](int g, int u) {
        float4v v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (xp[u] != nullptr) {
            if (g < g1) {
                v = *reinterpret_cast<const float4v*>(xp[u] + g * 16);
                if (a.x2_mode == 1) v += *reinterpret_cast<const float4v*>(x2p[u] + g * 16);
            } else {
                v = *reinterpret_cast<const float4v*>(x2p[u] + (g - g1) * 16);
            }
        }
        return v;
    };
    // One 16-channel group per trip of a rolled loop (an unrolled one lets the compiler hoist every group's operands:
    // 244 VGPRs at NB = 8).  Operands run ahead of the MFMAs: B (HBM latency) two groups, A (L2) one group; A is requested
    // before B so that waiting for it never waits for the younger B request (in-order vmcnt).
    float4v an[NB], bq[2][2];
#pragma unroll
    for (int m = 0; m < NB; ++m) an[m] = *reinterpret_cast<const float4v*>(wrow[m]);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int u = 0; u < 2; ++u) bq[gg][u] = gg < groups ? load_b(gg, u) : float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int g = 0; g < groups; ++g) {
        float4v af[NB], bf[2];
#pragma unroll
        for (int m = 0; m < NB; ++m) af[m] = an[m];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf[u] = bq[0][u];
            bq[0][u] = bq[1][u];
        }
        if (g + 1 < groups) {
#pragma unroll
            for (int m = 0; m < NB; ++m) an[m] = *reinterpret_cast<const float4v*>(wrow[m] + (g + 1) * 16);
        }
        if (g + 2 < groups) {
#pragma unroll
            for (int u = 0; u < 2; ++u) bq[1][u] = load_b(g + 2, u);
        }
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u][m] = mfma4(af[m][k4], bf[u][k4], acc[u][m]);
    }
    conv2d_epilogue<NB>(a, acc, ho_u, wo_u, b, co0, j16, q);
}

// K chunk of the 3x3 kernel = 16 channels.  The patch of a 32-channel chunk (62 KiB at stride 1, 114 KiB at stride 2) leaves
// 2 / 1 workgroups per CU; with 16 it is 4 / 2, and the extra barriers cost less than the lost overlap: ERes2NetV2-m32
// 5.76 k -> 6.19 k utt/s, ERes2Net-m32 4.64 k -> 5.12 k (r02e, same box).
static size_t conv2d_lds_bytes(int stride) { return (size_t)C2_SEGS * 3 * (15 * stride + 3) * C2_RS * sizeof(float); }

int conv2d_launch(const Conv2dDesc& d, hipStream_t stream) {
    MV_REQUIRE(d.x != nullptr && d.w != nullptr && d.bias != nullptr && d.y != nullptr, "conv2d: null pointer");
    MV_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0, "conv2d: empty input");
    MV_REQUIRE((d.ks == 1 || d.ks == 3) && (d.stride == 1 || d.stride == 2), "conv2d: kernel 1 or 3, stride 1 or 2");
    MV_REQUIRE(d.stride_w == 0 || d.stride_w == 1 || d.stride_w == 2, "conv2d: stride_w must be 0 (= stride), 1 or 2");
    const int stride_w = d.stride_w == 0 ? d.stride : d.stride_w;
    MV_REQUIRE(d.cin16 > 0 && d.cin16 % 16 == 0 && d.cout16 > 0 && d.cout16 % 16 == 0, "conv2d: channels must be padded to 16");
    MV_REQUIRE(d.ldx % 4 == 0 && d.ldy % 4 == 0, "conv2d: leading dimensions");
    MV_REQUIRE(d.ldx > 0 && d.ldy > 0 && (d.x2_mode == 0 || d.ldx2 > 0) && (d.res == nullptr || d.ldres > 0) && (d.res2 == nullptr || d.ldres2 > 0),
               "conv2d: leading dimensions must be positive");
    MV_REQUIRE((d.res == nullptr || d.ldres % 4 == 0) && (d.res2 == nullptr || d.ldres2 % 4 == 0), "conv2d: operand leading dimensions (16-byte rows)");
    MV_REQUIRE(d.x2_mode >= 0 && d.x2_mode <= 2 && (d.x2_mode == 0 || (d.x2 != nullptr && d.ldx2 % 4 == 0)), "conv2d: second input");
    if (d.x2_mode == 2) MV_REQUIRE(d.cin1 > 0 && d.cin1 % 4 == 0 && d.cin1 < d.cin16, "conv2d: concat split");
    MV_REQUIRE(d.epi >= 0 && d.epi <= 2, "conv2d: epilogue mode");
    if (d.epi == 2) MV_REQUIRE(d.res != nullptr && d.res2 != nullptr, "conv2d: AFF mix needs both operands");
    Conv2dArgs a;
    a.x = d.x; a.x2 = d.x2; a.w = d.w; a.bias = d.bias; a.res = d.res; a.res2 = d.res2; a.y = d.y;
    a.ldx = d.ldx; a.ldx2 = d.ldx2; a.ldres = d.ldres; a.ldres2 = d.ldres2; a.ldy = d.ldy;
    a.x2_mode = d.x2_mode; a.cin1 = d.x2_mode == 2 ? d.cin1 : d.cin16; a.cin16 = d.cin16; a.cout16 = d.cout16;
    a.B = d.B; a.H = d.H; a.W = d.W; a.ks = d.ks; a.stride = d.stride; a.stride_w = stride_w;
    const int p = d.ks / 2;
    a.Ho = (d.H + 2 * p - d.ks) / d.stride + 1;
    a.Wo = (d.W + 2 * p - d.ks) / stride_w + 1;
    a.epi = d.epi; a.lo = d.lo; a.hi = d.hi;
    // channel tiles: one when the layer has <= 8 blocks of 16 channels, else the most even split into tiles of <= 8 blocks
    const int nblk = d.cout16 / 16;
    // 1x1 kernel: at most 4 blocks of 16 channels per workgroup -- 124 VGPRs = 4 waves per SIMD instead of 2 at 8 blocks, which
    // hides more of the operand latency than reading the B operands once more from L2 costs (ERes2NetV2-m32 6.56 k -> 6.86 k
    // utt/s; 2 blocks: 6.62 k).  The 3x3 kernel keeps up to 8 blocks.
    const int cap = d.ks == 1 ? 4 : 8;
    const int ctiles = (nblk + cap - 1) / cap;
    const int nb = (nblk + ctiles - 1) / ctiles;
    const int nsegw = (a.Wo + 15) / 16;
    const int stiles = (a.Ho * nsegw + C2_SEGS - 1) / C2_SEGS;
    const dim3 grid((unsigned)(stiles * ctiles), (unsigned)d.B, 1);
    MV_REQUIRE(d.B <= 65535, "conv2d: batch too large for one launch");
    const size_t lds = conv2d_lds_bytes(stride_w);
    static DeviceOnce smem_set;   // (per device: the attribute belongs to the current device's code object)
    int smem_set_slot;
    if (device_once_pending(smem_set, &smem_set_slot)) {
        const int big = (int)conv2d_lds_bytes(2);
        if (MV_SET_MAX_SMEM(conv2d_kernel<1>, big) != hipSuccess || MV_SET_MAX_SMEM(conv2d_kernel<2>, big) != hipSuccess ||
            MV_SET_MAX_SMEM(conv2d_kernel<3>, big) != hipSuccess || MV_SET_MAX_SMEM(conv2d_kernel<4>, big) != hipSuccess ||
            MV_SET_MAX_SMEM(conv2d_kernel<5>, big) != hipSuccess || MV_SET_MAX_SMEM(conv2d_kernel<6>, big) != hipSuccess ||
            MV_SET_MAX_SMEM(conv2d_kernel<7>, big) != hipSuccess || MV_SET_MAX_SMEM(conv2d_kernel<8>, big) != hipSuccess)
            return fail(MV_ERR_HIP, "conv2d: cannot reserve dynamic LDS");
        device_once_done(smem_set, smem_set_slot);
    }
    const int prof = prof_begin(MV_PROF_CONV2D, 2.0 * d.B * a.Ho * a.Wo * (double)(d.cin_alg > 0 ? d.cin_alg : d.cin16) *
                                                (d.cout_alg > 0 ? d.cout_alg : d.cout16) * d.ks * d.ks, stream);
    if (d.ks == 1) {
        switch (nb) {
            case 8: MV_LAUNCH(conv2d_1x1_kernel<8>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 7: MV_LAUNCH(conv2d_1x1_kernel<7>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 6: MV_LAUNCH(conv2d_1x1_kernel<6>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 5: MV_LAUNCH(conv2d_1x1_kernel<5>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 4: MV_LAUNCH(conv2d_1x1_kernel<4>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 3: MV_LAUNCH(conv2d_1x1_kernel<3>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            case 2: MV_LAUNCH(conv2d_1x1_kernel<2>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
            default: MV_LAUNCH(conv2d_1x1_kernel<1>, (grid.x, grid.y, 1), (256, 1, 1), 0, stream, a); break;
        }
    } else
    switch (nb) {
        case 8: MV_LAUNCH(conv2d_kernel<8>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 7: MV_LAUNCH(conv2d_kernel<7>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 6: MV_LAUNCH(conv2d_kernel<6>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 5: MV_LAUNCH(conv2d_kernel<5>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 4: MV_LAUNCH(conv2d_kernel<4>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 3: MV_LAUNCH(conv2d_kernel<3>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        case 2: MV_LAUNCH(conv2d_kernel<2>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
        default: MV_LAUNCH(conv2d_kernel<1>, (grid.x, grid.y, 1), (256, 1, 1), lds, stream, a); break;
    }
    prof_end(prof, stream);
    return check_launch("conv2d_kernel");
}

// [Cout][Cin][k][k] fp32 (* out_scale[co]) -> fp32 [cout16][k*k][cin16], zero padded
__global__ void pack_conv2d_weight_kernel(const float* w, const float* out_scale, int cout, int cin, int taps, int cout16,
                                          int cin16, float* packed) {
    const int64_t total = (int64_t)cout16 * taps * cin16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin16);
        const int tap = (int)((i / cin16) % taps);
        const int co = (int)(i / ((int64_t)cin16 * taps));
        float v = 0.0f;
        if (co < cout && ci < cin) v = w[((int64_t)co * cin + ci) * taps + tap] * (out_scale != nullptr ? out_scale[co] : 1.0f);
        packed[i] = v;
    }
}

}  // namespace mv

extern "C" {

int64_t mv_conv2d_packed_elems(int32_t cout, int32_t cin, int32_t ks) {
    return (int64_t)mv::round_up(cout, 16) * ks * ks * mv::round_up(cin, 16);
}

int mv_conv2d_pack_weight(const float* w, const float* out_scale, int32_t cout, int32_t cin, int32_t ks, float* packed,
                          mv_stream_t stream) {
    MV_REQUIRE(w != nullptr && packed != nullptr && cout > 0 && cin > 0 && (ks == 1 || ks == 3), "mv_conv2d_pack_weight: bad argument");
    const int64_t total = mv_conv2d_packed_elems(cout, cin, ks);
    const int grid = (int)(mv::ceil_div(total, 256) < 4096 ? mv::ceil_div(total, 256) : 4096);
    MV_LAUNCH(mv::pack_conv2d_weight_kernel, (grid, 1, 1), (256, 1, 1), 0, static_cast<hipStream_t>(stream), w, out_scale, cout, cin,
              ks * ks, (int)mv::round_up(cout, 16), (int)mv::round_up(cin, 16), packed);
    return mv::check_launch("pack_conv2d_weight_kernel");
}

int mv_conv2d_forward(const MvConv2dDesc* d, mv_stream_t stream) {
    MV_REQUIRE(d != nullptr, "mv_conv2d_forward: null descriptor");
    return mv::conv2d_launch(*d, static_cast<hipStream_t>(stream));
}

}  // extern "C"
