// Internal launch functions shared between the layer-level C ABI and the model forwards.
#pragma once
#include "common.h"

namespace mv {

int conv1d_cin_pad(int cin);
int conv1d_cout_pad(int cout);
// finish pass of the time statistics a persistent-kernel epilogue can emit (MvConv1dDesc.stat_sum / stat_sq)
int conv_stats_finish_launch(const float* psum, const float* psq, const float* shift, int B, int T, int C, float* mean, float* stdv,
                             int64_t ld_out, float clamp_eps, hipStream_t stream);
int conv1d_launch(const MvConv1dDesc& d, hipStream_t stream);
// fused time statistics of a 1x1 layer's INPUT (MvConv1dDesc.in_stat_sum / in_stat_sq): partial buffer size and the finish pass
int64_t conv_in_stats_elems(int B, int T, int cin);
int conv_in_stats_finish_launch(const float* psum, const float* psq, int B, int T, int C, float* mean, float* stdv, int64_t ld_out,
                                float clamp_eps, hipStream_t stream);
// zh[b, t, :] <- tanh(relu(zh[b, t, :] + row_bias[b, :]) * scale + shift), fp16 in place: the deferred epilogue of the ASP hidden layer
int bn_relu_rows_launch(const half_t* x, int64_t ldx, const float* scale, const float* shift, half_t* y, int64_t ldy, int64_t n_rows, int C,
                        hipStream_t stream);
int asp_hidden_act_launch(half_t* zh, const float* row_bias, const float* scale, const float* shift, int B, int T, int A, hipStream_t stream);

int conv2d_first_launch(const float* feats, float* out, const float* w, const float* bias, int B, int T, int F, int C,
                        hipStream_t stream);
int tstp_launch(const float* x, int64_t ld, int B, int H, int W, int C, float* stats, hipStream_t stream);

// split-fp16 form of the same layers on S16 maps (conv2ds.hip)
int conv2ds_launch(const MvConv2dsDesc& d, hipStream_t stream);
int64_t conv2ds_packed_floats(int cout16, int cin16, int ks);
float conv2ds_pack_host(const float* w_dense, int cout16, int cin16, int ks, half_t* out);  // [cout16][taps][cin16] fp32 -> split; returns oscale
int map_split_launch(const float* x, void* y, int64_t n, hipStream_t stream);
int map_merge_launch(const void* x, float* y, int64_t n, hipStream_t stream);
// peak (optional device word): largest |64 * value| stored, as float bits (s16map.h)
int conv2d_first_s16_launch(const float* feats, half_t* out, const float* w, const float* bias, int B, int T, int F, int C, hipStream_t stream, unsigned* peak = nullptr);
int tstp_s16_launch(const half_t* x, int64_t ld, int B, int H, int W, int C, float* stats, hipStream_t stream);

// splitk_ws (optional, linear_f32_splitk_floats(B, K, O) floats; 0 = the direct kernel is what runs): long reductions over many rows as K slices + a
// slice-ordered sum (linear.hip)
int linear_f32_launch(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int act, float* y,
                      int64_t ldy, int B, int K, int O, int cosine, hipStream_t stream, float* splitk_ws = nullptr, size_t splitk_ws_floats = 0);
size_t linear_f32_splitk_floats(int B, int K, int O);

int time_stats_launch(const half_t* x, int64_t ld, int B, int T, int C, float* mean, float* stdv, int64_t ld_out,
                      int unbiased, float clamp_eps, hipStream_t stream, const float* in_scale = nullptr,
                      const float* in_shift = nullptr);
int fcm_conv1_launch(const float* feats, half_t* out, const float* w, const float* bias, int B, int T, int F,
                     hipStream_t stream);
int fcm_conv3x3_launch(const half_t* x, int Fin, int sf, const half_t* x2, int F2, int sf2, int mode2, const half_t* w,
                       const float* bias, half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int B, int T, int Fout,
                       hipStream_t stream);
// fp32 head of CAM++: maps fp32 [B, F8, T, 32] -> the TDNN's input rows fp16 [B, T, F8, 32] (saturating at +-65504)
int fcm_rows_from_f32_launch(const float* maps, half_t* rows, int B, int T, int F8, hipStream_t stream);
// the same from S16 maps (the split-fp16 head)
int fcm_rows_from_s16_launch(const half_t* maps, half_t* rows, int B, int T, int F8, hipStream_t stream, float scale = 1.0f);   // rows = maps * scale
// one BasicResBlock of the FCM head (campplus.py:221-254) as one launch, intermediate map kept in LDS (fcmblock.hip)
bool fcm_block_supported(const half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int T, int Fin);
// feats != nullptr (then x == nullptr, sf == 2): the block input is head.conv1 + bn1 + ReLU of the fp32 features [B, T, Fin], evaluated inside
// the kernel from the fcm_c1_pack fragments c1a and the folded shift c1b
int fcm_block_launch(const half_t* x, int Fin, int sf, const half_t* w1, const float* b1, const half_t* w2, const float* b2, int shortcut,
                     half_t* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int B, int T, hipStream_t stream, const float* feats = nullptr,
                     const half_t* c1a = nullptr, const float* c1b = nullptr);
void fcm_c1_pack(const float* w, half_t* out);  // [32][3 df][3 dt] fp32 (BN folded) -> [2][64][8] fp16 MFMA A fragments
// one CAMDenseTDNNLayer (campplus.py:114-150) as one launch, one workgroup per utterance (camdense.hip); T2 <= 160 frames
bool cam_dense_long_supported(int T2, int cin, int bottleneck, int growth, int dil, int seg_len);
int64_t cam_dense_long_part_floats(int B, int T2);
int cam_dense_long_launch(half_t* x, int64_t ldx, int B, int T2, int cin, const half_t* w1, const float* bn1_s, const float* bn1_t,
                          const float* bn2_s, const float* bn2_t, const half_t* wl, const float* wa, const float* ba, const float* wb,
                          const float* bb, int dil, int seg_len, half_t* hws, float* hpart, hipStream_t stream);
bool cam_dense_layer_supported(int T2, int cin, int bottleneck, int growth, int dil, int seg_len);
int cam_dense_layer_launch(half_t* x, int64_t ldx, int B, int T2, int cin, const half_t* w1, const float* bn1_s, const float* bn1_t,
                           const float* bn2_s, const float* bn2_t, const half_t* wl, const float* wa, const float* ba, const float* wb,
                           const float* bb, int dil, int seg_len, hipStream_t stream);
// all dense layers of one CAM++ block in one launch (camblock.hip): per-layer parameters as a device array
struct MvCamLayerDesc {
    const half_t* w1;                 // packed [128][1][cin_pad]
    const float *bn1_s, *bn1_t;       // [cin]
    const float *bn2_s, *bn2_t;       // [128]
    const half_t* wl;                 // packed [32][3][128]
    const float *wa, *ba, *wb, *bb;   // context FCs
    int cin, cin_pad;
};
bool cam_dense_block_supported(int T2, int c_in, int c_out, int bottleneck, int growth, int dil, int seg_len);
int cam_dense_block_launch(half_t* x, int64_t ldx, int B, int T2, const MvCamLayerDesc* layers_dev, int nlayers, int dil, int seg_len,
                           hipStream_t stream);
int seg_mean_launch(const half_t* x, int64_t ld, int B, int T, int C, int seg_len, float* ctx, hipStream_t stream);
int se_gate_residual_launch(const half_t* y, int64_t ldy, const float* gate, const half_t* res, int64_t ldr, half_t* out,
                            int64_t ldo, int B, int T, int C, hipStream_t stream);
int cast_reflect_pad_launch(const float* src, half_t* dst, int B, int T, int C, int pad, hipStream_t stream);
int cast_rows_f32_f16_launch(const float* src, int64_t lds_, half_t* dst, int64_t ldd, int64_t n_rows, int C, hipStream_t stream);
int copy_slice_launch(const half_t* src, int64_t lds_, half_t* dst, int64_t ldd, int C, int64_t n_rows, hipStream_t stream);
bool res2_chain_supported(int T, int width, int steps, int k, int dil);
int res2_chain_launch(const half_t* x, half_t* y, const half_t* const* w, const float* const* bias, const float* const* scale,
                      const float* const* shift, int B, int T, int C, int width, int steps, int k, int dil, hipStream_t stream);
int asp_pool_launch(const half_t* h, const half_t* w2_packed, const half_t* x, int64_t ldx, const float* gmean,
                    int64_t gmean_ld, float* out, int B, int T, int C, int A, float logit_bound_log2, hipStream_t stream);

}  // namespace mv
