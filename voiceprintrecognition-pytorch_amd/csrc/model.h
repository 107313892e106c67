// Internal model scaffolding shared by model.hip (EcapaTdnn, TDNN) and campplus.hip (CAM++).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace mv {

struct Weights {
    std::map<std::string, MvTensorRef> map;
    int init(const MvTensorRef* tensors, int n);
    bool has(const std::string& name) const;
    int dev(const std::string& name, int64_t numel, const float** out) const;       // device pointer, size-checked
    int host(const std::string& name, int64_t numel, std::vector<float>& out) const;  // host copy, size-checked
};

struct ConvLayer {
    half_t* w = nullptr;    // packed fp16 [Cout_pad][k][Cin_pad]
    float* bias = nullptr;  // [cout] or null
    int cout = 0, cin = 0, k = 1;
};

struct MvModelBase {
    int embd_dim = 0;
    int input_size = 0;
    std::vector<void*> owned;  // device allocations released by the destructor
    virtual ~MvModelBase();
    virtual int workspace_bytes(int B, int T, size_t* bytes) const = 0;
    virtual int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const = 0;
    // model-specific facts for tests and logs (mv_model_info): MV_INFO_* keys; unknown key -> error
    virtual int info(int key, float* value) const;

    void* dev_alloc(size_t bytes);
    float* upload(const std::vector<float>& v);
    int make_conv(const Weights& w, const std::string& weight_name, const std::string& bias_name, int cout, int cin, int k,
                  ConvLayer* out);
    int make_conv_from(const float* dev_w, const Weights* w, const std::string& bias_name, int cout, int cin, int k,
                       ConvLayer* out);
    int make_bn(const Weights& w, const std::string& prefix, int C, float** scale, float** shift);
};

int fold_bn(const Weights& w, const std::string& prefix, int C, std::vector<float>& scale, std::vector<float>& shift,
            float eps);
int fold_final_linear(MvModelBase* m, const Weights& w, const std::string& weight_name, const std::string& bias_name,
                      const std::string& bn_in, const std::string& bn_out, int O, int K, float** wf_out, float** bf_out);

int run_conv(const ConvLayer& L, const void* x, int x_dtype, int64_t ldx, const void* x2, int64_t ldx2, void* y, int y_dtype,
             int64_t ldy, int B, int T_in, int T_out, int dil, int pad, int pad_mode, int pre_act, const float* scale,
             const float* shift, int post_act, const float* row_bias, bool use_bias, hipStream_t stream,
             const half_t* add_src = nullptr, int64_t ld_add = 0, half_t* sum_dst = nullptr, int64_t ld_sum = 0,
             float* stat_sum = nullptr, float* stat_sq = nullptr);

// bump allocator over the caller-provided workspace (256-byte aligned slices)
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

// attentive statistics pooling (mvector/models/pooling.py:68-127)
struct AspLayer {
    int C = 0, A = 0;
    bool global_ctx = true;
    ConvLayer tdnn;   // x part of asp.tdnn (A x C)
    float* wms = nullptr;  // [A][2C] fp32: columns of asp.tdnn that multiply [mean; std]
    float* bn_scale = nullptr;
    float* bn_shift = nullptr;
    ConvLayer conv;   // asp.conv (C x A), weights times log2(e); the bias cancels in the softmax over time
    float logit_bound_log2 = -1.0f;  // max_c sum_k |W2[c,k]| * log2(e): bounds every attention logit
    int create(MvModelBase* m, const Weights& w, const std::string& prefix, int C, int A, bool global_ctx);
    size_t workspace_floats(int B, int T) const;
    // have_gstats: fws already holds the global mean | std of x ([B, 2C]) from the producer's fused statistics
    int forward(const half_t* x, int64_t ldx, int B, int T, half_t* h, float* fws, float* pooled, hipStream_t stream,
                bool have_gstats = false) const;
};

}  // namespace mv
