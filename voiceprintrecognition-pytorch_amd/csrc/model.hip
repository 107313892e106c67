// Backbone forwards orchestrated natively: EcapaTdnn (mvector/models/ecapa_tdnn.py:253-283) and TDNN
// (mvector/models/tdnn.py:46-68), both ending in attentive statistics pooling (mvector/models/pooling.py:86-127).
//
// create(): takes the reference-layout fp32 state_dict (device pointers), folds eval-mode BatchNorm into
// per-channel (scale, shift), packs conv weights for the MFMA kernel, folds asp_bn / bn5 / bn6 into the final
// linear layer, splits the ASP attention conv into its time-varying part (a GEMM over x) and its per-utterance
// part (W[:, C:3C] . [mean; std], the "context bias").
// forward(): a fixed sequence of kernel launches on the caller's stream over a caller-provided workspace; no
// allocation, no synchronisation, so it can be captured in a hipGraph.
#include <map>
#include <memory>
#include <vector>

#include "kernels.h"
#include "model.h"

namespace mv {

// ------------------------------------------------------------------------------------------ helpers

int MvModelBase::info(int, float*) const { return fail(MV_ERR_INVALID_ARGUMENT, "mv_model_info: this model has no such key"); }

MvModelBase::~MvModelBase() {
    for (void* p : owned) hipFree(p);
}

void* MvModelBase::dev_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    owned.push_back(p);
    return p;
}

float* MvModelBase::upload(const std::vector<float>& v) {
    float* p = static_cast<float*>(dev_alloc(v.size() * sizeof(float)));
    if (p != nullptr && hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
}

int Weights::init(const MvTensorRef* tensors, int n) {
    MV_REQUIRE(tensors != nullptr && n > 0, "model create: empty tensor list");
    for (int i = 0; i < n; ++i) {
        MV_REQUIRE(tensors[i].name != nullptr && tensors[i].data != nullptr, "model create: null tensor entry");
        map[tensors[i].name] = tensors[i];
    }
    return MV_OK;
}

bool Weights::has(const std::string& name) const { return map.count(name) != 0; }

int Weights::dev(const std::string& name, int64_t numel, const float** out) const {
    auto it = map.find(name);
    if (it == map.end()) return fail(MV_ERR_MISSING_TENSOR, "state_dict is missing '" + name + "'");
    if (it->second.numel != numel)
        return fail(MV_ERR_INVALID_ARGUMENT, "tensor '" + name + "' has " + std::to_string(it->second.numel) +
                                                 " elements, expected " + std::to_string(numel));
    *out = it->second.data;
    return MV_OK;
}

int Weights::host(const std::string& name, int64_t numel, std::vector<float>& out) const {
    const float* d = nullptr;
    int rc = dev(name, numel, &d);
    if (rc != MV_OK) return rc;
    out.resize((size_t)numel);
    MV_HIP_OK(hipMemcpy(out.data(), d, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost));
    return MV_OK;
}

// eval-mode BatchNorm -> y = x*scale + shift  (affine optional: campplus.py:19-21)
int fold_bn(const Weights& w, const std::string& prefix, int C, std::vector<float>& scale, std::vector<float>& shift,
            float eps) {
    std::vector<float> mean, var, gamma, beta;
    int rc;
    if ((rc = w.host(prefix + ".running_mean", C, mean)) || (rc = w.host(prefix + ".running_var", C, var))) return rc;
    if (w.has(prefix + ".weight")) {
        if ((rc = w.host(prefix + ".weight", C, gamma)) || (rc = w.host(prefix + ".bias", C, beta))) return rc;
    } else {
        gamma.assign(C, 1.0f);
        beta.assign(C, 0.0f);
    }
    scale.resize(C);
    shift.resize(C);
    for (int c = 0; c < C; ++c) {
        const double s = (double)gamma[c] / sqrt((double)var[c] + (double)eps);
        scale[c] = (float)s;
        shift[c] = (float)((double)beta[c] - (double)mean[c] * s);
    }
    return MV_OK;
}

int MvModelBase::make_conv(const Weights& w, const std::string& weight_name, const std::string& bias_name, int cout,
                           int cin, int k, ConvLayer* out) {
    const float* dw = nullptr;
    int rc = w.dev(weight_name, (int64_t)cout * cin * k, &dw);
    if (rc != MV_OK) return rc;
    return make_conv_from(dw, bias_name.empty() ? nullptr : &w, bias_name, cout, cin, k, out);
}

int MvModelBase::make_conv_from(const float* dev_w, const Weights* w, const std::string& bias_name, int cout, int cin,
                                int k, ConvLayer* out) {
    out->cout = cout;
    out->cin = cin;
    out->k = k;
    const int64_t elems = mv_conv1d_packed_elems(cout, cin, k);
    out->w = static_cast<half_t*>(dev_alloc((size_t)elems * sizeof(half_t)));
    if (out->w == nullptr) return fail(MV_ERR_HIP, "model create: out of device memory for packed weights");
    int rc = mv_conv1d_pack_weight(dev_w, cout, cin, k, out->w, nullptr);
    if (rc != MV_OK) return rc;
    out->bias = nullptr;
    if (w != nullptr && !bias_name.empty() && w->has(bias_name)) {
        std::vector<float> b;
        if ((rc = w->host(bias_name, cout, b))) return rc;
        out->bias = upload(b);
        if (out->bias == nullptr) return fail(MV_ERR_HIP, "model create: bias upload failed");
    }
    return MV_OK;
}

int MvModelBase::make_bn(const Weights& w, const std::string& prefix, int C, float** scale, float** shift) {
    std::vector<float> s, t;
    int rc = fold_bn(w, prefix, C, s, t, 1e-5f);
    if (rc != MV_OK) return rc;
    *scale = upload(s);
    *shift = upload(t);
    if (*scale == nullptr || *shift == nullptr) return fail(MV_ERR_HIP, "model create: BN upload failed");
    return MV_OK;
}

// y = BN(ReLU(conv(x)))  -- TDNNBlock (models/utils.py:138) and TDNN.forward (tdnn.py:57-64)
int run_conv(const ConvLayer& L, const void* x, int x_dtype, int64_t ldx, const void* x2, int64_t ldx2, void* y, int y_dtype,
             int64_t ldy, int B, int T_in, int T_out, int dil, int pad, int pad_mode, int pre_act, const float* scale,
             const float* shift, int post_act, const float* row_bias, bool use_bias, hipStream_t stream,
             const half_t* add_src, int64_t ld_add, half_t* sum_dst, int64_t ld_sum, float* stat_sum, float* stat_sq) {
    MvConv1dDesc d;
    memset(&d, 0, sizeof(d));
    d.stat_sum = stat_sum;
    d.stat_sq = stat_sq;
    d.add_src = add_src;
    d.sum_dst = sum_dst;
    d.ld_add = ld_add;
    d.ld_sum = ld_sum;
    d.x = x;
    d.x2 = x2;
    d.x_dtype = x_dtype;
    d.ldx = ldx;
    d.ldx2 = ldx2;
    d.w_packed = L.w;
    d.bias = use_bias ? L.bias : nullptr;
    d.row_bias = row_bias;
    d.pre_act = pre_act;
    d.scale = scale;
    d.shift = shift;
    d.post_act = post_act;
    d.y = y;
    d.y_dtype = y_dtype;
    d.ldy = ldy;
    d.B = B;
    d.T_in = T_in;
    d.T_out = T_out;
    d.cin = L.cin;
    d.cout = L.cout;
    d.k = L.k;
    d.dilation = dil;
    d.stride = 1;
    d.pad = pad;
    d.pad_mode = pad_mode;
    return conv1d_launch(d, stream);
}

// --------------------------------------------------------------------------------------- ASP tail

int AspLayer::create(MvModelBase* m, const Weights& w, const std::string& prefix, int C_, int A_, bool global_ctx_) {
    C = C_;
    A = A_;
    global_ctx = global_ctx_;
    const int cin_total = global_ctx ? 3 * C : C;
    std::vector<float> wt;
    int rc = w.host(prefix + ".tdnn.conv.conv.weight", (int64_t)A * cin_total, wt);
    if (rc != MV_OK) return rc;
    std::vector<float> wx((size_t)A * C), wms;
    for (int a = 0; a < A; ++a) memcpy(&wx[(size_t)a * C], &wt[(size_t)a * cin_total], (size_t)C * sizeof(float));
    if (global_ctx) {
        wms.resize((size_t)A * 2 * C);
        for (int a = 0; a < A; ++a)
            memcpy(&wms[(size_t)a * 2 * C], &wt[(size_t)a * cin_total + C], (size_t)2 * C * sizeof(float));
        this->wms = m->upload(wms);
        if (this->wms == nullptr) return fail(MV_ERR_HIP, "asp create: upload failed");
    }
    // pack the x part; the temporary fp32 copy lives until create() synchronises
    float* tmp = m->upload(wx);
    if (tmp == nullptr) return fail(MV_ERR_HIP, "asp create: upload failed");
    if ((rc = m->make_conv_from(tmp, &w, prefix + ".tdnn.conv.conv.bias", A, C, 1, &tdnn))) return rc;
    if ((rc = m->make_bn(w, prefix + ".tdnn.norm.norm", A, &bn_scale, &bn_shift))) return rc;
    // attention projection asp.conv (C x A): stored times log2(e) so that the pooling kernel's softmax weight is a bare
    // 2^logit; its bias is constant over time and cancels in the softmax over time (pooling.py:117-119), so it is dropped.
    // h = tanh(.) is bounded by 1, so sum_k |W2[c,k]| bounds every logit (-> NOMAX form of the kernel)
    {
        const float log2e = 1.4426950408889634f;
        std::vector<float> w2;
        if ((rc = w.host(prefix + ".conv.conv.weight", (int64_t)C * A, w2))) return rc;
        double bound = 0.0;
        for (int c = 0; c < C; ++c) {
            double sabs = 0.0;
            for (int k = 0; k < A; ++k) sabs += fabs((double)w2[(size_t)c * A + k]);
            bound = sabs > bound ? sabs : bound;
        }
        for (float& v : w2) v *= log2e;
        logit_bound_log2 = std::isfinite(bound) ? (float)(bound * log2e * 1.001) : -1.0f;  // margin for the fp16 rounding of W2
        float* tmp2 = m->upload(w2);
        if (tmp2 == nullptr) return fail(MV_ERR_HIP, "asp create: upload failed");
        if ((rc = m->make_conv_from(tmp2, nullptr, "", C, A, 1, &conv))) return rc;
    }
    return MV_OK;
}

// [B, 2C] global mean | std, [B, A] context bias, the two partial buffers of the hidden conv's fused input statistics, the K slices of the
// context-bias layer (linear.hip, split-K form: [2C -> A] over B rows)
size_t AspLayer::workspace_floats(int B, int T) const {
    (void)T;
    return (size_t)B * (2 * C + A) + 2 * (size_t)conv_in_stats_elems(B, T, C) + linear_f32_splitk_floats(B, 2 * C, A);
}

// x: [B, T, ldx] fp16 -> pooled [B, 2C] fp32.  h: [B*T, A] fp16 scratch, fws: workspace_floats(B) fp32 scratch.
int AspLayer::forward(const half_t* x, int64_t ldx, int B, int T, half_t* h, float* fws, float* pooled,
                      hipStream_t stream, bool have_gstats) const {
    int rc;
    float* gstats = fws;                     // [B, 2C]  mean | std
    float* ctxb = fws + (size_t)B * 2 * C;   // [B, A]
    float* lin_ws = fws + (size_t)B * (2 * C + A) + 2 * (size_t)conv_in_stats_elems(B, T, C);
    const size_t lin_ws_floats = linear_f32_splitk_floats(B, 2 * C, A);
    const float* gmean = nullptr;
    if (global_ctx && !have_gstats && A <= 128 && A % 8 == 0 && ldx % 8 == 0) {
        // x is streamed ONCE for the global statistics and the hidden layer: the 1x1 conv over x collects the time sums of its own x
        // tiles and leaves the pre-activation z = Wx . x in h; the context columns [mean; std] enter as a per-utterance bias that
        // is added afterwards, together with ReLU -> BatchNorm -> tanh (asp_hidden_act_kernel, in place).  Saves the separate
        // statistics pass over x (469 MB at the bench shape).
        float* psum = ctxb + (size_t)B * A;
        float* psq = psum + conv_in_stats_elems(B, T, C);
        MvConv1dDesc d;
        memset(&d, 0, sizeof(d));
        d.x = x;
        d.x_dtype = MV_DT_F16;
        d.ldx = ldx;
        d.w_packed = tdnn.w;
        d.y = h;
        d.y_dtype = MV_DT_F16;
        d.ldy = A;
        d.B = B;
        d.T_in = d.T_out = T;
        d.cin = C;
        d.cout = A;
        d.k = 1;
        d.dilation = d.stride = 1;
        d.pad_mode = MV_PAD_REFLECT;
        d.in_stat_sum = psum;
        d.in_stat_sq = psq;
        if ((rc = conv1d_launch(d, stream))) return rc;
        if ((rc = conv_in_stats_finish_launch(psum, psq, B, T, C, gstats, gstats + C, 2 * C, 1e-12f, stream))) return rc;
        if ((rc = linear_f32_launch(gstats, 2 * C, wms, 2 * C, tdnn.bias, MV_ACT_NONE, ctxb, A, B, 2 * C, A, 0, stream, lin_ws, lin_ws_floats))) return rc;
        if ((rc = asp_hidden_act_launch(h, ctxb, bn_scale, bn_shift, B, T, A, stream))) return rc;
        return asp_pool_launch(h, conv.w, x, ldx, gstats, 2 * C, pooled, B, T, C, A, logit_bound_log2, stream);
    }
    if (!have_gstats) {  // else: already written by the producer's fused epilogue statistics
        if ((rc = time_stats_launch(x, ldx, B, T, C, gstats, gstats + C, 2 * C, 0, 1e-12f, stream))) return rc;
    }
    gmean = gstats;
    const float* row_bias = nullptr;
    if (global_ctx) {
        // context bias = W[:, C:3C] . [mean; std] + b  (pooling.py:104-117 with the T-constant columns hoisted)
        if ((rc = linear_f32_launch(gstats, 2 * C, wms, 2 * C, tdnn.bias, MV_ACT_NONE, ctxb, A, B, 2 * C, A, 0, stream, lin_ws, lin_ws_floats)))
            return rc;
        row_bias = ctxb;
    }
    // h = tanh(BN(ReLU(Wx . x + bias)))
    if ((rc = run_conv(tdnn, x, MV_DT_F16, ldx, nullptr, 0, h, MV_DT_F16, A, B, T, T, 1, 0, MV_PAD_REFLECT, MV_ACT_RELU,
                       bn_scale, bn_shift, MV_ACT_TANH, row_bias, /*use_bias=*/!global_ctx, stream)))
        return rc;
    return asp_pool_launch(h, conv.w, x, ldx, gmean, 2 * C, pooled, B, T, C, A, logit_bound_log2, stream);
}

// fold y = BN_out( W . BN_in(p) + b ) into one affine map (either BN optional)
int fold_final_linear(MvModelBase* m, const Weights& w, const std::string& weight_name, const std::string& bias_name,
                      const std::string& bn_in, const std::string& bn_out, int O, int K, float** wf_out, float** bf_out) {
    std::vector<float> W, b, s_in, t_in, s_out, t_out;
    int rc;
    if ((rc = w.host(weight_name, (int64_t)O * K, W))) return rc;
    if (!bias_name.empty() && w.has(bias_name)) {
        if ((rc = w.host(bias_name, O, b))) return rc;
    } else {
        b.assign(O, 0.0f);
    }
    if (!bn_in.empty()) {
        if ((rc = fold_bn(w, bn_in, K, s_in, t_in, 1e-5f))) return rc;
    } else {
        s_in.assign(K, 1.0f);
        t_in.assign(K, 0.0f);
    }
    if (!bn_out.empty()) {
        if ((rc = fold_bn(w, bn_out, O, s_out, t_out, 1e-5f))) return rc;
    } else {
        s_out.assign(O, 1.0f);
        t_out.assign(O, 0.0f);
    }
    std::vector<float> Wf((size_t)O * K), bf(O);
    for (int o = 0; o < O; ++o) {
        double acc = b[o];
        for (int k = 0; k < K; ++k) {
            acc += (double)W[(size_t)o * K + k] * t_in[k];
            Wf[(size_t)o * K + k] = (float)((double)W[(size_t)o * K + k] * s_in[k] * s_out[o]);
        }
        bf[o] = (float)(acc * s_out[o] + t_out[o]);
    }
    *wf_out = m->upload(Wf);
    *bf_out = m->upload(bf);
    if (*wf_out == nullptr || *bf_out == nullptr) return fail(MV_ERR_HIP, "final linear upload failed");
    return MV_OK;
}

// --------------------------------------------------------------------------------------- EcapaTdnn

struct EcapaModel : MvModelBase {
    MvEcapaCfg cfg;
    int nblocks = 0;  // SE-Res2Net blocks
    struct TdnnBlk {
        ConvLayer conv;
        float* scale = nullptr;
        float* shift = nullptr;
    };
    struct SeRes2 {
        int cin, cout, dil, k, width;
        TdnnBlk tdnn1, tdnn2;
        std::vector<TdnnBlk> res2;
        float* se_w1 = nullptr;  // [se, cout] fp32
        float* se_b1 = nullptr;
        float* se_w2 = nullptr;  // [cout, se]
        float* se_b2 = nullptr;
        bool has_shortcut = false;
        ConvLayer shortcut;
    };
    TdnnBlk block0, mfa;
    ConvLayer block0w;         // block 0 as a 1x1 conv over the contiguous k*F window of the reflect-padded features
    bool block0_window = false;
    std::vector<SeRes2> blocks;
    AspLayer asp;
    float* fc_w = nullptr;
    float* fc_b = nullptr;
    int ccat = 0, cmax = 0;

    int make_tdnn(const Weights& w, const std::string& prefix, int cout, int cin, int k, TdnnBlk* out) {
        int rc = make_conv(w, prefix + ".conv.conv.weight", prefix + ".conv.conv.bias", cout, cin, k, &out->conv);
        if (rc != MV_OK) return rc;
        return make_bn(w, prefix + ".norm.norm", cout, &out->scale, &out->shift);
    }

    int create(const MvEcapaCfg& c, const Weights& w) {
        cfg = c;
        input_size = c.input_size;
        embd_dim = c.embd_dim;
        int rc;
        MV_REQUIRE(c.res2net_scale >= 2 && c.res2net_scale <= 16, "ecapa: res2net_scale out of range");
        if ((rc = make_tdnn(w, "blocks.0", c.channels[0], c.input_size, c.kernel_sizes[0], &block0))) return rc;
        // Channel-last features with F % 8 == 0 and a dilation-1 first conv: the k taps of output step t are the k*F
        // CONTIGUOUS values starting at row t of the reflect-padded features, so block 0 is a 1x1 conv with cin = k*F and
        // row stride F (rows overlap).  K = 5*80 = 400 -> 7 stages of 64 instead of 5 taps x 128 (80 padded) = 10.
        block0_window = c.dilations[0] == 1 && c.input_size % 8 == 0 && c.kernel_sizes[0] > 1 && (c.kernel_sizes[0] % 2) == 1;
        if (block0_window) {
            const int F = c.input_size, k0 = c.kernel_sizes[0], C0 = c.channels[0];
            std::vector<float> w0, wr((size_t)C0 * k0 * F);
            if ((rc = w.host("blocks.0.conv.conv.weight", (int64_t)C0 * F * k0, w0))) return rc;
            for (int co = 0; co < C0; ++co)
                for (int ci = 0; ci < F; ++ci)
                    for (int j = 0; j < k0; ++j) wr[((size_t)co * k0 + j) * F + ci] = w0[((size_t)co * F + ci) * k0 + j];
            float* tmp = upload(wr);
            if (tmp == nullptr) return fail(MV_ERR_HIP, "ecapa create: upload failed");
            if ((rc = make_conv_from(tmp, &w, "blocks.0.conv.conv.bias", C0, k0 * F, 1, &block0w))) return rc;
        }
        nblocks = 3;
        blocks.resize(nblocks);
        ccat = 0;
        cmax = c.channels[0];
        for (int i = 0; i < nblocks; ++i) {
            SeRes2& b = blocks[i];
            b.cin = c.channels[i];
            b.cout = c.channels[i + 1];
            b.k = c.kernel_sizes[i + 1];
            b.dil = c.dilations[i + 1];
            MV_REQUIRE(b.cout % (8 * c.res2net_scale) == 0, "ecapa: channels must be a multiple of 8 * res2net_scale");
            b.width = b.cout / c.res2net_scale;
            const std::string p = "blocks." + std::to_string(i + 1);
            if ((rc = make_tdnn(w, p + ".tdnn1", b.cout, b.cin, 1, &b.tdnn1))) return rc;
            b.res2.resize(c.res2net_scale - 1);
            for (int j = 0; j < c.res2net_scale - 1; ++j)
                if ((rc = make_tdnn(w, p + ".res2net_block.blocks." + std::to_string(j), b.width, b.width, b.k, &b.res2[j])))
                    return rc;
            if ((rc = make_tdnn(w, p + ".tdnn2", b.cout, b.cout, 1, &b.tdnn2))) return rc;
            std::vector<float> t;
            if ((rc = w.host(p + ".se_block.conv1.conv.weight", (int64_t)c.se_channels * b.cout, t))) return rc;
            b.se_w1 = upload(t);
            if ((rc = w.host(p + ".se_block.conv1.conv.bias", c.se_channels, t))) return rc;
            b.se_b1 = upload(t);
            if ((rc = w.host(p + ".se_block.conv2.conv.weight", (int64_t)b.cout * c.se_channels, t))) return rc;
            b.se_w2 = upload(t);
            if ((rc = w.host(p + ".se_block.conv2.conv.bias", b.cout, t))) return rc;
            b.se_b2 = upload(t);
            b.has_shortcut = b.cin != b.cout;
            if (b.has_shortcut)
                if ((rc = make_conv(w, p + ".shortcut.conv.weight", p + ".shortcut.conv.bias", b.cout, b.cin, 1, &b.shortcut)))
                    return rc;
            ccat += b.cout;
            cmax = b.cout > cmax ? b.cout : cmax;
        }
        MV_REQUIRE(ccat == c.channels[4], "ecapa: channels[-1] must equal the sum of the SE-Res2Net block widths");
        if ((rc = make_tdnn(w, "mfa", c.channels[4], ccat, c.kernel_sizes[4], &mfa))) return rc;
        if ((rc = asp.create(this, w, "asp", c.channels[4], c.attention_channels, c.global_context != 0))) return rc;
        if ((rc = fold_final_linear(this, w, "fc.conv.weight", "fc.conv.bias", "asp_bn.norm", "", c.embd_dim, 2 * c.channels[4],
                                    &fc_w, &fc_b)))
            return rc;
        MV_HIP_OK(hipDeviceSynchronize());
        return MV_OK;
    }

    struct Ws {
        half_t *x16, *a0, *cat, *t1, *r2, *t2, *sc, *mfa, *h, *rs[2];
        float *se_mean, *se_hid, *gate, *asp_f, *pooled, *fc_ws;
        size_t bytes, fc_ws_floats;
    };

    Ws carve(void* base, int B, int T) const {
        const size_t N = (size_t)B * T;
        Carver c(base);
        Ws s;
        s.x16 = c.take<half_t>((size_t)B * (T + cfg.kernel_sizes[0]) * round_up(cfg.input_size, 8));  // incl. the reflect halo rows
        s.a0 = c.take<half_t>(N * cfg.channels[0]);
        s.cat = c.take<half_t>(N * ccat);
        s.rs[0] = c.take<half_t>(N * (cmax / cfg.res2net_scale));
        s.rs[1] = c.take<half_t>(N * (cmax / cfg.res2net_scale));
        s.t1 = c.take<half_t>(N * cmax);
        s.r2 = c.take<half_t>(N * cmax);
        s.t2 = c.take<half_t>(N * cmax);
        s.sc = c.take<half_t>(N * cmax);
        s.mfa = c.take<half_t>(N * cfg.channels[4]);
        s.h = c.take<half_t>(N * cfg.attention_channels);
        s.se_mean = c.take<float>((size_t)B * cmax);
        s.se_hid = c.take<float>((size_t)B * cfg.se_channels);
        s.gate = c.take<float>((size_t)B * cmax);
        s.asp_f = c.take<float>(asp.workspace_floats(B, T));
        s.pooled = c.take<float>((size_t)B * 2 * cfg.channels[4]);
        s.fc_ws_floats = linear_f32_splitk_floats(B, 2 * cfg.channels[4], cfg.embd_dim);   // K slices of the final layer (linear.hip)
        s.fc_ws = c.take<float>(s.fc_ws_floats);
        s.bytes = c.total();
        return s;
    }

    int workspace_bytes(int B, int T, size_t* bytes) const override {
        MV_REQUIRE(B > 0 && T > 0 && bytes != nullptr, "workspace_bytes: bad argument");
        *bytes = carve(nullptr, B, T).bytes;
        return MV_OK;
    }

    int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const override {
        MV_REQUIRE(feats != nullptr && emb != nullptr && ws != nullptr, "ecapa forward: null buffer");
        MV_REQUIRE(B > 0 && T > 0, "ecapa forward: empty batch");
        int maxpad = cfg.dilations[0] * (cfg.kernel_sizes[0] - 1) / 2;
        for (int i = 1; i < 5; ++i) {
            const int p = cfg.dilations[i] * (cfg.kernel_sizes[i] - 1) / 2;
            maxpad = p > maxpad ? p : maxpad;
        }
        MV_REQUIRE(T > maxpad, "ecapa forward: too few frames for the reflect padding (torch raises here as well)");
        const Ws s = carve(ws, B, T);
        if (s.bytes > ws_bytes) return fail(MV_ERR_WORKSPACE, "ecapa forward: workspace too small");
        int rc;
        const int R = MV_PAD_REFLECT;
        // features to fp16 once (12 MB), then blocks.0 on the direct-to-LDS path
        const int64_t ldf = round_up(cfg.input_size, 8);
        const int pad0 = cfg.dilations[0] * (cfg.kernel_sizes[0] - 1) / 2;
        if (block0_window) {
            if ((rc = cast_reflect_pad_launch(feats, s.x16, B, T, cfg.input_size, pad0, st))) return rc;
            if ((rc = run_conv(block0w, s.x16, MV_DT_F16, cfg.input_size, nullptr, 0, s.a0, MV_DT_F16, cfg.channels[0], B, T + 2 * pad0, T,
                               1, 0, MV_PAD_ZERO, MV_ACT_RELU, block0.scale, block0.shift, MV_ACT_NONE, nullptr, true, st)))
                return rc;
        } else {
            if ((rc = cast_rows_f32_f16_launch(feats, cfg.input_size, s.x16, ldf, (int64_t)B * T, cfg.input_size, st))) return rc;
            if ((rc = run_conv(block0.conv, s.x16, MV_DT_F16, ldf, nullptr, 0, s.a0, MV_DT_F16, cfg.channels[0], B, T, T,
                               cfg.dilations[0], pad0, R, MV_ACT_RELU, block0.scale, block0.shift, MV_ACT_NONE, nullptr, true, st)))
                return rc;
        }
        const half_t* xin = s.a0;
        int64_t ldin = cfg.channels[0];
        int cat_off = 0;
        for (int i = 0; i < nblocks; ++i) {
            const SeRes2& b = blocks[i];
            const int C = b.cout;
            const half_t* res = xin;
            int64_t ldres = ldin;
            if (b.has_shortcut) {
                if ((rc = run_conv(b.shortcut, xin, MV_DT_F16, ldin, nullptr, 0, s.sc, MV_DT_F16, C, B, T, T, 1, 0, R, MV_ACT_NONE,
                                   nullptr, nullptr, MV_ACT_NONE, nullptr, true, st)))
                    return rc;
                res = s.sc;
                ldres = C;
            }
            if ((rc = run_conv(b.tdnn1.conv, xin, MV_DT_F16, ldin, nullptr, 0, s.t1, MV_DT_F16, C, B, T, T, 1, 0, R, MV_ACT_RELU,
                               b.tdnn1.scale, b.tdnn1.shift, MV_ACT_NONE, nullptr, true, st)))
                return rc;
            const int steps = cfg.res2net_scale - 1;
            if (res2_chain_supported(T, b.width, steps, b.k, b.dil)) {
                // whole chain in one launch, one workgroup per utterance (res2.hip)
                const half_t* wp[16];
                const float *bp[16], *sp[16], *tp[16];
                for (int j = 0; j < steps; ++j) {
                    wp[j] = b.res2[j].conv.w;
                    bp[j] = b.res2[j].conv.bias;
                    sp[j] = b.res2[j].scale;
                    tp[j] = b.res2[j].shift;
                }
                if ((rc = res2_chain_launch(s.t1, s.r2, wp, bp, sp, tp, B, T, C, b.width, steps, b.k, b.dil, st))) return rc;
            } else {
                // Res2Net: slice 0 passes through, slice j = blk_{j-1}(x_j [+ y_{j-1}]).  Step j's epilogue also emits the next
                // step's input x_{j+1} + y_j into a ping-pong scratch slice, so every step reads one plain fp16 tensor.
                if ((rc = copy_slice_launch(s.t1, C, s.r2, C, b.width, (int64_t)B * T, st))) return rc;
                const int pad = b.dil * (b.k - 1) / 2;
                for (int j = 1; j < cfg.res2net_scale; ++j) {
                    const half_t* in = j == 1 ? s.t1 + (size_t)b.width : s.rs[j & 1];
                    const int64_t ldin_j = j == 1 ? C : b.width;
                    const bool more = j + 1 < cfg.res2net_scale;
                    if ((rc = run_conv(b.res2[j - 1].conv, in, MV_DT_F16, ldin_j, nullptr, 0, s.r2 + (size_t)j * b.width, MV_DT_F16, C, B,
                                       T, T, b.dil, pad, R, MV_ACT_RELU, b.res2[j - 1].scale, b.res2[j - 1].shift, MV_ACT_NONE, nullptr,
                                       true, st, more ? s.t1 + (size_t)(j + 1) * b.width : nullptr, C, more ? s.rs[(j + 1) & 1] : nullptr,
                                       b.width)))
                        return rc;
                }
            }
            // SE: squeeze -> FC/ReLU -> FC/sigmoid -> gate * y + residual, written into the aggregation slice.  The squeeze (mean over
            // time, ecapa_tdnn.py:79) is its own single pass: taken in tdnn2's epilogue (MvConv1dDesc.stat_sum) it costs that layer as
            // much as the pass it replaces (r02a: +32 us against 29 us).
            if ((rc = run_conv(b.tdnn2.conv, s.r2, MV_DT_F16, C, nullptr, 0, s.t2, MV_DT_F16, C, B, T, T, 1, 0, R, MV_ACT_RELU,
                               b.tdnn2.scale, b.tdnn2.shift, MV_ACT_NONE, nullptr, true, st)))
                return rc;
            if ((rc = time_stats_launch(s.t2, C, B, T, C, s.se_mean, nullptr, C, 0, 0.0f, st))) return rc;
            if ((rc = linear_f32_launch(s.se_mean, C, b.se_w1, C, b.se_b1, MV_ACT_RELU, s.se_hid, cfg.se_channels, B, C,
                                        cfg.se_channels, 0, st)))
                return rc;
            if ((rc = linear_f32_launch(s.se_hid, cfg.se_channels, b.se_w2, cfg.se_channels, b.se_b2, MV_ACT_SIGMOID, s.gate, C,
                                        B, cfg.se_channels, C, 0, st)))
                return rc;
            half_t* out = s.cat + cat_off;
            if ((rc = se_gate_residual_launch(s.t2, C, s.gate, res, ldres, out, ccat, B, T, C, st))) return rc;
            xin = out;
            ldin = ccat;
            cat_off += C;
        }
        // multi-layer feature aggregation reads the three block outputs in place
        const int Cm = cfg.channels[4];
        // (the ASP global mean / std of pooling.py:104-109 are collected by the ASP hidden conv from its own input tiles: AspLayer::forward)
        if ((rc = run_conv(mfa.conv, s.cat, MV_DT_F16, ccat, nullptr, 0, s.mfa, MV_DT_F16, Cm, B, T, T, cfg.dilations[4],
                           cfg.dilations[4] * (cfg.kernel_sizes[4] - 1) / 2, R, MV_ACT_RELU, mfa.scale, mfa.shift, MV_ACT_NONE,
                           nullptr, true, st)))
            return rc;
        if ((rc = asp.forward(s.mfa, Cm, B, T, s.h, s.asp_f, s.pooled, st))) return rc;
        // asp_bn folded into fc
        return linear_f32_launch(s.pooled, 2 * Cm, fc_w, 2 * Cm, fc_b, MV_ACT_NONE, emb, cfg.embd_dim, B, 2 * Cm, cfg.embd_dim, 0,
                                 st, s.fc_ws, s.fc_ws_floats);
    }
};

// --------------------------------------------------------------------------------------- TDNN (x-vector)

struct TdnnModel : MvModelBase {
    MvTdnnCfg cfg;
    ConvLayer conv[5];
    float* scale[4] = {nullptr, nullptr, nullptr, nullptr};
    float* shift[4] = {nullptr, nullptr, nullptr, nullptr};
    AspLayer asp;
    float* fc_w = nullptr;
    float* fc_b = nullptr;
    static constexpr int K[5] = {5, 3, 3, 1, 1};
    static constexpr int D[5] = {1, 2, 3, 1, 1};

    int create(const MvTdnnCfg& c, const Weights& w) {
        cfg = c;
        input_size = c.input_size;
        embd_dim = c.embd_dim;
        MV_REQUIRE(c.channels % 8 == 0, "tdnn: channels must be a multiple of 8");
        int rc;
        for (int i = 0; i < 5; ++i) {
            const std::string p = "td_layer" + std::to_string(i + 1);
            if ((rc = make_conv(w, p + ".weight", p + ".bias", c.channels, i == 0 ? c.input_size : c.channels, K[i], &conv[i])))
                return rc;
            if (i < 4)
                if ((rc = make_bn(w, "bn" + std::to_string(i + 1), c.channels, &scale[i], &shift[i]))) return rc;
        }
        if ((rc = asp.create(this, w, "pooling", c.channels, 128, true))) return rc;
        if ((rc = fold_final_linear(this, w, "linear.weight", "linear.bias", "bn5", "bn6", c.embd_dim, 2 * c.channels, &fc_w,
                                    &fc_b)))
            return rc;
        MV_HIP_OK(hipDeviceSynchronize());
        return MV_OK;
    }

    struct Ws {
        half_t *a, *b, *h;
        float *asp_f, *pooled;
        size_t bytes;
    };
    Ws carve(void* base, int B, int T) const {
        const size_t N = (size_t)B * T;
        Carver c(base);
        Ws s;
        s.a = c.take<half_t>(N * cfg.channels);
        s.b = c.take<half_t>(N * cfg.channels);
        s.h = c.take<half_t>(N * 128);
        s.asp_f = c.take<float>(asp.workspace_floats(B, T));
        s.pooled = c.take<float>((size_t)B * 2 * cfg.channels);
        s.bytes = c.total();
        return s;
    }
    int workspace_bytes(int B, int T, size_t* bytes) const override {
        MV_REQUIRE(B > 0 && T > 0 && bytes != nullptr, "workspace_bytes: bad argument");
        *bytes = carve(nullptr, B, T).bytes;
        return MV_OK;
    }
    int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const override {
        MV_REQUIRE(feats != nullptr && emb != nullptr && ws != nullptr, "tdnn forward: null buffer");
        MV_REQUIRE(B > 0 && T > 14, "tdnn forward: the unpadded convolutions need more than 14 frames");
        const Ws s = carve(ws, B, T);
        if (s.bytes > ws_bytes) return fail(MV_ERR_WORKSPACE, "tdnn forward: workspace too small");
        int rc;
        const void* x = feats;
        int xdt = MV_DT_F32;
        int64_t ldx = cfg.input_size;
        int Tin = T;
        half_t* bufs[2] = {s.a, s.b};
        for (int i = 0; i < 5; ++i) {
            const int Tout = Tin - D[i] * (K[i] - 1);
            half_t* y = bufs[i & 1];
            if ((rc = run_conv(conv[i], x, xdt, ldx, nullptr, 0, y, MV_DT_F16, cfg.channels, B, Tin, Tout, D[i], 0, MV_PAD_ZERO,
                               MV_ACT_RELU, i < 4 ? scale[i] : nullptr, i < 4 ? shift[i] : nullptr, MV_ACT_NONE, nullptr, true,
                               st)))
                return rc;
            x = y;
            xdt = MV_DT_F16;
            ldx = cfg.channels;
            Tin = Tout;
        }
        if ((rc = asp.forward(static_cast<const half_t*>(x), cfg.channels, B, Tin, s.h, s.asp_f, s.pooled, st))) return rc;
        return linear_f32_launch(s.pooled, 2 * cfg.channels, fc_w, 2 * cfg.channels, fc_b, MV_ACT_NONE, emb, cfg.embd_dim, B,
                                 2 * cfg.channels, cfg.embd_dim, 0, st);
    }
};
constexpr int TdnnModel::K[5];
constexpr int TdnnModel::D[5];

}  // namespace mv

extern "C" {

int mv_ecapa_create(const MvEcapaCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_ecapa_create: null argument");
    mv::Weights w;
    int rc = w.init(tensors, num_tensors);
    if (rc != MV_OK) return rc;
    auto m = std::make_unique<mv::EcapaModel>();
    rc = m->create(*cfg, w);
    if (rc != MV_OK) return rc;
    *out = reinterpret_cast<MvModel*>(static_cast<mv::MvModelBase*>(m.release()));
    return MV_OK;
}

int mv_tdnn_create(const MvTdnnCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_tdnn_create: null argument");
    mv::Weights w;
    int rc = w.init(tensors, num_tensors);
    if (rc != MV_OK) return rc;
    auto m = std::make_unique<mv::TdnnModel>();
    rc = m->create(*cfg, w);
    if (rc != MV_OK) return rc;
    *out = reinterpret_cast<MvModel*>(static_cast<mv::MvModelBase*>(m.release()));
    return MV_OK;
}

int mv_model_destroy(MvModel* m) {
    delete reinterpret_cast<mv::MvModelBase*>(m);
    return MV_OK;
}

int mv_model_embd_dim(const MvModel* m, int32_t* embd_dim) {
    MV_REQUIRE(m != nullptr && embd_dim != nullptr, "mv_model_embd_dim: null argument");
    *embd_dim = reinterpret_cast<const mv::MvModelBase*>(m)->embd_dim;
    return MV_OK;
}

int mv_model_info(const MvModel* m, int32_t key, float* value) {
    MV_REQUIRE(m != nullptr && value != nullptr, "mv_model_info: null argument");
    return reinterpret_cast<const mv::MvModelBase*>(m)->info(key, value);
}

int mv_model_workspace_bytes(const MvModel* m, int32_t B, int32_t T, size_t* bytes) {
    MV_REQUIRE(m != nullptr, "mv_model_workspace_bytes: null model");
    return reinterpret_cast<const mv::MvModelBase*>(m)->workspace_bytes(B, T, bytes);
}

int mv_model_forward(const MvModel* m, const float* feats, int32_t B, int32_t T, float* emb, void* workspace,
                     size_t workspace_bytes, mv_stream_t stream) {
    MV_REQUIRE(m != nullptr, "mv_model_forward: null model");
    return reinterpret_cast<const mv::MvModelBase*>(m)->forward(feats, B, T, emb, workspace, workspace_bytes,
                                                                static_cast<hipStream_t>(stream));
}

}  // extern "C"
