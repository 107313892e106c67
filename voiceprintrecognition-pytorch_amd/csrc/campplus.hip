// CAM++ forward orchestrated natively (mvector/models/campplus.py:295-357).
//
//   head      FCM: conv3x3(1->32)+BN+ReLU, 2 x [BasicResBlock(stride (2,1)), BasicResBlock], conv3x3 stride (2,1)+BN+ReLU
//             on channel-last maps [B, F, T, 32] (fcm.hip); the last conv writes [B, T, F/8, 32], i.e. the 320-channel
//             row the TDNN reads (the reference's reshape order c*10+f is absorbed into the packed TDNN weight).
//   xvector   tdnn (k=5, stride 2, BN, ReLU) -> three CAM dense-TDNN blocks.  A block owns ONE [B, T2, C_final] buffer and
//             every layer writes its 32 new channels into its slice (the reference re-copies the growing tensor with
//             torch.cat on each of the 52 layers, campplus.py:178-181).  Per layer:
//                 h   = ReLU(BN2(W1 . ReLU(BN1(x))))           1x1 conv, BN1/ReLU applied on load, BN2/ReLU in the epilogue
//                 ctx = mean_T(h) + segmean_100(h)             (campplus.py:94-111) -> [B, nseg, 128]
//                 m   = sigmoid(W_b . ReLU(W_a . ctx + b_a) + b_b)   evaluated once per segment, not per frame
//                 y   = conv_k3(h) * m                          gate multiplied in the conv epilogue
//             (one launch per layer for utterances of up to 160 strided frames: camdense.hip; five launches beyond)
//             transit: 1x1 conv with BN/ReLU on load; out_nonlinear + StatsPool (unbiased std) fused into one reduction;
//             dense + BN(affine=False) folded into one fp32 linear layer.
#include <array>
#include <memory>

#include "kernels.h"
#include "model.h"

namespace mv {

struct CamppModel : MvModelBase {
    MvCamppCfg cfg;
    // head
    float* c1_w = nullptr;  // [32][9] fp32 (BN folded)
    float* c1_b = nullptr;
    struct Conv2d {
        half_t* w = nullptr;  // [ntaps][32][32]
        float* bias = nullptr;
        int ntaps = 9;
    };
    struct ResBlock {
        Conv2d conv1, conv2;  // conv2 carries the shortcut tap when the block has one
        bool has_shortcut = false;
        int stride = 1;
    };
    ResBlock res[4];
    Conv2d head_out;
    // xvector
    ConvLayer tdnn;
    float *tdnn_scale = nullptr, *tdnn_shift = nullptr;
    struct DenseLayer {
        int cin;
        float *bn1_s, *bn1_t, *bn2_s, *bn2_t;
        ConvLayer lin1, local;
        float *wa, *ba, *wb, *bb;  // context FCs 128 -> 64 -> 32 (fp32)
    };
    struct Block {
        int c_in, c_out, dil;
        std::vector<DenseLayer> layers;
        float *tr_s, *tr_t;  // transit BN
        ConvLayer transit;
    };
    Block blocks[3];
    float *out_s = nullptr, *out_t = nullptr;
    float *dense_w = nullptr, *dense_b = nullptr;
    int F8 = 0, bn_ch = 0, cfin = 0;

    int fold_conv2d(const Weights& w, const std::string& conv, const std::string& bn, const std::string& sc_conv,
                    const std::string& sc_bn, Conv2d* out) {
        std::vector<float> W, s, t, Ws, ss, ts;
        int rc;
        if ((rc = w.host(conv + ".weight", 32 * 32 * 9, W)) || (rc = fold_bn(w, bn, 32, s, t, 1e-5f))) return rc;
        const bool sc = !sc_conv.empty();
        if (sc)
            if ((rc = w.host(sc_conv + ".weight", 32 * 32, Ws)) || (rc = fold_bn(w, sc_bn, 32, ss, ts, 1e-5f))) return rc;
        out->ntaps = sc ? 10 : 9;
        std::vector<float> packed((size_t)out->ntaps * 32 * 32), bias(32);
        for (int co = 0; co < 32; ++co) {
            bias[co] = t[co] + (sc ? ts[co] : 0.0f);
            for (int ci = 0; ci < 32; ++ci) {
                for (int tap = 0; tap < 9; ++tap)
                    packed[((size_t)tap * 32 + co) * 32 + ci] = W[((size_t)co * 32 + ci) * 9 + tap] * s[co];
                if (sc) packed[((size_t)9 * 32 + co) * 32 + ci] = Ws[(size_t)co * 32 + ci] * ss[co];
            }
        }
        // fp32 -> fp16 on the host (round-to-nearest-even via _Float16)
        std::vector<half_t> ph(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) ph[i] = (half_t)packed[i];
        out->w = static_cast<half_t*>(dev_alloc(ph.size() * sizeof(half_t)));
        if (out->w == nullptr) return fail(MV_ERR_HIP, "campp create: out of device memory");
        MV_HIP_OK(hipMemcpy(out->w, ph.data(), ph.size() * sizeof(half_t), hipMemcpyHostToDevice));
        out->bias = upload(bias);
        return out->bias ? MV_OK : fail(MV_ERR_HIP, "campp create: upload failed");
    }

    int create(const MvCamppCfg& c, const Weights& w) {
        cfg = c;
        input_size = c.input_size;
        embd_dim = c.embd_dim;
        MV_REQUIRE(c.input_size >= 8, "campp: input_size must be at least 8");
        MV_REQUIRE(c.growth_rate == 32 && c.bn_size * c.growth_rate == 128 && c.init_channels % 64 == 0,
                   "campp: only growth_rate=32, bn_size=4 and init_channels multiple of 64 are implemented");
        F8 = (c.input_size + 7) / 8;
        bn_ch = c.bn_size * c.growth_rate;
        int rc;
        {  // head.conv1 + bn1 (one input map)
            std::vector<float> W, s, t;
            if ((rc = w.host("head.conv1.weight", 32 * 9, W)) || (rc = fold_bn(w, "head.bn1", 32, s, t, 1e-5f))) return rc;
            for (int co = 0; co < 32; ++co)
                for (int j = 0; j < 9; ++j) W[co * 9 + j] *= s[co];
            c1_w = upload(W);
            c1_b = upload(t);
        }
        const char* names[4] = {"head.layer1.0", "head.layer1.1", "head.layer2.0", "head.layer2.1"};
        for (int i = 0; i < 4; ++i) {
            const std::string p = names[i];
            res[i].has_shortcut = (i % 2 == 0);
            res[i].stride = res[i].has_shortcut ? 2 : 1;
            if ((rc = fold_conv2d(w, p + ".conv1", p + ".bn1", "", "", &res[i].conv1))) return rc;
            if ((rc = fold_conv2d(w, p + ".conv2", p + ".bn2", res[i].has_shortcut ? p + ".shortcut.0" : "",
                                  res[i].has_shortcut ? p + ".shortcut.1" : "", &res[i].conv2)))
                return rc;
        }
        if ((rc = fold_conv2d(w, "head.conv2", "head.bn2", "", "", &head_out))) return rc;
        {  // xvector.tdnn: weight [init, 32*F8, 5] with input channel c*F8+f  ->  our row layout f*32+c
            const int cin = 32 * F8;
            std::vector<float> W;
            if ((rc = w.host("xvector.tdnn.linear.weight", (int64_t)c.init_channels * cin * 5, W))) return rc;
            std::vector<float> P(W.size());
            for (int co = 0; co < c.init_channels; ++co)
                for (int ch = 0; ch < 32; ++ch)
                    for (int f = 0; f < F8; ++f)
                        for (int j = 0; j < 5; ++j)
                            P[((size_t)co * cin + (f * 32 + ch)) * 5 + j] = W[((size_t)co * cin + (ch * F8 + f)) * 5 + j];
            float* tmp = upload(P);
            if (tmp == nullptr) return fail(MV_ERR_HIP, "campp create: upload failed");
            if ((rc = make_conv_from(tmp, nullptr, "", c.init_channels, cin, 5, &tdnn))) return rc;
            if ((rc = make_bn(w, "xvector.tdnn.nonlinear.batchnorm", c.init_channels, &tdnn_scale, &tdnn_shift))) return rc;
        }
        const int nlayers[3] = {12, 24, 16};
        const int dils[3] = {1, 2, 2};
        int channels = c.init_channels;
        for (int bi = 0; bi < 3; ++bi) {
            Block& B = blocks[bi];
            B.c_in = channels;
            B.dil = dils[bi];
            B.layers.resize(nlayers[bi]);
            for (int li = 0; li < nlayers[bi]; ++li) {
                DenseLayer& L = B.layers[li];
                L.cin = channels + li * c.growth_rate;
                const std::string p = "xvector.block" + std::to_string(bi + 1) + ".tdnnd" + std::to_string(li + 1);
                if ((rc = make_bn(w, p + ".nonlinear1.batchnorm", L.cin, &L.bn1_s, &L.bn1_t))) return rc;
                if ((rc = make_conv(w, p + ".linear1.weight", "", bn_ch, L.cin, 1, &L.lin1))) return rc;
                if ((rc = make_bn(w, p + ".nonlinear2.batchnorm", bn_ch, &L.bn2_s, &L.bn2_t))) return rc;
                if ((rc = make_conv(w, p + ".cam_layer.linear_local.weight", "", c.growth_rate, bn_ch, 3, &L.local))) return rc;
                std::vector<float> t;
                if ((rc = w.host(p + ".cam_layer.linear1.weight", (int64_t)(bn_ch / 2) * bn_ch, t))) return rc;
                L.wa = upload(t);
                if ((rc = w.host(p + ".cam_layer.linear1.bias", bn_ch / 2, t))) return rc;
                L.ba = upload(t);
                if ((rc = w.host(p + ".cam_layer.linear2.weight", (int64_t)c.growth_rate * (bn_ch / 2), t))) return rc;
                L.wb = upload(t);
                if ((rc = w.host(p + ".cam_layer.linear2.bias", c.growth_rate, t))) return rc;
                L.bb = upload(t);
            }
            channels += nlayers[bi] * c.growth_rate;
            B.c_out = channels;
            const std::string tp = "xvector.transit" + std::to_string(bi + 1);
            if ((rc = make_bn(w, tp + ".nonlinear.batchnorm", channels, &B.tr_s, &B.tr_t))) return rc;
            if ((rc = make_conv(w, tp + ".linear.weight", tp + ".linear.bias", channels / 2, channels, 1, &B.transit))) return rc;
            channels /= 2;
        }
        cfin = channels;
        if ((rc = make_bn(w, "xvector.out_nonlinear.batchnorm", cfin, &out_s, &out_t))) return rc;
        if ((rc = fold_final_linear(this, w, "xvector.dense.linear.weight", "xvector.dense.linear.bias", "",
                                    "xvector.dense.nonlinear.batchnorm", c.embd_dim, 2 * cfin, &dense_w, &dense_b)))
            return rc;
        MV_HIP_OK(hipDeviceSynchronize());
        return MV_OK;
    }

    struct Ws {
        half_t *m0, *m1, *m2;  // FCM maps (ping-pong)
        half_t* rows;          // [B, T, 32*F8]
        half_t* xb[3];         // dense-block buffers [B, T2, c_out]
        half_t* last;          // [B, T2, cfin]
        half_t* h;             // [B, T2, 128]
        float *ctx, *g1, *gate, *stats;
        size_t bytes;
        int T2, nseg;
    };

    Ws carve(void* base, int B, int T) const {
        Carver c(base);
        Ws s;
        const int F = cfg.input_size;
        s.T2 = (T - 1) / 2 + 1;
        s.nseg = (s.T2 + 99) / 100;
        const size_t full = (size_t)B * F * T * 32;
        s.m0 = c.take<half_t>(full);
        s.m1 = c.take<half_t>(full / 2);
        s.m2 = c.take<half_t>(full / 2);
        s.rows = c.take<half_t>((size_t)B * T * 32 * F8);
        const size_t N2 = (size_t)B * s.T2;
        for (int i = 0; i < 3; ++i) s.xb[i] = c.take<half_t>(N2 * blocks[i].c_out);
        s.last = c.take<half_t>(N2 * cfin);
        s.h = c.take<half_t>(N2 * bn_ch);
        s.ctx = c.take<float>((size_t)B * s.nseg * bn_ch);
        s.g1 = c.take<float>((size_t)B * s.nseg * (bn_ch / 2));
        s.gate = c.take<float>((size_t)B * s.nseg * cfg.growth_rate);
        s.stats = c.take<float>((size_t)B * 2 * cfin);
        s.bytes = c.total();
        return s;
    }

    int workspace_bytes(int B, int T, size_t* bytes) const override {
        MV_REQUIRE(B > 0 && T > 0 && bytes != nullptr, "workspace_bytes: bad argument");
        *bytes = carve(nullptr, B, T).bytes;
        return MV_OK;
    }

    int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const override {
        MV_REQUIRE(feats != nullptr && emb != nullptr && ws != nullptr, "campp forward: null buffer");
        MV_REQUIRE(B > 0 && T >= 3, "campp forward: needs at least 3 frames (unbiased std over the strided time axis)");
        const Ws s = carve(ws, B, T);
        if (s.bytes > ws_bytes) return fail(MV_ERR_WORKSPACE, "campp forward: workspace too small");
        const int F = cfg.input_size;
        int rc;
        // ---- head ----
        if ((rc = fcm_conv1_launch(feats, s.m0, c1_w, c1_b, B, T, F, st))) return rc;
        auto plain = [&](int Fd) { return std::array<int64_t, 3>{(int64_t)Fd * T * 32, (int64_t)T * 32, 32}; };
        const half_t* cur = s.m0;
        int Fc = F;
        half_t* pp[2] = {s.m1, s.m2};
        // MV_FCM_FUSED=0 (measurement knob) keeps the two launches per BasicResBlock with the intermediate map in HBM
        const char* fcm_env = getenv("MV_FCM_FUSED");
        const bool fused_block = !(fcm_env != nullptr && fcm_env[0] == '0');
        for (int i = 0; i < 4; ++i) {
            const ResBlock& r = res[i];
            const int Fo = (Fc - 1) / r.stride + 1;
            auto so = plain(Fo);
            if (fused_block) {
                // one launch per block (fcmblock.hip): x read once, mid map in LDS, output written once
                half_t* t2 = (cur == pp[0]) ? pp[1] : pp[0];
                if (fcm_block_supported(t2, so[0], so[1], so[2], T, Fc)) {
                    if ((rc = fcm_block_launch(cur, Fc, r.stride, r.conv1.w, r.conv1.bias, r.conv2.w, r.conv2.bias, r.has_shortcut ? 1 : 0, t2,
                                               so[0], so[1], so[2], B, T, st)))
                        return rc;
                    cur = t2;
                    Fc = Fo;
                    continue;
                }
            }
            // conv1 (stride on the frequency axis) + BN + ReLU
            half_t* t1 = (cur == pp[0]) ? pp[1] : pp[0];
            if ((rc = fcm_conv3x3_launch(cur, Fc, r.stride, nullptr, 0, 1, 0, r.conv1.w, r.conv1.bias, t1, so[0], so[1], so[2], B, T,
                                         Fo, st)))
                return rc;
            // conv2 + BN + (shortcut conv+BN | identity) + ReLU.  Output must not alias either input.
            half_t* t2;
            if (cur == s.m0) {
                t2 = (t1 == pp[0]) ? pp[1] : pp[0];
            } else {
                t2 = s.m0;  // m0 is free once the first block has consumed it (it is larger than needed)
            }
            if ((rc = fcm_conv3x3_launch(t1, Fo, 1, cur, Fc, r.stride, r.has_shortcut ? 1 : 2, r.conv2.w, r.conv2.bias, t2, so[0],
                                         so[1], so[2], B, T, Fo, st)))
                return rc;
            cur = t2;
            Fc = Fo;
        }
        {  // head.conv2 (stride 2 in frequency) -> rows [B, T, F8, 32]
            const int Fo = (Fc - 1) / 2 + 1;
            MV_REQUIRE(Fo == F8, "campp forward: unexpected frequency size after the head");
            if ((rc = fcm_conv3x3_launch(cur, Fc, 2, nullptr, 0, 1, 0, head_out.w, head_out.bias, s.rows, (int64_t)T * F8 * 32, 32,
                                         (int64_t)F8 * 32, B, T, Fo, st)))
                return rc;
        }
        // ---- xvector.tdnn: k=5, stride 2, zero pad 2, BN, ReLU -> first slice of block 1's buffer ----
        const int T2 = s.T2;
        {
            MvConv1dDesc d;
            memset(&d, 0, sizeof(d));
            d.x = s.rows;
            d.x_dtype = MV_DT_F16;
            d.ldx = 32 * F8;
            d.w_packed = tdnn.w;
            d.scale = tdnn_scale;
            d.shift = tdnn_shift;
            d.post_act = MV_ACT_RELU;
            d.y = s.xb[0];
            d.y_dtype = MV_DT_F16;
            d.ldy = blocks[0].c_out;
            d.B = B;
            d.T_in = T;
            d.T_out = T2;
            d.cin = 32 * F8;
            d.cout = cfg.init_channels;
            d.k = 5;
            d.dilation = 1;
            d.stride = 2;
            d.pad = 2;
            d.pad_mode = MV_PAD_ZERO;
            if ((rc = conv1d_launch(d, st))) return rc;
        }
        const int G = cfg.growth_rate;
        const char* fused_env = getenv("MV_CAMPP_FUSED");  // measurement knob: MV_CAMPP_FUSED=0 keeps the five launches per layer
        const bool fused_dense = !(fused_env != nullptr && fused_env[0] == '0');
        for (int bi = 0; bi < 3; ++bi) {
            const Block& Bk = blocks[bi];
            half_t* X = s.xb[bi];
            const int64_t ld = Bk.c_out;
            for (const DenseLayer& L : Bk.layers) {
                // utterances of up to 160 strided frames (3.2 s): the whole layer is one launch with the bottleneck kept in LDS
                if (fused_dense && cam_dense_layer_supported(T2, L.cin, bn_ch, G, Bk.dil, 100)) {
                    if ((rc = cam_dense_layer_launch(X, ld, B, T2, L.cin, L.lin1.w, L.bn1_s, L.bn1_t, L.bn2_s, L.bn2_t, L.local.w, L.wa, L.ba,
                                                     L.wb, L.bb, Bk.dil, 100, st)))
                        return rc;
                    continue;
                }
                MvConv1dDesc d;
                memset(&d, 0, sizeof(d));
                d.x = X;
                d.x_dtype = MV_DT_F16;
                d.ldx = ld;
                d.in_scale = L.bn1_s;
                d.in_shift = L.bn1_t;
                d.w_packed = L.lin1.w;
                d.scale = L.bn2_s;
                d.shift = L.bn2_t;
                d.post_act = MV_ACT_RELU;
                d.y = s.h;
                d.y_dtype = MV_DT_F16;
                d.ldy = bn_ch;
                d.B = B;
                d.T_in = d.T_out = T2;
                d.cin = L.cin;
                d.cout = bn_ch;
                d.k = 1;
                d.dilation = 1;
                d.stride = 1;
                d.pad = 0;
                d.pad_mode = MV_PAD_ZERO;
                if ((rc = conv1d_launch(d, st))) return rc;
                if ((rc = seg_mean_launch(s.h, bn_ch, B, T2, bn_ch, 100, s.ctx, st))) return rc;
                const int rows = B * s.nseg;
                if ((rc = linear_f32_launch(s.ctx, bn_ch, L.wa, bn_ch, L.ba, MV_ACT_RELU, s.g1, bn_ch / 2, rows, bn_ch, bn_ch / 2, 0, st)))
                    return rc;
                if ((rc = linear_f32_launch(s.g1, bn_ch / 2, L.wb, bn_ch / 2, L.bb, MV_ACT_SIGMOID, s.gate, G, rows, bn_ch / 2, G, 0, st)))
                    return rc;
                memset(&d, 0, sizeof(d));
                d.x = s.h;
                d.x_dtype = MV_DT_F16;
                d.ldx = bn_ch;
                d.w_packed = L.local.w;
                d.gate = s.gate;
                d.gate_seg_len = 100;
                d.y = X + L.cin;
                d.y_dtype = MV_DT_F16;
                d.ldy = ld;
                d.B = B;
                d.T_in = d.T_out = T2;
                d.cin = bn_ch;
                d.cout = G;
                d.k = 3;
                d.dilation = Bk.dil;
                d.stride = 1;
                d.pad = Bk.dil;
                d.pad_mode = MV_PAD_ZERO;
                if ((rc = conv1d_launch(d, st))) return rc;
            }
            // transit: BN + ReLU on load, 1x1 conv halves the channels into the next block's buffer
            MvConv1dDesc d;
            memset(&d, 0, sizeof(d));
            d.x = X;
            d.x_dtype = MV_DT_F16;
            d.ldx = ld;
            d.in_scale = Bk.tr_s;
            d.in_shift = Bk.tr_t;
            d.w_packed = Bk.transit.w;
            d.bias = Bk.transit.bias;
            d.y = bi < 2 ? s.xb[bi + 1] : s.last;
            d.y_dtype = MV_DT_F16;
            d.ldy = bi < 2 ? blocks[bi + 1].c_out : cfin;
            d.B = B;
            d.T_in = d.T_out = T2;
            d.cin = Bk.c_out;
            d.cout = Bk.c_out / 2;
            d.k = 1;
            d.dilation = 1;
            d.stride = 1;
            d.pad = 0;
            d.pad_mode = MV_PAD_ZERO;
            if ((rc = conv1d_launch(d, st))) return rc;
        }
        // out_nonlinear (BN+ReLU) + StatsPool (mean, unbiased std) in one reduction, then dense + BN(affine=False)
        if ((rc = time_stats_launch(s.last, cfin, B, T2, cfin, s.stats, s.stats + cfin, 2 * cfin, 1, 0.0f, st, out_s, out_t)))
            return rc;
        return linear_f32_launch(s.stats, 2 * cfin, dense_w, 2 * cfin, dense_b, MV_ACT_NONE, emb, cfg.embd_dim, B, 2 * cfin,
                                 cfg.embd_dim, 0, st);
    }
};

}  // namespace mv

extern "C" {

int mv_campp_create(const MvCamppCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_campp_create: null argument");
    mv::Weights w;
    int rc = w.init(tensors, num_tensors);
    if (rc != MV_OK) return rc;
    auto m = std::make_unique<mv::CamppModel>();
    rc = m->create(*cfg, w);
    if (rc != MV_OK) return rc;
    *out = reinterpret_cast<MvModel*>(static_cast<mv::MvModelBase*>(m.release()));
    return MV_OK;
}

}  // extern "C"
