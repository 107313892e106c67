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
//             (one launch per block / layer for utterances of up to 160 strided frames: camblock.hip, camdense.hip; two launches per layer over
//             160-frame chunks beyond: camdense.hip "long utterances"; five launches per layer only where neither form applies)
//             transit: 1x1 conv with BN/ReLU on load; out_nonlinear + StatsPool (unbiased std) fused into one reduction;
//             dense + BN(affine=False) folded into one fp32 linear layer.
//
// Head precision.  The fp16 head (fp16 maps and tap matrices, fp32 accumulation) reproduces trained-like checkpoints to 6e-8, but a head
// with large BatchNorm gains amplifies the 2^-11 operand rounding of its ~20 sites ~100 x (tests/budget_campp.py: the stress golden
// lands at 4e-4, no single site owning it).  So the handle also carries the head in an EXACT form -- the "fp32 head" of the interface: since
// round 4 the split-operand kernels of the ERes2Net family (conv2ds.hip: maps and weights as hi + lo fp16 pairs = 22 bits, three fp16 MFMA
// passes, the frequency-only stride through MvConv2dsDesc.stride_w; rounds 2-3: fp32 maps on v_mfma_f32_16x16x4_f32, conv2d.hip) -- and
// DECIDES AT CREATION which one it runs: both heads embed THREE fixed probe utterances (white uniform, white bell-shaped with the time mean
// removed, smooth voiced-like); if the embeddings of any probe differ by more than 5e-6 in 1 - cos the checkpoint is ill-conditioned for fp16
// maps and the handle takes the exact head (a multiple of the step time: bench leg config3_campp_fp32_head), otherwise the fp16 head.  How far such a figure is from the miss on a given input
// depends on the input (tests/budget_campp.py + profiles/r12_campp_head_probes.log: on the BatchNorm-calibrated family of the stress
// golden one white probe alone under-reads the miss on the test input by 2-40 x, the bell-shaped probe -- the test input's own statistics
// -- tracks it within 2 x), hence several probes with different statistics and the largest figure.
// MvCamppCfg.head_precision pins the head (measurements, tests, one numerics on every rank); mv_model_info reports the choice and the figures.
#include <array>
#include <cfloat>
#include <memory>

#include "kernels.h"
#include "model.h"
#include "s16map.h"

namespace mv {

struct CamppModel : MvModelBase {
    MvCamppCfg cfg;
    // head
    float* c1_w = nullptr;  // [32][9] fp32 (BN folded)
    float* c1_b = nullptr;
    half_t* c1_a = nullptr;  // the same weights as MFMA fragments (fcm_c1_pack): first conv inside the first block's kernel
    struct Conv2d {
        half_t* w = nullptr;  // [ntaps][32][32]
        float* bias = nullptr;
        int ntaps = 9;
        // the same conv for the fp32 head (conv2d.hip layout [32 co][9 taps][32 ci], BN folded) and its own BN shift; the block's
        // shortcut conv separately ([32][1][32] + its BN shift)
        // the exact head (split fp16 operands, conv2ds.hip): packed weights + output scales, fp32 biases
        half_t *w32 = nullptr, *sc_w32 = nullptr;
        float *b32 = nullptr, *sc_b32 = nullptr;
        float osc32 = 0.0f, sc_osc32 = 0.0f;
        std::vector<float> b32_host, sc_b32_host;   // unscaled (set_exact_gain uploads them times the head gain)
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
        MvCamLayerDesc* descs = nullptr;  // device array for cam_dense_block_kernel
    };
    Block blocks[3];
    float *out_s = nullptr, *out_t = nullptr;
    float *dense_w = nullptr, *dense_b = nullptr;
    int F8 = 0, bn_ch = 0, cfin = 0;
    // Range of the exact head.  S16 maps hold 64 * value in fp16 pairs and saturate at |value| = 1023.5 -- 64 x less than the fp16 head's maps, on
    // exactly the checkpoints (large BatchNorm gains) the exact head exists for.  The head is positively homogeneous (conv + folded BN + ReLU,
    // residual sums): run on gain * x with every bias times gain it produces gain * maps, exactly for a power of two.  create() measures the
    // largest stored value on the probe utterances (MvConv2dsDesc.peak) and picks gain = 2^-k so that it sits 16 x below the saturation point;
    // the first conv's weights / bias and the biases of the exact head carry the gain, the rows handed to the TDNN are multiplied by 2^k.
    int head_gain_k = 0;
    float *c1_w32 = nullptr, *c1_b32 = nullptr;   // head.conv1 (+ bn1) of the exact head: c1_w / c1_b times the gain
    std::vector<float> c1_w_host, c1_b_host;
    unsigned* d_peak = nullptr;     // device word: largest |64 * gain * value| an exact-head launch wanted to store (float bits; sticky, diagnostic only)
    float probe_peak = 0.0f;        // the largest head activation over the probe utterances (real units), what the gain was chosen from
    bool head_f32 = false;          // which head forward() runs (decided in create())
    float calibration = -1.0f;      // largest 1 - cos between the two heads over the probe utterances (-1: forced by MvCamppCfg.head_precision)
    float probe_calibration[3] = {-1.0f, -1.0f, -1.0f};
    static constexpr float CAMPP_HEAD_THRESHOLD = 5e-6f;

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
        {   // fp32 head: [co][tap][ci]
            std::vector<float> w32((size_t)32 * 9 * 32), scw;
            for (int co = 0; co < 32; ++co)
                for (int tap = 0; tap < 9; ++tap)
                    for (int ci = 0; ci < 32; ++ci) w32[((size_t)co * 9 + tap) * 32 + ci] = packed[((size_t)tap * 32 + co) * 32 + ci];
            auto upload_split = [&](const std::vector<float>& dense, int ks, half_t** dst, float* osc) {
                std::vector<half_t> sp((size_t)conv2ds_packed_floats(32, 32, ks) * 2);
                *osc = conv2ds_pack_host(dense.data(), 32, 32, ks, sp.data());
                *dst = static_cast<half_t*>(dev_alloc(sp.size() * sizeof(half_t)));
                return *dst != nullptr && hipMemcpy(*dst, sp.data(), sp.size() * sizeof(half_t), hipMemcpyHostToDevice) == hipSuccess;
            };
            if (!upload_split(w32, 3, &out->w32, &out->osc32)) return fail(MV_ERR_HIP, "campp create: upload failed");
            out->b32 = upload(t);
            out->b32_host = t;
            if (sc) {
                scw.assign(packed.begin() + (size_t)9 * 32 * 32, packed.end());  // [co][ci]
                if (!upload_split(scw, 1, &out->sc_w32, &out->sc_osc32)) return fail(MV_ERR_HIP, "campp create: upload failed");
                out->sc_b32 = upload(ts);
                out->sc_b32_host = ts;
            }
        }
        return out->bias && out->w32 && out->b32 ? MV_OK : fail(MV_ERR_HIP, "campp create: upload failed");
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
            c1_w32 = upload(W);
            c1_b32 = upload(t);
            c1_w_host = W;
            c1_b_host = t;
            d_peak = static_cast<unsigned*>(dev_alloc(sizeof(unsigned)));
            if (c1_w32 == nullptr || c1_b32 == nullptr || d_peak == nullptr) return fail(MV_ERR_HIP, "campp create: out of device memory");
            MV_HIP_OK(hipMemset(d_peak, 0, sizeof(unsigned)));
            std::vector<half_t> frag(2 * 64 * 8);
            fcm_c1_pack(W.data(), frag.data());
            c1_a = static_cast<half_t*>(dev_alloc(frag.size() * sizeof(half_t)));
            if (c1_a == nullptr) return fail(MV_ERR_HIP, "campp create: out of device memory");
            MV_HIP_OK(hipMemcpy(c1_a, frag.data(), frag.size() * sizeof(half_t), hipMemcpyHostToDevice));
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
            {   // the block's layers as one device array (camblock.hip)
                std::vector<MvCamLayerDesc> hd(B.layers.size());
                for (size_t li = 0; li < B.layers.size(); ++li) {
                    const DenseLayer& L = B.layers[li];
                    hd[li] = MvCamLayerDesc{L.lin1.w, L.bn1_s, L.bn1_t, L.bn2_s, L.bn2_t, L.local.w, L.wa, L.ba, L.wb, L.bb, L.cin, conv1d_cin_pad(L.cin)};
                }
                B.descs = static_cast<MvCamLayerDesc*>(dev_alloc(hd.size() * sizeof(MvCamLayerDesc)));
                if (B.descs == nullptr) return fail(MV_ERR_HIP, "campp create: out of device memory");
                MV_HIP_OK(hipMemcpy(B.descs, hd.data(), hd.size() * sizeof(MvCamLayerDesc), hipMemcpyHostToDevice));
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
        if ((rc = choose_head())) return rc;
        return c.xvector_probe == MV_CAMPP_XVEC_PROBE_OFF ? MV_OK : probe_xvector(w);
    }

    int info(int key, float* value) const override {
        if (key == MV_INFO_CAMPP_HEAD_F32) {
            *value = head_f32 ? 1.0f : 0.0f;
            return MV_OK;
        }
        if (key == MV_INFO_CAMPP_CALIBRATION) {
            *value = calibration;
            return MV_OK;
        }
        if (key >= MV_INFO_CAMPP_PROBE0 && key < MV_INFO_CAMPP_PROBE0 + 3) {
            *value = probe_calibration[key - MV_INFO_CAMPP_PROBE0];
            return MV_OK;
        }
        if (key == MV_INFO_CAMPP_HEAD_GAIN_LOG2) {
            *value = (float)-head_gain_k;
            return MV_OK;
        }
        if (key == MV_INFO_CAMPP_PROBE_PEAK) {
            *value = probe_peak;
            return MV_OK;
        }
        if (key == MV_INFO_CAMPP_XVEC_SENSITIVITY) {
            *value = xvec_sensitivity;
            return MV_OK;
        }
        if (key >= MV_INFO_CAMPP_XVEC_PROBE0 && key < MV_INFO_CAMPP_XVEC_PROBE0 + NPROBE) {
            *value = xvec_probe[key - MV_INFO_CAMPP_XVEC_PROBE0];
            return MV_OK;
        }
        if (key == MV_INFO_CAMPP_HEAD_PEAK || key == MV_INFO_CAMPP_HEAD_SATURATED) {   // (waits for the device: a diagnostic, not a hot-path call)
            float v = 0.0f;
            MV_HIP_OK(hipDeviceSynchronize());
            if (read_peak(&v) != MV_OK) return MV_ERR_HIP;
            *value = key == MV_INFO_CAMPP_HEAD_PEAK ? ldexpf(v, head_gain_k) / CS_XSCALE : (v >= 65504.0f ? 1.0f : 0.0f);
            return MV_OK;
        }
        return MvModelBase::info(key, value);
    }

    int read_peak(float* v) const {   // largest scaled value the exact head wanted to store since the word was last cleared
        unsigned bits = 0;
        MV_HIP_OK(hipMemcpy(&bits, d_peak, sizeof(bits), hipMemcpyDeviceToHost));
        *v = __builtin_bit_cast(float, bits);
        return MV_OK;
    }

    // the exact head's first conv and biases times gain = 2^-k (see head_gain_k)
    int set_exact_gain(int k) {
        head_gain_k = k;
        const float g = ldexpf(1.0f, -k);
        auto put = [&](float* dst, const std::vector<float>& src) {
            if (dst == nullptr || src.empty()) return MV_OK;
            std::vector<float> v(src);
            for (float& x : v) x *= g;
            MV_HIP_OK(hipMemcpy(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
            return MV_OK;
        };
        int rc;
        if ((rc = put(c1_w32, c1_w_host)) || (rc = put(c1_b32, c1_b_host))) return rc;
        for (int i = 0; i < 4; ++i)
            if ((rc = put(res[i].conv1.b32, res[i].conv1.b32_host)) || (rc = put(res[i].conv2.b32, res[i].conv2.b32_host)) ||
                (rc = put(res[i].conv2.sc_b32, res[i].conv2.sc_b32_host)))
                return rc;
        return put(head_out.b32, head_out.b32_host);
    }

    // Three fixed probe utterances: they set the exact head's gain (largest stored value 16 x below the S16 saturation point) and, through both
    // heads, the largest 1 - cos between the two embeddings decides which head forward() runs (header comment).
    int choose_head() {
        if (cfg.head_precision == MV_CAMPP_HEAD_F16) {
            head_f32 = false;
            return MV_OK;
        }
        const int F = cfg.input_size, D = cfg.embd_dim;
        std::vector<float> feats[NPROBE];
        make_probes(feats);
        size_t wsb = 0;
        for (int p = 0; p < NPROBE; ++p) {
            const size_t b16 = carve(nullptr, 1, probe_T[p], false).bytes, b32 = carve(nullptr, 1, probe_T[p], true).bytes;
            wsb = std::max(wsb, std::max(b16, b32));
        }
        return choose_head_on(feats, wsb, F, D);
    }

    static constexpr int NPROBE = 3;
    static constexpr int probe_T[NPROBE] = {96, 150, 200};
    void make_probes(std::vector<float> (&feats)[NPROBE]) const {
        const int F = cfg.input_size;
        uint32_t lcg = 0x2545F491u;
        auto uni = [&]() {  // uniform in [-0.5, 0.5)
            lcg = lcg * 1664525u + 1013904223u;
            return (float)(lcg >> 8) / 16777216.0f - 0.5f;
        };
        // probe 0: white, uniform in [-3, 3) -- the scale of mean-normalised log-mel features
        feats[0].resize((size_t)probe_T[0] * F);
        for (float& v : feats[0]) v = uni() * 6.0f;
        // probe 1: white, bell-shaped (sum of four uniforms), standard deviation 2, time mean removed -- what a noise-like recording gives
        feats[1].resize((size_t)probe_T[1] * F);
        for (float& v : feats[1]) v = (uni() + uni() + uni() + uni()) * (2.0f / 0.57735f);
        // probe 2: smooth in time and frequency like voiced audio -- drifting harmonics, a spectral tilt, a voiced / silent alternation, a
        // little noise; standard deviation ~2, time mean removed
        feats[2].resize((size_t)probe_T[2] * F);
        for (int t = 0; t < probe_T[2]; ++t)
            for (int m = 0; m < F; ++m) {
                const double tw = 6.283185307179586;
                const double v = 3.0 * sin(tw * (t / 37.0 + m / 23.0)) + 2.0 * sin(tw * t / 9.3) * ((double)m / F) + 4.0 * (0.5 * F - m) / F +
                                 (sin(tw * t / 61.0) > 0.0 ? 3.0 : 0.0) + 1.4 * uni();
                feats[2][(size_t)t * F + m] = (float)(0.7 * v);
            }
        for (int p = 1; p < NPROBE; ++p)  // AudioFeaturizer's time-mean subtraction (featurizer.py:79)
            for (int m = 0; m < F; ++m) {
                double s = 0.0;
                for (int t = 0; t < probe_T[p]; ++t) s += feats[p][(size_t)t * F + m];
                const float mean = (float)(s / probe_T[p]);
                for (int t = 0; t < probe_T[p]; ++t) feats[p][(size_t)t * F + m] -= mean;
            }
    }

    int choose_head_on(std::vector<float> (&feats)[NPROBE], size_t wsb, int F, int D) {
        struct Scratch {  // freed on every path out of this function
            float *dfe = nullptr, *demb = nullptr;
            void* ws = nullptr;
            ~Scratch() {
                hipFree(dfe);
                hipFree(demb);
                hipFree(ws);
            }
        } sc;
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.dfe), (size_t)probe_T[NPROBE - 1] * F * sizeof(float)));
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.demb), (size_t)2 * D * sizeof(float)));
        MV_HIP_OK(hipMalloc(&sc.ws, wsb));
        // ---- the exact head's gain: the probes' largest stored value, 16 x below the saturation point (the measurement repeats at the new gain: a map
        //      clamped upstream makes everything downstream of it read low) ----
        for (int round = 0; round < 8; ++round) {
            MV_HIP_OK(hipMemset(d_peak, 0, sizeof(unsigned)));
            for (int p = 0; p < NPROBE; ++p) {
                MV_HIP_OK(hipMemcpy(sc.dfe, feats[p].data(), feats[p].size() * sizeof(float), hipMemcpyHostToDevice));
                const int rc = forward_impl(sc.dfe, 1, probe_T[p], sc.demb, sc.ws, wsb, nullptr, true);
                if (rc != MV_OK) return rc;
            }
            MV_HIP_OK(hipDeviceSynchronize());
            float v = 0.0f;
            if (read_peak(&v) != MV_OK) return MV_ERR_HIP;
            probe_peak = ldexpf(v, head_gain_k) / CS_XSCALE;
            if (!(v > 65504.0f / 16.0f) || head_gain_k >= 96) break;   // (an infinite or NaN peak: nothing a gain repairs)
            int e2 = 0;
            frexpf(v / (65504.0f / 16.0f), &e2);   // v / limit = f * 2^e2, f in [0.5, 1): 2^-e2 brings it to or below the limit
            const int step = std::isfinite(v) ? (e2 > 0 ? e2 : 1) : 16;
            const int rc = set_exact_gain(head_gain_k + step);
            if (rc != MV_OK) return rc;
        }
        MV_HIP_OK(hipMemset(d_peak, 0, sizeof(unsigned)));   // from here on the word reports what real inputs do
        if (cfg.head_precision == MV_CAMPP_HEAD_F32) {
            head_f32 = true;
            return MV_OK;
        }
        float worst = 0.0f;
        std::vector<float> e((size_t)2 * D);
        for (int p = 0; p < NPROBE; ++p) {
            const int T = probe_T[p];
            MV_HIP_OK(hipMemcpy(sc.dfe, feats[p].data(), feats[p].size() * sizeof(float), hipMemcpyHostToDevice));
            int rc = forward_impl(sc.dfe, 1, T, sc.demb, sc.ws, wsb, nullptr, false);
            if (rc == MV_OK) rc = forward_impl(sc.dfe, 1, T, sc.demb + D, sc.ws, wsb, nullptr, true);
            if (rc != MV_OK) return rc;
            MV_HIP_OK(hipMemcpy(e.data(), sc.demb, e.size() * sizeof(float), hipMemcpyDeviceToHost));
            double ab = 0.0, aa = 0.0, bb = 0.0;
            for (int i = 0; i < D; ++i) {
                ab += (double)e[i] * e[i + D];
                aa += (double)e[i] * e[i];
                bb += (double)e[D + i] * e[D + i];
            }
            const double cosv = (aa > 0.0 && bb > 0.0) ? ab / sqrt(aa * bb) : 0.0;
            const float c = (float)(1.0 - cosv);
            probe_calibration[p] = c;
            if (!(c <= worst)) worst = c;  // (a NaN from a non-finite fp16-head embedding sticks)
        }
        calibration = worst;
        head_f32 = !(calibration <= CAMPP_HEAD_THRESHOLD);  // also taken when the fp16 head produced a non-finite embedding
        MV_HIP_OK(hipMemset(d_peak, 0, sizeof(unsigned)));
        return MV_OK;
    }


    // ---- x-vector sensitivity (round 6) -----------------------------------------------------------------------------------
    // The head probes above compare the two FCM HEADS; a checkpoint whose embedding is sensitive to the 11-bit operands of the x-vector part
    // (`campp_c64` seed 31: 3.3e-4 from the fp16 rounding of the x-vector weights alone) passes them unnoticed, and there is no
    // exact form of that part to fall back to.  So create() MEASURES it: the x-vector part of the three probe utterances once more in exact fp32
    // -- on the very rows the chosen head produced, the caller's fp32 weights in the reference layout as they are, every GEMM on the exact-fp32
    // linear kernel (linear.hip, fp32 MFMA), everything else in host loops (a probe, not a path: ~330 small launches and copies, once) -- and
    // 1 - cos against the shipped path's embedding of the same probe = what fp16 weights / stored tensors / pre-activation operands of the
    // x-vector part cost on that input.  mv_model_info keys MV_INFO_CAMPP_XVEC_PROBE0 + p and MV_INFO_CAMPP_XVEC_SENSITIVITY (the largest) report
    // it; mvector/models/campplus.py warns when it exceeds MV_CAMPP_XVEC_WARN.  Reference arithmetic: campplus.py:94-181,186-189,27-33.
    float xvec_probe[NPROBE] = {-1.0f, -1.0f, -1.0f};
    float xvec_sensitivity = -1.0f;   // -1: not measured (MvCamppCfg.xvector_probe = MV_CAMPP_XVEC_PROBE_OFF)

    struct ExactScratch {
        float *a = nullptr, *y = nullptr;
        ~ExactScratch() {
            hipFree(a);
            hipFree(y);
        }
    };

    // rows [T][32 * F8] in OUR channel order f * 32 + c (the chosen head's fp16 output as floats) -> emb [D], exact fp32
    int xvector_exact(const Weights& w, const std::vector<float>& rows, int T, std::vector<float>& emb) const {
        const int cin0 = 32 * F8, G = cfg.growth_rate, D = cfg.embd_dim;
        const int T2 = (T - 1) / 2 + 1;
        ExactScratch sc;
        const size_t na = (size_t)T2 * std::max(5 * cin0, 1024 + 32 * 24), ny = (size_t)T2 * 1024;
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.a), na * sizeof(float)));
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.y), ny * sizeof(float)));
        auto gemm = [&](const std::vector<float>& A, int n, int K, const float* Wdev, const float* bias_dev, int O, std::vector<float>& Y) -> int {
            if ((size_t)n * K > na || (size_t)n * O > ny) return fail(MV_ERR_UNSUPPORTED, "campp x-vector probe: scratch too small");
            MV_HIP_OK(hipMemcpy(sc.a, A.data(), (size_t)n * K * sizeof(float), hipMemcpyHostToDevice));
            const int rc = linear_f32_launch(sc.a, K, Wdev, K, bias_dev, MV_ACT_NONE, sc.y, O, n, K, O, 0, nullptr);
            if (rc != MV_OK) return rc;
            Y.resize((size_t)n * O);
            MV_HIP_OK(hipMemcpy(Y.data(), sc.y, Y.size() * sizeof(float), hipMemcpyDeviceToHost));
            return MV_OK;
        };
        int rc;
        std::vector<float> s, t, A, Y;
        // tdnn: k = 5, stride 2, zero padding 2 on the reference's channel order c * F8 + f, then BN, ReLU (campplus.py:319-326)
        const float* Wd = nullptr;
        if ((rc = w.dev("xvector.tdnn.linear.weight", (int64_t)cfg.init_channels * cin0 * 5, &Wd))) return rc;
        A.assign((size_t)T2 * cin0 * 5, 0.0f);
        for (int t2 = 0; t2 < T2; ++t2)
            for (int j = 0; j < 5; ++j) {
                const int ti = 2 * t2 + j - 2;
                if (ti < 0 || ti >= T) continue;
                for (int c = 0; c < 32; ++c)
                    for (int f = 0; f < F8; ++f) A[((size_t)t2 * cin0 + (c * F8 + f)) * 5 + j] = rows[(size_t)ti * cin0 + f * 32 + c];
            }
        if ((rc = gemm(A, T2, cin0 * 5, Wd, nullptr, cfg.init_channels, Y))) return rc;
        if ((rc = fold_bn(w, "xvector.tdnn.nonlinear.batchnorm", cfg.init_channels, s, t, 1e-5f))) return rc;
        std::vector<float> X, Xn;
        int ld = blocks[0].c_out;
        X.assign((size_t)T2 * ld, 0.0f);
        for (int i = 0; i < T2; ++i)
            for (int c = 0; c < cfg.init_channels; ++c) X[(size_t)i * ld + c] = std::max(0.0f, Y[(size_t)i * cfg.init_channels + c] * s[c] + t[c]);
        const int nseg = (T2 + 99) / 100;
        for (int bi = 0; bi < 3; ++bi) {
            const Block& Bk = blocks[bi];
            ld = Bk.c_out;
            for (size_t li = 0; li < Bk.layers.size(); ++li) {
                const int cin = Bk.layers[li].cin;
                const std::string p = "xvector.block" + std::to_string(bi + 1) + ".tdnnd" + std::to_string(li + 1);
                std::vector<float> s2, t2v, wa, ba, wb, bb, H, h((size_t)T2 * bn_ch);
                if ((rc = fold_bn(w, p + ".nonlinear1.batchnorm", cin, s, t, 1e-5f)) || (rc = fold_bn(w, p + ".nonlinear2.batchnorm", bn_ch, s2, t2v, 1e-5f)))
                    return rc;
                A.resize((size_t)T2 * cin);
                for (int i = 0; i < T2; ++i)
                    for (int c = 0; c < cin; ++c) A[(size_t)i * cin + c] = std::max(0.0f, X[(size_t)i * ld + c] * s[c] + t[c]);
                if ((rc = w.dev(p + ".linear1.weight", (int64_t)bn_ch * cin, &Wd)) || (rc = gemm(A, T2, cin, Wd, nullptr, bn_ch, H))) return rc;
                for (int i = 0; i < T2; ++i)
                    for (int c = 0; c < bn_ch; ++c) h[(size_t)i * bn_ch + c] = std::max(0.0f, H[(size_t)i * bn_ch + c] * s2[c] + t2v[c]);
                // context = mean over time + mean over the frame's 100-frame segment (ceil mode: the last segment averages its own frames)
                if ((rc = w.host(p + ".cam_layer.linear1.weight", (int64_t)(bn_ch / 2) * bn_ch, wa)) || (rc = w.host(p + ".cam_layer.linear1.bias", bn_ch / 2, ba)) ||
                    (rc = w.host(p + ".cam_layer.linear2.weight", (int64_t)G * (bn_ch / 2), wb)) || (rc = w.host(p + ".cam_layer.linear2.bias", G, bb)))
                    return rc;
                std::vector<double> gmean(bn_ch, 0.0);
                for (int i = 0; i < T2; ++i)
                    for (int c = 0; c < bn_ch; ++c) gmean[c] += h[(size_t)i * bn_ch + c];
                std::vector<float> gate((size_t)nseg * G);
                for (int sg = 0; sg < nseg; ++sg) {
                    const int lo = sg * 100, hi = std::min(T2, lo + 100);
                    std::vector<double> ctx(bn_ch, 0.0), g1(bn_ch / 2);
                    for (int i = lo; i < hi; ++i)
                        for (int c = 0; c < bn_ch; ++c) ctx[c] += h[(size_t)i * bn_ch + c];
                    for (int c = 0; c < bn_ch; ++c) ctx[c] = (double)((float)(ctx[c] / (hi - lo)) + (float)(gmean[c] / T2));
                    for (int o = 0; o < bn_ch / 2; ++o) {
                        double acc = ba[o];
                        for (int c = 0; c < bn_ch; ++c) acc += (double)wa[(size_t)o * bn_ch + c] * ctx[c];
                        g1[o] = acc > 0.0 ? acc : 0.0;
                    }
                    for (int o = 0; o < G; ++o) {
                        double acc = bb[o];
                        for (int c = 0; c < bn_ch / 2; ++c) acc += (double)wb[(size_t)o * (bn_ch / 2) + c] * g1[c];
                        gate[(size_t)sg * G + o] = (float)(1.0 / (1.0 + exp(-acc)));
                    }
                }
                // k = 3 dilated conv, zero padding: columns ci * 3 + j as the reference weight [G][bn_ch][3] has them
                A.assign((size_t)T2 * bn_ch * 3, 0.0f);
                for (int i = 0; i < T2; ++i)
                    for (int j = 0; j < 3; ++j) {
                        const int ti = i + (j - 1) * Bk.dil;
                        if (ti < 0 || ti >= T2) continue;
                        for (int c = 0; c < bn_ch; ++c) A[((size_t)i * bn_ch + c) * 3 + j] = h[(size_t)ti * bn_ch + c];
                    }
                if ((rc = w.dev(p + ".cam_layer.linear_local.weight", (int64_t)G * bn_ch * 3, &Wd)) || (rc = gemm(A, T2, bn_ch * 3, Wd, nullptr, G, Y))) return rc;
                for (int i = 0; i < T2; ++i)
                    for (int o = 0; o < G; ++o) X[(size_t)i * ld + cin + o] = Y[(size_t)i * G + o] * gate[(size_t)(i / 100) * G + o];
            }
            const std::string tp = "xvector.transit" + std::to_string(bi + 1);
            if ((rc = fold_bn(w, tp + ".nonlinear.batchnorm", ld, s, t, 1e-5f))) return rc;
            A.resize((size_t)T2 * ld);
            for (int i = 0; i < T2; ++i)
                for (int c = 0; c < ld; ++c) A[(size_t)i * ld + c] = std::max(0.0f, X[(size_t)i * ld + c] * s[c] + t[c]);
            const float* bias_dev = nullptr;
            if (w.has(tp + ".linear.bias") && (rc = w.dev(tp + ".linear.bias", ld / 2, &bias_dev))) return rc;
            if ((rc = w.dev(tp + ".linear.weight", (int64_t)(ld / 2) * ld, &Wd)) || (rc = gemm(A, T2, ld, Wd, bias_dev, ld / 2, Y))) return rc;
            const int ldn = bi < 2 ? blocks[bi + 1].c_out : cfin;
            Xn.assign((size_t)T2 * ldn, 0.0f);
            for (int i = 0; i < T2; ++i)
                for (int c = 0; c < ld / 2; ++c) Xn[(size_t)i * ldn + c] = Y[(size_t)i * (ld / 2) + c];
            X.swap(Xn);
        }
        // out_nonlinear (BN, ReLU) -> mean | unbiased std over time -> dense + BN(affine = False) (folded at create: dense_w / dense_b, exact fp32)
        if ((rc = fold_bn(w, "xvector.out_nonlinear.batchnorm", cfin, s, t, 1e-5f))) return rc;
        std::vector<float> stats((size_t)2 * cfin);
        for (int c = 0; c < cfin; ++c) {
            double m = 0.0, q = 0.0;
            for (int i = 0; i < T2; ++i) m += std::max(0.0f, X[(size_t)i * cfin + c] * s[c] + t[c]);
            m /= T2;
            for (int i = 0; i < T2; ++i) {
                const double d = (double)std::max(0.0f, X[(size_t)i * cfin + c] * s[c] + t[c]) - m;
                q += d * d;
            }
            stats[c] = (float)m;
            stats[cfin + c] = (float)sqrt(q / (T2 - 1));
        }
        MV_HIP_OK(hipMemcpy(sc.a, stats.data(), stats.size() * sizeof(float), hipMemcpyHostToDevice));
        if ((rc = linear_f32_launch(sc.a, 2 * cfin, dense_w, 2 * cfin, dense_b, MV_ACT_NONE, sc.y, D, 1, 2 * cfin, D, 0, nullptr))) return rc;
        emb.resize(D);
        MV_HIP_OK(hipMemcpy(emb.data(), sc.y, (size_t)D * sizeof(float), hipMemcpyDeviceToHost));
        return MV_OK;
    }

    int probe_xvector(const Weights& w) {
        const int F = cfg.input_size, D = cfg.embd_dim, cin0 = 32 * F8;
        std::vector<float> feats[NPROBE];
        make_probes(feats);
        struct Scratch {
            float *dfe = nullptr, *demb = nullptr;
            void* ws = nullptr;
            ~Scratch() {
                hipFree(dfe);
                hipFree(demb);
                hipFree(ws);
            }
        } sc;
        size_t wsb = 0;
        for (int p = 0; p < NPROBE; ++p) wsb = std::max(wsb, carve(nullptr, 1, probe_T[p], head_f32).bytes);
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.dfe), (size_t)probe_T[NPROBE - 1] * F * sizeof(float)));
        MV_HIP_OK(hipMalloc(reinterpret_cast<void**>(&sc.demb), (size_t)D * sizeof(float)));
        MV_HIP_OK(hipMalloc(&sc.ws, wsb));
        float worst = 0.0f;
        for (int p = 0; p < NPROBE; ++p) {
            const int T = probe_T[p];
            MV_HIP_OK(hipMemcpy(sc.dfe, feats[p].data(), feats[p].size() * sizeof(float), hipMemcpyHostToDevice));
            int rc = forward_impl(sc.dfe, 1, T, sc.demb, sc.ws, wsb, nullptr, head_f32);
            if (rc != MV_OK) return rc;
            std::vector<float> e(D), ex;
            std::vector<half_t> rows16((size_t)T * cin0);
            MV_HIP_OK(hipMemcpy(e.data(), sc.demb, (size_t)D * sizeof(float), hipMemcpyDeviceToHost));
            MV_HIP_OK(hipMemcpy(rows16.data(), carve(sc.ws, 1, T, head_f32).rows, rows16.size() * sizeof(half_t), hipMemcpyDeviceToHost));
            std::vector<float> rows(rows16.size());
            for (size_t i = 0; i < rows.size(); ++i) rows[i] = (float)rows16[i];
            if ((rc = xvector_exact(w, rows, T, ex))) return rc;
            double ab = 0.0, aa = 0.0, bb = 0.0;
            for (int i = 0; i < D; ++i) {
                ab += (double)e[i] * ex[i];
                aa += (double)e[i] * e[i];
                bb += (double)ex[i] * ex[i];
            }
            const float c = (float)(1.0 - ((aa > 0.0 && bb > 0.0) ? ab / sqrt(aa * bb) : 0.0));
            xvec_probe[p] = c;
            if (!(c <= worst)) worst = c;
        }
        xvec_sensitivity = worst;
        MV_HIP_OK(hipMemset(d_peak, 0, sizeof(unsigned)));   // (the probes are not the caller's inputs)
        return MV_OK;
    }

    struct Ws {
        float *f0, *fa, *fb, *fc;  // fp32 head: the full-resolution map and three half-resolution ones
        half_t *m0, *m1, *m2;  // FCM maps (ping-pong)
        half_t* rows;          // [B, T, 32*F8]
        half_t* xb[3];         // dense-block buffers [B, T2, c_out]
        half_t* last;          // [B, T2, cfin]
        half_t* h;             // [B, T2, 128]
        half_t* act;           // [B, T2, widest block]: pre-activated transit input
        float *ctx, *g1, *gate, *stats;
        float* hpart;          // long utterances: time sums of h per (utterance, 160-frame chunk, segment inside the chunk)
        size_t bytes;
        int T2, nseg;
    };

    Ws carve(void* base, int B, int T, bool f32) const {
        Carver c(base);
        Ws s;
        const int F = cfg.input_size;
        s.T2 = (T - 1) / 2 + 1;
        s.nseg = (s.T2 + 99) / 100;
        const size_t full = (size_t)B * F * T * 32;
        const size_t half = (size_t)B * ((F + 1) / 2) * T * 32;
        s.f0 = s.fa = s.fb = s.fc = nullptr;
        s.m0 = s.m1 = s.m2 = nullptr;
        if (f32) {
            s.f0 = c.take<float>(full);
            s.fa = c.take<float>(half);
            s.fb = c.take<float>(half);
            s.fc = c.take<float>(half);
        } else {
            s.m0 = c.take<half_t>(full);
            s.m1 = c.take<half_t>(half);
            s.m2 = c.take<half_t>(half);
        }
        s.rows = c.take<half_t>((size_t)B * T * 32 * F8);
        const size_t N2 = (size_t)B * s.T2;
        for (int i = 0; i < 3; ++i) s.xb[i] = c.take<half_t>(N2 * blocks[i].c_out);
        s.last = c.take<half_t>(N2 * cfin);
        s.h = c.take<half_t>(N2 * bn_ch);
        {
            size_t widest = 0;
            for (int i = 0; i < 3; ++i) widest = (size_t)blocks[i].c_out > widest ? (size_t)blocks[i].c_out : widest;
            s.act = c.take<half_t>(N2 * widest);
        }
        s.ctx = c.take<float>((size_t)B * s.nseg * bn_ch);
        s.hpart = c.take<float>((size_t)cam_dense_long_part_floats(B, s.T2));
        s.g1 = c.take<float>((size_t)B * s.nseg * (bn_ch / 2));
        s.gate = c.take<float>((size_t)B * s.nseg * cfg.growth_rate);
        s.stats = c.take<float>((size_t)B * 2 * cfin);
        s.bytes = c.total();
        return s;
    }

    int workspace_bytes(int B, int T, size_t* bytes) const override {
        MV_REQUIRE(B > 0 && T > 0 && bytes != nullptr, "workspace_bytes: bad argument");
        *bytes = carve(nullptr, B, T, head_f32).bytes;
        return MV_OK;
    }

    int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const override {
        return forward_impl(feats, B, T, emb, ws, ws_bytes, st, head_f32);
    }

    // exact FCM head: S16 maps [B, F, T, 32] (hi + lo fp16 pairs, 4 bytes per channel) through the split-operand conv2ds kernels (header comment):
    // s.rows <- fp16 [B, T, F8, 32]
    int head_fp32(const float* feats, int B, int T, const Ws& s, hipStream_t st) const {
        const int F = cfg.input_size;
        int rc;
        if ((rc = conv2d_first_s16_launch(feats, reinterpret_cast<half_t*>(s.f0), c1_w32, c1_b32, B, T, F, 32, st, d_peak))) return rc;
        auto conv = [&](const float* x, int H, const half_t* w, float osc, const float* bias, int ks, int stride, const float* res, bool relu, float* y) {
            MvConv2dsDesc d{};
            d.x = x;
            d.ldx = 32;
            d.w = w;
            d.oscale = osc;
            d.bias = bias;
            d.res = res;
            d.ldres = 32;
            d.y = y;
            d.ldy = 32;
            d.B = B;
            d.H = H;
            d.W = T;
            d.cin16 = d.cout16 = 32;
            d.ks = ks;
            d.stride = stride;
            d.stride_w = 1;
            d.epi = 0;
            d.lo = relu ? 0.0f : -FLT_MAX;
            d.hi = FLT_MAX;
            d.peak = d_peak;
            return conv2ds_launch(d, st);
        };
        const float* cur = s.f0;
        int Fc = F;
        for (int i = 0; i < 4; ++i) {
            const ResBlock& r = res[i];
            const int Fo = (Fc - 1) / r.stride + 1;
            // buffers: block 0 f0 -> fc, block 1 fc -> fb, block 2 fb -> f0, block 3 f0 -> fb (conv1 output always in fa; shortcut of
            // block 0 in fb, of block 2 in fc)
            float* out = i == 0 ? s.fc : (i == 2 ? s.f0 : s.fb);
            float* scb = i == 0 ? s.fb : s.fc;
            if ((rc = conv(cur, Fc, r.conv1.w32, r.conv1.osc32, r.conv1.b32, 3, r.stride, nullptr, true, s.fa))) return rc;
            const float* resid = cur;
            if (r.has_shortcut) {
                if ((rc = conv(cur, Fc, r.conv2.sc_w32, r.conv2.sc_osc32, r.conv2.sc_b32, 1, r.stride, nullptr, false, scb))) return rc;
                resid = scb;
            }
            if ((rc = conv(s.fa, Fo, r.conv2.w32, r.conv2.osc32, r.conv2.b32, 3, 1, resid, true, out))) return rc;
            cur = out;
            Fc = Fo;
        }
        MV_REQUIRE((Fc - 1) / 2 + 1 == F8, "campp forward: unexpected frequency size after the head");
        if ((rc = conv(cur, Fc, head_out.w32, head_out.osc32, head_out.b32, 3, 2, nullptr, true, s.fa))) return rc;
        return fcm_rows_from_s16_launch(reinterpret_cast<const half_t*>(s.fa), s.rows, B, T, F8, st, ldexpf(1.0f, head_gain_k));
    }

    int forward_impl(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st, bool f32) const {
        MV_REQUIRE(feats != nullptr && emb != nullptr && ws != nullptr, "campp forward: null buffer");
        MV_REQUIRE(B > 0 && T >= 3, "campp forward: needs at least 3 frames (unbiased std over the strided time axis)");
        const Ws s = carve(ws, B, T, f32);
        if (s.bytes > ws_bytes) return fail(MV_ERR_WORKSPACE, "campp forward: workspace too small");
        const int F = cfg.input_size;
        int rc;
        // ---- head ----
        if (f32) {
            if ((rc = head_fp32(feats, B, T, s, st))) return rc;
        } else {
        auto plain = [&](int Fd) { return std::array<int64_t, 3>{(int64_t)Fd * T * 32, (int64_t)T * 32, 32}; };
        const half_t* cur = s.m0;
        int Fc = F;
        half_t* pp[2] = {s.m1, s.m2};
        // head.conv1 is evaluated inside the first block's kernel (no [B, F, T, 32] image of the features in HBM) whenever the block has the
        // reference's shape (strided, with its shortcut conv); fcm_conv1_kernel (fp32 weights on the vector pipe) covers the rest
        const bool c1_inside = res[0].stride == 2 && res[0].has_shortcut && F >= 3 && (int64_t)B * T * F < ((int64_t)1 << 31);
        if (!c1_inside)
            if ((rc = fcm_conv1_launch(feats, s.m0, c1_w, c1_b, B, T, F, st))) return rc;
        for (int i = 0; i < 4; ++i) {
            // one launch per BasicResBlock (fcmblock.hip): x read once, mid map in LDS, output written once
            const ResBlock& r = res[i];
            const int Fo = (Fc - 1) / r.stride + 1;
            auto so = plain(Fo);
            half_t* t2 = (cur == pp[0]) ? pp[1] : pp[0];
            if (!fcm_block_supported(t2, so[0], so[1], so[2], T, Fc))
                return fail(MV_ERR_UNSUPPORTED, "campp forward: utterance too long for the FCM head (T * 32 * F must stay below 2^31 elements)");
            const bool c1 = i == 0 && c1_inside;  // the block's input rows are made from the features inside the kernel
            if ((rc = fcm_block_launch(c1 ? nullptr : cur, Fc, r.stride, r.conv1.w, r.conv1.bias, r.conv2.w, r.conv2.bias, r.has_shortcut ? 1 : 0,
                                       t2, so[0], so[1], so[2], B, T, st, c1 ? feats : nullptr, c1_a, c1_b)))
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
        }  // fp16 head
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
        for (int bi = 0; bi < 3; ++bi) {
            const Block& Bk = blocks[bi];
            half_t* X = s.xb[bi];
            const int64_t ld = Bk.c_out;
            // all layers of the block in one launch, the next layer's first operands requested under the current layer's tail (camblock.hip);
            // geometries it does not take fall through to one launch per layer, long utterances to two, anything else to five
            const bool block_kernel = cam_dense_block_supported(T2, Bk.c_in, Bk.c_out, bn_ch, G, Bk.dil, 100);
            if (block_kernel) {
                if ((rc = cam_dense_block_launch(X, ld, B, T2, Bk.descs, (int)Bk.layers.size(), Bk.dil, 100, st))) return rc;
            }
            for (const DenseLayer& L : Bk.layers) {
                if (block_kernel) break;
                // utterances of up to 160 strided frames (3.2 s): the whole layer is one launch with the bottleneck kept in LDS
                if (cam_dense_layer_supported(T2, L.cin, bn_ch, G, Bk.dil, 100)) {
                    if ((rc = cam_dense_layer_launch(X, ld, B, T2, L.cin, L.lin1.w, L.bn1_s, L.bn1_t, L.bn2_s, L.bn2_t, L.local.w, L.wa, L.ba,
                                                     L.wb, L.bb, Bk.dil, 100, st)))
                        return rc;
                    continue;
                }
                // longer utterances: two launches per layer over chunks of 160 strided frames (camdense.hip, "long utterances")
                if (cam_dense_long_supported(T2, L.cin, bn_ch, G, Bk.dil, 100)) {
                    if ((rc = cam_dense_long_launch(X, ld, B, T2, L.cin, L.lin1.w, L.bn1_s, L.bn1_t, L.bn2_s, L.bn2_t, L.local.w, L.wa, L.ba, L.wb, L.bb,
                                                    Bk.dil, 100, s.h, s.hpart, st)))
                        return rc;
                    continue;
                }
                // every other geometry (bottleneck / growth widths the fused kernels are not built for): five launches
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
            // transit: BN + ReLU, then a 1x1 conv that halves the channels into the next block's buffer.  The pre-activation is written out once
            // (bn_relu_rows_kernel) and the conv runs on the direct global -> LDS path (ring kernel, 256 x 256 tiles) instead of transforming on
            // load through registers: 124 -> 83 us for the two 1024-channel blocks, 60 -> ~35 for the 512-channel one (r10d, r10i).
            // The 512-channel block (c_out 256 after it) keeps transform-on-load (measured equal).
            const bool pre = Bk.c_out >= 512;
            if (pre && (rc = bn_relu_rows_launch(X, ld, Bk.tr_s, Bk.tr_t, s.act, Bk.c_out, (int64_t)B * T2, Bk.c_out, st))) return rc;
            MvConv1dDesc d;
            memset(&d, 0, sizeof(d));
            d.x = pre ? s.act : X;
            d.x_dtype = MV_DT_F16;
            d.ldx = ld;  // (the pre-activated copy keeps the block buffer's leading dimension)
            if (pre && (Bk.c_out / 2) % 256 == 0 && (int64_t)B * T2 >= 16384) d.tile = 256;  // 256 x 256 tiles on the ring kernel even where they do not fill the chip (149 tiles for the first transit: r10i)
            d.in_scale = pre ? nullptr : Bk.tr_s;
            d.in_shift = pre ? nullptr : Bk.tr_t;
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
