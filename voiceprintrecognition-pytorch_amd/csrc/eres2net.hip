// ERes2Net / ERes2NetV2 forward orchestrated natively (mvector/models/eres2net.py:173-287, 383-456).
//
// create(): reads the reference-layout fp32 state_dict, folds every eval-mode BatchNorm into the conv in front of it
// (BN follows each conv directly: eres2net.py:86-87, 96, 101-102; AFF eres2net.py:39-45), pads the channel groups to
// multiples of 16 and packs the weights for conv2ds_kernel: split fp16 pairs with a per-layer power-of-two scale, maps in the S16 form (conv2ds.hip;
// the fp32 kernels of conv2d.hip remain the CAM++ fp32 head's).  The two width-sized channel groups that
// torch.split / torch.cat move around (eres2net.py:89, 99) are fixed slices of two buffers instead:
//   A  = conv1 output   [.., scale * wpad]   group i = spx[i]
//   Bc = "cat" buffer   [.., scale * wpad]   group i = relu(bn_i(conv_i(.)))
// so a 3x3 conv writes slice i of Bc -- and, in the plain blocks, "its output + slice i+1 of A" (eres2net.py:92) as a second output into
// a width-sized temporary that the next 3x3 conv reads (its loader is a plain copy); the AFF blocks fuse (slice i-1 of Bc, slice i of A) first;
// conv3 reads Bc whole.  Padded channels carry exact zeros through every layer.
// forward(): a fixed sequence of launches on the caller's stream over the caller's workspace.
#include <memory>
#include <vector>

#include "kernels.h"
#include "model.h"

namespace mv {

namespace {

struct C2Layer {
    half_t* w = nullptr;   // split-packed (conv2ds_pack_host)
    float* bias = nullptr;
    float oscale = 0.0f;
    int cin16 = 0, cout16 = 0, ks = 1, stride = 1;
    int cin = 0, cout = 0;  // the layer's own channel counts (algorithmic FLOP accounting)
};

struct AffLayer {  // AFF (eres2net.py:32-52) on `ch` channels (padded chp per operand)
    C2Layer att1, att2;
    int ch = 0, chp = 0, inter16 = 0;
};

struct Block {
    C2Layer conv1, conv3, shortcut;
    std::vector<C2Layer> convs;
    std::vector<AffLayer> fuse;  // empty for the plain blocks
    bool has_shortcut = false;
    int in_c16 = 0, out_c16 = 0, width = 0, wpad = 0, stride = 1;
};

std::vector<int> dense_map(int c) {
    std::vector<int> m(c);
    for (int i = 0; i < c; ++i) m[i] = i;
    return m;
}

// channel c of a [scale * width] tensor lives at (c / width) * wpad + c % width
std::vector<int> grouped_map(int width, int scale, int wpad) {
    std::vector<int> m((size_t)width * scale);
    for (int c = 0; c < width * scale; ++c) m[c] = (c / width) * wpad + c % width;
    return m;
}

}  // namespace

struct Eres2Model : MvModelBase {
    MvEres2Cfg cfg;
    int m = 0, scale = 2;
    float* stem_w = nullptr;  // [m][9] BN folded
    float* stem_b = nullptr;
    std::vector<Block> layers[4];
    // ERes2Net: layer{1,2,3}_downsample + fuse_mode{12,123,1234}; ERes2NetV2: layer3_ds + fuse34 (slot 2)
    C2Layer ds[3];
    AffLayer top_fuse[3];
    float* seg1_w = nullptr;
    float* seg1_b = nullptr;
    float* seg2_w = nullptr;
    float* seg2_b = nullptr;
    int final_c = 0, final_h = 0;

    // ---- packing -----------------------------------------------------------------------------------------------
    // w: [cout][cin][ks][ks]; out channel c -> row out_pos[c] scaled by oscale[c]; in channel c -> column in_pos[c]
    int pack(const std::vector<float>& w, int cout, int cin, int ks, const std::vector<float>& oscale, const std::vector<float>& obias,
             const std::vector<int>& out_pos, int cout16, const std::vector<int>& in_pos, int cin16, int stride, C2Layer* L) {
        const int taps = ks * ks;
        std::vector<float> packed((size_t)cout16 * taps * cin16, 0.0f);
        std::vector<float> bias((size_t)cout16, 0.0f);
        for (int co = 0; co < cout; ++co) {
            const float s = oscale.empty() ? 1.0f : oscale[co];
            bias[out_pos[co]] = obias.empty() ? 0.0f : obias[co];
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t) {
                    packed[((size_t)out_pos[co] * taps + t) * cin16 + in_pos[ci]] = w[((size_t)co * cin + ci) * taps + t] * s;
                }
        }
        std::vector<half_t> split((size_t)conv2ds_packed_floats(cout16, cin16, ks) * 2);
        L->oscale = conv2ds_pack_host(packed.data(), cout16, cin16, ks, split.data());
        L->w = static_cast<half_t*>(dev_alloc(split.size() * sizeof(half_t)));
        L->bias = upload(bias);
        if (L->w == nullptr || L->bias == nullptr) return fail(MV_ERR_HIP, "eres2net create: out of device memory");
        MV_HIP_OK(hipMemcpy(L->w, split.data(), split.size() * sizeof(half_t), hipMemcpyHostToDevice));
        L->cin16 = cin16;
        L->cout16 = cout16;
        L->cin = cin;
        L->cout = cout;
        L->ks = ks;
        L->stride = stride;
        return MV_OK;
    }

    // conv (no bias) followed by BatchNorm `bn` (empty: none)
    int make_conv_bn(const Weights& w, const std::string& conv, const std::string& bn, int cout, int cin, int ks, int stride,
                     const std::vector<int>& out_pos, int cout16, const std::vector<int>& in_pos, int cin16, C2Layer* L) {
        std::vector<float> W, s, t, cb;
        int rc;
        if ((rc = w.host(conv + ".weight", (int64_t)cout * cin * ks * ks, W))) return rc;
        if (w.has(conv + ".bias") && (rc = w.host(conv + ".bias", cout, cb))) return rc;
        if (!bn.empty()) {
            if ((rc = fold_bn(w, bn, cout, s, t, 1e-5f))) return rc;
            if (!cb.empty())
                for (int c = 0; c < cout; ++c) t[c] += cb[c] * s[c];
        } else if (!cb.empty()) {
            t = cb;
        }
        return pack(W, cout, cin, ks, s, t, out_pos, cout16, in_pos, cin16, stride, L);
    }

    int make_aff(const Weights& w, const std::string& prefix, int ch, AffLayer* A) {
        const int r = 4, inter = ch / r;
        MV_REQUIRE(inter > 0, "eres2net: AFF needs at least 4 channels");
        A->ch = ch;
        A->chp = (int)round_up(ch, 16);
        A->inter16 = (int)round_up(inter, 16);
        std::vector<int> in_pos((size_t)2 * ch);
        for (int c = 0; c < 2 * ch; ++c) in_pos[c] = c < ch ? c : A->chp + (c - ch);
        int rc;
        if ((rc = make_conv_bn(w, prefix + ".local_att.0", prefix + ".local_att.1", inter, 2 * ch, 1, 1, dense_map(inter), A->inter16,
                               in_pos, 2 * A->chp, &A->att1)))
            return rc;
        return make_conv_bn(w, prefix + ".local_att.3", prefix + ".local_att.4", ch, inter, 1, 1, dense_map(ch), A->chp, dense_map(inter),
                            A->inter16, &A->att2);
    }

    int make_block(const Weights& w, const std::string& p, int in_planes, int planes, int stride, bool aff, Block* b) {
        const int width = (int)((double)planes * ((double)cfg.base_width / 64.0));  // floor (eres2net.py:59)
        MV_REQUIRE(width >= 4, "eres2net: block width too small");
        b->width = width;
        b->wpad = (int)round_up(width, 16);
        b->stride = stride;
        b->in_c16 = (int)round_up(in_planes, 16);
        const int out_planes = planes * cfg.expansion;
        b->out_c16 = (int)round_up(out_planes, 16);
        const std::vector<int> gmap = grouped_map(width, scale, b->wpad);
        int rc;
        if ((rc = make_conv_bn(w, p + ".conv1", p + ".bn1", width * scale, in_planes, 1, stride, gmap, scale * b->wpad, dense_map(in_planes),
                               b->in_c16, &b->conv1)))
            return rc;
        b->convs.resize(scale);
        for (int i = 0; i < scale; ++i)
            if ((rc = make_conv_bn(w, p + ".convs." + std::to_string(i), p + ".bns." + std::to_string(i), width, width, 3, 1,
                                   dense_map(width), b->wpad, dense_map(width), b->wpad, &b->convs[i])))
                return rc;
        if (aff) {
            b->fuse.resize(scale - 1);
            for (int j = 0; j < scale - 1; ++j)
                if ((rc = make_aff(w, p + ".fuse_models." + std::to_string(j), width, &b->fuse[j]))) return rc;
        }
        if ((rc = make_conv_bn(w, p + ".conv3", p + ".bn3", out_planes, width * scale, 1, 1, dense_map(out_planes), b->out_c16, gmap,
                               scale * b->wpad, &b->conv3)))
            return rc;
        b->has_shortcut = stride != 1 || in_planes != out_planes;
        if (b->has_shortcut)
            if ((rc = make_conv_bn(w, p + ".shortcut.0", p + ".shortcut.1", out_planes, in_planes, 1, stride, dense_map(out_planes),
                                   b->out_c16, dense_map(in_planes), b->in_c16, &b->shortcut)))
                return rc;
        return MV_OK;
    }

    int create(const MvEres2Cfg& c, const Weights& w) {
        cfg = c;
        MV_REQUIRE(c.version == 1 || c.version == 2, "eres2net: version must be 1 (ERes2Net) or 2 (ERes2NetV2)");
        MV_REQUIRE(c.input_size >= 8 && c.input_size % 8 == 0, "eres2net: input_size must be a multiple of 8");
        MV_REQUIRE(c.m_channels >= 16 && c.m_channels % 16 == 0, "eres2net: m_channels must be a multiple of 16");
        MV_REQUIRE(c.scale >= 2 && c.scale <= 8 && c.expansion == 2, "eres2net: scale in 2..8, expansion 2 (eres2net.py:414 fixes m_channels*8)");
        if (c.version == 1) MV_REQUIRE(c.mul_channel == 1, "eres2net: mul_channel must be 1 (the fusion shapes only match then)");
        for (int i = 0; i < 4; ++i) MV_REQUIRE(c.num_blocks[i] >= 1, "eres2net: every stage needs a block");
        m = c.m_channels;
        scale = c.scale;
        embd_dim = c.embd_dim;
        input_size = c.input_size;
        int rc;
        {   // conv1 + bn1 + relu (eres2net.py:196-201, 270): fp32 weights for the VALU stem kernel
            std::vector<float> W, s, t;
            if ((rc = w.host("conv1.weight", (int64_t)m * 9, W)) || (rc = fold_bn(w, "bn1", m, s, t, 1e-5f))) return rc;
            for (int co = 0; co < m; ++co)
                for (int j = 0; j < 9; ++j) W[(size_t)co * 9 + j] *= s[co];
            stem_w = upload(W);
            stem_b = upload(t);
            if (stem_w == nullptr || stem_b == nullptr) return fail(MV_ERR_HIP, "eres2net create: upload failed");
        }
        int in_planes = m;
        for (int l = 0; l < 4; ++l) {
            const int planes = m << l;
            const bool aff = l >= 2;  // layer3 / layer4 use the *_AFF blocks
            layers[l].resize(c.num_blocks[l]);
            for (int j = 0; j < c.num_blocks[l]; ++j) {
                const int stride = (j == 0 && l > 0) ? 2 : 1;
                if ((rc = make_block(w, "layer" + std::to_string(l + 1) + "." + std::to_string(j), in_planes, planes, stride, aff,
                                     &layers[l][j])))
                    return rc;
                in_planes = planes * c.expansion;
            }
        }
        const int e = c.expansion;
        if (c.version == 1) {
            const char* dsn[3] = {"layer1_downsample", "layer2_downsample", "layer3_downsample"};
            const char* fn[3] = {"fuse_mode12", "fuse_mode123", "fuse_mode1234"};
            for (int i = 0; i < 3; ++i) {
                const int cin = (m << i) * e, cout = (m << (i + 1)) * e;
                if ((rc = make_conv_bn(w, dsn[i], "", cout, cin, 3, 2, dense_map(cout), (int)round_up(cout, 16), dense_map(cin),
                                       (int)round_up(cin, 16), &ds[i])) ||
                    (rc = make_aff(w, fn[i], cout, &top_fuse[i])))
                    return rc;
            }
        } else {
            const int cin = (m << 2) * e, cout = (m << 3) * e;
            if ((rc = make_conv_bn(w, "layer3_ds", "", cout, cin, 3, 2, dense_map(cout), (int)round_up(cout, 16), dense_map(cin),
                                   (int)round_up(cin, 16), &ds[2])) ||
                (rc = make_aff(w, "fuse34", cout, &top_fuse[2])))
                return rc;
        }
        final_c = (m << 3) * e;
        final_h = c.input_size / 8;
        const int K = 2 * final_c * final_h;
        if ((rc = fold_final_linear(this, w, "seg_1.weight", "seg_1.bias", "", "", c.embd_dim, K, &seg1_w, &seg1_b))) return rc;
        if (c.two_emb_layer)
            if ((rc = fold_final_linear(this, w, "seg_2.weight", "seg_2.bias", "seg_bn_1", "", c.embd_dim, c.embd_dim, &seg2_w, &seg2_b)))
                return rc;
        return MV_OK;
    }

    // ---- workspace ---------------------------------------------------------------------------------------------
    struct Ws {
        float *ping[2], *a, *bc, *r, *t, *t2, *hh, *out[4], *dsb, *fuse[2];   // maps: S16 form, 4 bytes per channel
        float *stats, *emb_a, *lin_ws;
        size_t bytes, lin_ws_floats;
    };
    static int down(int n) { return (n - 1) / 2 + 1; }

    Ws carve(void* base, int B, int T) const {
        Carver c(base);
        Ws s;
        int H[4], W[4];
        H[0] = cfg.input_size;
        W[0] = T;
        for (int l = 1; l < 4; ++l) {
            H[l] = down(H[l - 1]);
            W[l] = down(W[l - 1]);
        }
        size_t max_io = 0, max_a = 0, max_r = 0, max_t = 0, max_h = 0;
        for (int l = 0; l < 4; ++l) {
            const size_t px = (size_t)B * H[l] * W[l];
            const size_t px_in = l == 0 ? px : (size_t)B * H[l - 1] * W[l - 1];
            max_io = std::max(max_io, px_in * (size_t)(l == 0 ? round_up(m, 16) : layers[l][0].in_c16));
            for (const Block& b : layers[l]) {
                max_io = std::max(max_io, px * (size_t)b.out_c16);
                max_a = std::max(max_a, px * (size_t)scale * b.wpad);
                max_r = std::max(max_r, px * (size_t)b.out_c16);
                max_t = std::max(max_t, px * (size_t)b.wpad);
                for (const AffLayer& f : b.fuse) max_h = std::max(max_h, px * (size_t)f.inter16);
            }
        }
        for (int i = 0; i < 3; ++i)
            if (top_fuse[i].ch) max_h = std::max(max_h, (size_t)B * H[i + 1] * W[i + 1] * top_fuse[i].inter16);
        const size_t slack = 64;  // the loader reads whole 16-byte chunks
        s.ping[0] = c.take<float>(max_io + slack);
        s.ping[1] = c.take<float>(max_io + slack);
        s.a = c.take<float>(max_a + slack);
        s.bc = c.take<float>(max_a + slack);
        s.r = c.take<float>(max_r + slack);
        s.t = c.take<float>(max_t + slack);
        s.t2 = c.take<float>(max_t + slack);
        s.hh = c.take<float>(max_h + slack);
        for (int l = 0; l < 4; ++l) s.out[l] = c.take<float>((size_t)B * H[l] * W[l] * layers[l].back().out_c16 + slack);
        size_t max_ds = 0, max_f = 0;
        for (int i = 0; i < 3; ++i)
            if (top_fuse[i].ch) {
                max_ds = std::max(max_ds, (size_t)B * H[i + 1] * W[i + 1] * ds[i].cout16);
                max_f = std::max(max_f, (size_t)B * H[i + 1] * W[i + 1] * top_fuse[i].chp);
            }
        s.dsb = c.take<float>(max_ds + slack);
        s.fuse[0] = c.take<float>(max_f + slack);
        s.fuse[1] = c.take<float>(max_f + slack);
        s.stats = c.take<float>((size_t)B * 2 * final_c * final_h);
        s.emb_a = c.take<float>((size_t)B * cfg.embd_dim);
        s.lin_ws_floats = linear_f32_splitk_floats(B, 2 * final_c * final_h, cfg.embd_dim);   // K slices of seg_1 (linear.hip)
        s.lin_ws = c.take<float>(s.lin_ws_floats);
        s.bytes = c.total();
        return s;
    }

    int workspace_bytes(int B, int T, size_t* bytes) const override {
        MV_REQUIRE(B > 0 && T >= 9 && bytes != nullptr, "eres2net workspace: needs B > 0 and at least 9 frames");
        *bytes = carve(nullptr, B, T).bytes;
        return MV_OK;
    }

    // ---- launches ----------------------------------------------------------------------------------------------
    // x2 != nullptr: the channels behind the first cin1 come from x2 (AFF concatenation); y2 != nullptr: second output y + add
    static int conv(const C2Layer& L, const float* x, int64_t ldx, const float* x2, int64_t ldx2, int cin1, float* y, int64_t ldy, int B, int H,
                    int W, int epi, float lo, float hi, const float* res, int64_t ldres, const float* res2, int64_t ldres2, hipStream_t st,
                    const float* add = nullptr, int64_t ldadd = 0, float* y2 = nullptr, int64_t ldy2 = 0) {
        MvConv2dsDesc d{};
        d.x = x; d.x2 = x2; d.cin1 = cin1; d.ldx = ldx; d.ldx2 = ldx2;
        d.w = L.w; d.bias = L.bias; d.oscale = L.oscale; d.res = res; d.res2 = res2; d.ldres = ldres; d.ldres2 = ldres2;
        d.add = add; d.ldadd = ldadd; d.y2 = y2; d.ldy2 = ldy2;
        d.y = y; d.ldy = ldy; d.B = B; d.H = H; d.W = W; d.cin16 = L.cin16; d.cout16 = L.cout16; d.ks = L.ks; d.stride = L.stride;
        d.epi = epi; d.lo = lo; d.hi = hi;
        d.cin_alg = L.cin; d.cout_alg = L.cout;
        return conv2ds_launch(d, st);
    }

    // out = xa * (1 + tanh(att)) + ya * (1 - tanh(att)),  att = BN(conv(SiLU(BN(conv(cat(xa, ya))))))   (eres2net.py:47-52)
    static int aff(const AffLayer& A, const float* xa, int64_t ldxa, const float* ya, int64_t ldya, float* hidden, float* out,
                   int64_t ldo, int B, int H, int W, hipStream_t st) {
        int rc = conv(A.att1, xa, ldxa, ya, ldya, A.chp, hidden, A.inter16, B, H, W, MV_EPI_SILU, 0.0f, 0.0f, nullptr, 0, nullptr, 0, st);
        if (rc != MV_OK) return rc;
        return conv(A.att2, hidden, A.inter16, nullptr, 0, 0, out, ldo, B, H, W, MV_EPI_AFF, 0.0f, 0.0f, xa, ldxa, ya, ldya, st);
    }

    int run_block(const Block& b, const float* x, float* y, const Ws& s, int B, int Hin, int Win, hipStream_t st) const {
        const float NEG = -3.0e38f, POS = 3.0e38f;
        const int Ho = b.stride == 2 ? down(Hin) : Hin, Wo = b.stride == 2 ? down(Win) : Win;
        const int64_t lda = (int64_t)scale * b.wpad;
        int rc;
        // out = relu(bn1(conv1(x)))   (eres2net.py:86-88)
        if ((rc = conv(b.conv1, x, b.in_c16, nullptr, 0, 0, s.a, lda, B, Hin, Win, MV_EPI_CLAMP, 0.0f, 20.0f, nullptr, 0, nullptr, 0, st)))
            return rc;
        float* const sums[2] = {s.t, s.t2};
        for (int i = 0; i < scale; ++i) {
            const float* spx = s.a + (int64_t)i * b.wpad;
            float* dst = s.bc + (int64_t)i * b.wpad;
            if (b.fuse.empty()) {
                // sp = sp + spx[i] (eres2net.py:92) was formed by the previous conv's second output; this one forms the next
                const bool more = i + 1 < scale;
                rc = conv(b.convs[i], i == 0 ? spx : sums[(i - 1) & 1], i == 0 ? lda : b.wpad, nullptr, 0, 0, dst, lda, B, Ho, Wo, MV_EPI_CLAMP, 0.0f, 20.0f,
                          nullptr, 0, nullptr, 0, st, more ? spx + b.wpad : nullptr, lda, more ? sums[i & 1] : nullptr, b.wpad);
            } else if (i == 0) {
                rc = conv(b.convs[0], spx, lda, nullptr, 0, 0, dst, lda, B, Ho, Wo, MV_EPI_CLAMP, 0.0f, 20.0f, nullptr, 0, nullptr, 0, st);
            } else {                      // sp = fuse_models[i-1](sp, spx[i])   (eres2net.py:152)
                if ((rc = aff(b.fuse[i - 1], dst - b.wpad, lda, spx, lda, s.hh, s.t, b.wpad, B, Ho, Wo, st))) return rc;
                rc = conv(b.convs[i], s.t, b.wpad, nullptr, 0, 0, dst, lda, B, Ho, Wo, MV_EPI_CLAMP, 0.0f, 20.0f, nullptr, 0, nullptr, 0, st);
            }
            if (rc != MV_OK) return rc;
        }
        const float* resid = x;
        int64_t ldr = b.in_c16;
        if (b.has_shortcut) {
            if ((rc = conv(b.shortcut, x, b.in_c16, nullptr, 0, 0, s.r, b.out_c16, B, Hin, Win, MV_EPI_CLAMP, NEG, POS, nullptr, 0, nullptr, 0, st)))
                return rc;
            resid = s.r;
            ldr = b.out_c16;
        }
        // relu(bn3(conv3(cat)) + residual)   (eres2net.py:101-106)
        return conv(b.conv3, s.bc, lda, nullptr, 0, 0, y, b.out_c16, B, Ho, Wo, MV_EPI_CLAMP, 0.0f, 20.0f, resid, ldr, nullptr, 0, st);
    }

    int forward(const float* feats, int B, int T, float* emb, void* ws, size_t ws_bytes, hipStream_t st) const override {
        MV_REQUIRE(feats != nullptr && emb != nullptr && ws != nullptr, "eres2net forward: null buffer");
        MV_REQUIRE(B > 0 && T >= 9, "eres2net forward: needs at least 9 frames (two time steps after three stride-2 stages)");
        const Ws s = carve(ws, B, T);
        if (s.bytes > ws_bytes) return fail(MV_ERR_WORKSPACE, "eres2net forward: workspace too small");
        const float NEG = -3.0e38f, POS = 3.0e38f;
        int rc;
        int H = cfg.input_size, W = T;
        int Hs[4], Wsz[4];
        // x.permute(0, 2, 1).unsqueeze(1) -> relu(bn1(conv1(x)))   (eres2net.py:267-270)
        if ((rc = conv2d_first_s16_launch(feats, reinterpret_cast<half_t*>(s.ping[0]), stem_w, stem_b, B, T, cfg.input_size, m, st))) return rc;
        const float* cur = s.ping[0];
        int pp = 1;
        for (int l = 0; l < 4; ++l) {
            const int nb = (int)layers[l].size();
            for (int j = 0; j < nb; ++j) {
                const Block& b = layers[l][j];
                float* dst = j == nb - 1 ? s.out[l] : s.ping[pp];
                if ((rc = run_block(b, cur, dst, s, B, H, W, st))) return rc;
                if (b.stride == 2) {
                    H = down(H);
                    W = down(W);
                }
                cur = dst;
                if (j != nb - 1) pp ^= 1;
            }
            Hs[l] = H;
            Wsz[l] = W;
        }
        const float* pooled = nullptr;
        int64_t pooled_ld = 0;
        if (cfg.version == 1) {
            // bottom-up fusion (eres2net.py:273-281): fuse(out_{k+1}, downsample(previous fused / out1))
            const float* prev = s.out[0];
            int64_t prev_ld = layers[0].back().out_c16;
            for (int i = 0; i < 3; ++i) {
                if ((rc = conv(ds[i], prev, prev_ld, nullptr, 0, 0, s.dsb, ds[i].cout16, B, Hs[i], Wsz[i], MV_EPI_CLAMP, NEG, POS, nullptr, 0,
                               nullptr, 0, st)))
                    return rc;
                float* fo = s.fuse[i & 1];
                if ((rc = aff(top_fuse[i], s.out[i + 1], layers[i + 1].back().out_c16, s.dsb, ds[i].cout16, s.hh, fo, top_fuse[i].chp, B,
                              Hs[i + 1], Wsz[i + 1], st)))
                    return rc;
                prev = fo;
                prev_ld = top_fuse[i].chp;
            }
            pooled = prev;
            pooled_ld = prev_ld;
        } else {
            // fuse34(out4, layer3_ds(out3))   (eres2net.py:446-447)
            if ((rc = conv(ds[2], s.out[2], layers[2].back().out_c16, nullptr, 0, 0, s.dsb, ds[2].cout16, B, Hs[2], Wsz[2], MV_EPI_CLAMP, NEG,
                           POS, nullptr, 0, nullptr, 0, st)) ||
                (rc = aff(top_fuse[2], s.out[3], layers[3].back().out_c16, s.dsb, ds[2].cout16, s.hh, s.fuse[0], top_fuse[2].chp, B, Hs[3],
                          Wsz[3], st)))
                return rc;
            pooled = s.fuse[0];
            pooled_ld = top_fuse[2].chp;
        }
        MV_REQUIRE(Hs[3] == final_h, "eres2net forward: unexpected frequency size after the four stages");
        if ((rc = tstp_s16_launch(reinterpret_cast<const half_t*>(pooled), pooled_ld, B, Hs[3], Wsz[3], final_c, s.stats, st))) return rc;
        const int K = 2 * final_c * final_h;
        if (!cfg.two_emb_layer)
            return linear_f32_launch(s.stats, K, seg1_w, K, seg1_b, MV_ACT_NONE, emb, cfg.embd_dim, B, K, cfg.embd_dim, 0, st, s.lin_ws, s.lin_ws_floats);
        // embed_a -> relu -> seg_bn_1 (folded into seg_2) -> seg_2   (eres2net.py:285-288)
        if ((rc = linear_f32_launch(s.stats, K, seg1_w, K, seg1_b, MV_ACT_RELU, s.emb_a, cfg.embd_dim, B, K, cfg.embd_dim, 0, st, s.lin_ws, s.lin_ws_floats))) return rc;
        return linear_f32_launch(s.emb_a, cfg.embd_dim, seg2_w, cfg.embd_dim, seg2_b, MV_ACT_NONE, emb, cfg.embd_dim, B, cfg.embd_dim,
                                 cfg.embd_dim, 0, st);
    }
};

}  // namespace mv

extern "C" {

int mv_eres2net_create(const MvEres2Cfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out) {
    MV_REQUIRE(cfg != nullptr && out != nullptr, "mv_eres2net_create: null argument");
    mv::Weights w;
    int rc = w.init(tensors, num_tensors);
    if (rc != MV_OK) return rc;
    auto m = std::make_unique<mv::Eres2Model>();
    rc = m->create(*cfg, w);
    if (rc != MV_OK) return rc;
    *out = reinterpret_cast<MvModel*>(static_cast<mv::MvModelBase*>(m.release()));
    return MV_OK;
}

}  // extern "C"
