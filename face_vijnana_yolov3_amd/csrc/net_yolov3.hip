// Secondary path: the full three-scale YOLOv3 inference graph of the reference's
// make_yolov3_model (yolov3_detect.py:217-311), reached only from yolov3_detect.py:_main_ (COCO
// demo), never by FaceDetector.  Base layers 0..73 are the FaceDetector base (same kernels, same
// flat layout); layers 75..105 add the 13/26/52 heads with two UpSampling2D(2)+concatenate routes.
// Inference only (the reference defines no training loss for it).
#include <vector>
#include "conv.h"
#include "elementwise.h"
#include "ops.h"

namespace {

constexpr float BN_EPS = 1e-3f;
constexpr float LEAKY = 0.1f;

enum Src { PREV = 0, BASE, ROUTE79, ROUTE91, CAT61, CAT36 };

struct YLayer {
    fv_layer_desc d;   // role: 0/1/2 as the base; 4 = extra conv+BN+leaky; 5 = detection conv (bias, linear)
    int src;
};

struct YNet {
    std::vector<YLayer> L;
    int64_t nparam = 0, nstate = 0;
    int nbase = 0;
    explicit YNet(int out_ch) {
        const int nb = fv_num_layers() - 1;   // base layers of the FaceDetector table (head excluded)
        for (int i = 0; i < nb; ++i) {
            YLayer y{}; fv_layer(i, &y.d); y.src = PREV;
            L.push_back(y);
            nparam = y.d.beta_off + y.d.cout; nstate = y.d.var_off + y.d.cout;
        }
        nbase = nb;
        auto add = [&](int idx, int k, int cin, int cout, bool bn, int src, int div) {
            YLayer y{};
            y.d.darknet_index = idx; y.d.ksize = k; y.d.stride = 1; y.d.cin = cin; y.d.cout = cout; y.d.has_bn = bn ? 1 : 0;
            y.d.role = bn ? 4 : 5; y.d.in_div = div; y.d.out_div = div;
            y.d.w_off = nparam; nparam += (int64_t)cout * k * k * cin;
            if (bn) {
                y.d.gamma_off = nparam; nparam += cout; y.d.beta_off = nparam; nparam += cout;
                y.d.mean_off = nstate; nstate += cout; y.d.var_off = nstate; nstate += cout;
            } else {
                y.d.gamma_off = -1; y.d.beta_off = nparam; nparam += cout; y.d.mean_off = y.d.var_off = -1;
            }
            y.src = src;
            L.push_back(y);
        };
        // 13x13 branch: five alternating 1x1/3x3, then 3x3 + detection 1x1 (yd.py:269-278)
        int c = 1024;
        const int b13[5][3] = {{75, 1, 512}, {76, 3, 1024}, {77, 1, 512}, {78, 3, 1024}, {79, 1, 512}};
        for (int i = 0; i < 5; ++i) { add(b13[i][0], b13[i][1], c, b13[i][2], true, i == 0 ? BASE : PREV, 32); c = b13[i][2]; }
        add(80, 3, 512, 1024, true, PREV, 32);
        add(81, 1, 1024, out_ch, false, PREV, 32);
        add(84, 1, 512, 256, true, ROUTE79, 32);          // yd.py:281-283 (+ upsample, concat skip_61)
        c = 768;
        const int b26[5][3] = {{87, 1, 256}, {88, 3, 512}, {89, 1, 256}, {90, 3, 512}, {91, 1, 256}};
        for (int i = 0; i < 5; ++i) { add(b26[i][0], b26[i][1], c, b26[i][2], true, i == 0 ? CAT61 : PREV, 16); c = b26[i][2]; }
        add(92, 3, 256, 512, true, PREV, 16);
        add(93, 1, 512, out_ch, false, PREV, 16);
        add(96, 1, 256, 128, true, ROUTE91, 16);          // yd.py:297-299 (+ upsample, concat skip_36)
        c = 384;
        const int b52[6][3] = {{99, 1, 128}, {100, 3, 256}, {101, 1, 128}, {102, 3, 256}, {103, 1, 128}, {104, 3, 256}};
        for (int i = 0; i < 6; ++i) { add(b52[i][0], b52[i][1], c, b52[i][2], true, i == 0 ? CAT36 : PREV, 8); c = b52[i][2]; }
        add(105, 1, 256, out_ch, false, PREV, 8);
    }
};

const YNet& ynet(int out_ch) {
    static YNet n255(255);
    static std::vector<std::pair<int, YNet*>> others;
    if (out_ch == 255) return n255;
    for (auto& o : others) if (o.first == out_ch) return *o.second;
    others.push_back({out_ch, new YNet(out_ch)});
    return *others.back().second;
}

struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    float* take(size_t floats) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += (floats * sizeof(float) + 255) & ~(size_t)255;
        return p;
    }
};

struct YPlan {
    float *scale, *shift, *w0p, *G[3], *s36, *s61, *r79, *r91, *cat, *slab;
    size_t bytes;
};

YPlan yplan(void* base, const YNet& N, int B, int S) {
    YPlan p{};
    Carver c(base);
    p.scale = c.take((size_t)N.nstate / 2); p.shift = c.take((size_t)N.nstate / 2);
    p.w0p = c.take(32 * 32);
    size_t max_act = 0, max_slab = 0;
    for (size_t l = 0; l < N.L.size(); ++l) {
        const auto& d = N.L[l].d;
        size_t rows = (size_t)B * (S / d.out_div) * (S / d.out_div);
        if (rows * d.cout > max_act) max_act = rows * d.cout;
        if (l > 0) {
            int ks = fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32);
            if (ks > 1 && ks * rows * d.cout > max_slab) max_slab = ks * rows * d.cout;
        }
    }
    for (int i = 0; i < 3; ++i) p.G[i] = c.take(max_act);
    p.s36 = c.take((size_t)B * (S / 8) * (S / 8) * 256);
    p.s61 = c.take((size_t)B * (S / 16) * (S / 16) * 512);
    p.r79 = c.take((size_t)B * (S / 32) * (S / 32) * 512);
    p.r91 = c.take((size_t)B * (S / 16) * (S / 16) * 256);
    p.cat = c.take((size_t)B * (S / 8) * (S / 8) * 384 > (size_t)B * (S / 16) * (S / 16) * 768
                       ? (size_t)B * (S / 8) * (S / 8) * 384 : (size_t)B * (S / 16) * (S / 16) * 768);
    p.slab = max_slab ? c.take(max_slab) : nullptr;
    p.bytes = c.off;
    return p;
}

}  // namespace

extern "C" {

int fv_yolov3_num_layers(void) { return (int)ynet(255).L.size(); }
int fv_yolov3_layer(int i, int out_channels, fv_layer_desc* out) {
    const YNet& N = ynet(out_channels);
    if (!out || i < 0 || i >= (int)N.L.size()) return FV_ERR_INVALID;
    *out = N.L[i].d;
    return FV_OK;
}
int64_t fv_yolov3_param_count(int out_channels) { return ynet(out_channels).nparam; }
int64_t fv_yolov3_state_count(int out_channels) { return ynet(out_channels).nstate; }
size_t fv_yolov3_workspace_bytes(int batch, int image_size, int out_channels) {
    if (batch < 1 || image_size < 32 || image_size % 32 || out_channels < 1) return 0;
    return yplan(nullptr, ynet(out_channels), batch, image_size).bytes;
}

int fv_yolov3_forward(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch, int image_size,
                      int out_channels, void* workspace, size_t workspace_bytes, float* y13, float* y26, float* y52) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, params && bn_state && x && workspace && y13 && y26 && y52, "yolov3_forward: NULL buffer");
    FV_REQUIRE(ctx, batch >= 1 && image_size >= 32 && image_size % 32 == 0 && out_channels >= 1, "yolov3_forward: bad shape");
    FV_REQUIRE(ctx, (long long)batch * image_size * image_size * 32 < (1ll << 29), "yolov3_forward: batch too large");
    const YNet& N = ynet(out_channels);
    const int S = image_size;
    YPlan p = yplan(workspace, N, batch, S);
    if (p.bytes > workspace_bytes) return fv_fail(ctx, FV_ERR_WORKSPACE, "yolov3_forward: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    {   // fold every BN layer's moving statistics in one launch
        std::vector<int> chb; std::vector<long long> go, bo, mo, vo;
        for (auto& y : N.L) if (y.d.has_bn) {
            chb.push_back((int)(y.d.mean_off / 2)); go.push_back(y.d.gamma_off); bo.push_back(y.d.beta_off);
            mo.push_back(y.d.mean_off); vo.push_back(y.d.var_off);
        }
        // the fold table holds 64 layers: do it in two halves
        const int nbn = (int)chb.size(), half = nbn / 2;
        if (int rc = fv_ew_bn_fold_all(ctx, params, bn_state, half, chb.data(), go.data(), bo.data(), mo.data(), vo.data(), BN_EPS,
                                       chb[half], p.scale, p.shift)) return rc;
        std::vector<int> chb2(chb.begin() + half, chb.end());
        const int base2 = chb2[0];
        for (auto& v : chb2) v -= base2;
        if (int rc = fv_ew_bn_fold_all(ctx, params, bn_state, nbn - half, chb2.data(), go.data() + half, bo.data() + half,
                                       mo.data() + half, vo.data() + half, BN_EPS, (int)(N.nstate / 2) - base2, p.scale + base2,
                                       p.shift + base2)) return rc;
    }
    if (int rc = fv_ew_pad_rows(ctx, params + N.L[0].d.w_off, p.w0p, 32, 27, 32)) return rc;

    auto conv = [&](const YLayer& y, const float* in, const float* skip, float* out) -> int {
        const auto& d = y.d;
        const int H = S / d.in_div;
        const long long rows = (long long)batch * (H / d.stride) * (H / d.stride);
        const float* w = d.darknet_index == 0 ? p.w0p : params + d.w_off;
        const float* sc = d.has_bn ? p.scale + d.mean_off / 2 : nullptr;
        const float* sh = d.has_bn ? p.shift + d.mean_off / 2 : params + d.beta_off;
        const int ks = d.darknet_index == 0 ? 1 : fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32);
        if (ks > 1) {
            if (int rc = fv_op_conv_forward(ctx, in, w, batch, H, H, d.cin, d.cout, d.ksize, d.stride, 0, nullptr, nullptr, 0.f, nullptr,
                                            p.slab, nullptr, nullptr, ks)) return rc;
            return fv_ew_splitk_finish(ctx, p.slab, ks, rows * d.cout, sc, sh, skip, out, rows * d.cout, d.cout, LEAKY, d.has_bn);
        }
        int epi = FV_EPI_AFFINE | (d.has_bn ? FV_EPI_LEAKY : 0) | (skip ? FV_EPI_ADD : 0);
        return fv_op_conv_forward(ctx, in, w, batch, H, H, d.cin, d.cout, d.ksize, d.stride, epi, sc, sh, LEAKY, skip, out, nullptr, nullptr);
    };

    // ---- base (rotating buffers; the two routed block outputs go to dedicated buffers)
    const float* cur = x;
    int icur = -1, iskip = -1;
    const float* skip = nullptr;
    for (int l = 0; l < N.nbase; ++l) {
        const auto& y = N.L[l];
        if (y.d.role == 1) { skip = cur; iskip = icur; }
        int iout = 0;
        while (iout == icur || (iout == iskip && (y.d.role == 1 || y.d.role == 2))) ++iout;
        float* out = p.G[iout];
        int inew = iout;
        if (y.d.darknet_index == 35) { out = p.s36; inew = -2; }
        if (y.d.darknet_index == 60) { out = p.s61; inew = -3; }
        if (int rc = conv(y, cur, y.d.role == 2 ? skip : nullptr, out)) return rc;
        cur = out; icur = inew;
        if (y.d.role == 2) { skip = nullptr; iskip = -1; }
    }
    const float* base_out = cur;
    const int ibase = icur;
    // ---- heads
    const float* prev = base_out;
    int iprev = ibase;
    for (size_t l = N.nbase; l < N.L.size(); ++l) {
        const auto& y = N.L[l];
        const auto& d = y.d;
        const float* in = prev;
        if (y.src == BASE) in = base_out;
        else if (y.src == ROUTE79) in = p.r79;
        else if (y.src == ROUTE91) in = p.r91;
        else if (y.src == CAT61 || y.src == CAT36) {
            // UpSampling2D(2) of the previous 1x1 output, concatenated in front of the routed skip
            const int Hs = S / (y.src == CAT61 ? 32 : 16), C1 = y.src == CAT61 ? 256 : 128, C2 = y.src == CAT61 ? 512 : 256;
            if (int rc = fv_ew_upsample_concat(ctx, prev, y.src == CAT61 ? p.s61 : p.s36, p.cat, batch, Hs, Hs, C1, C2)) return rc;
            in = p.cat;
        }
        float* out;
        int iout = -1;
        if (d.role == 5) out = d.in_div == 32 ? y13 : (d.in_div == 16 ? y26 : y52);
        else if (d.darknet_index == 79) out = p.r79;
        else if (d.darknet_index == 91) out = p.r91;
        else {
            iout = 0;
            while (iout == iprev || iout == ibase) ++iout;   // keep the base output alive until conv_75 consumed it
            out = p.G[iout];
        }
        if (int rc = conv(y, in, nullptr, out)) return rc;
        if (d.role != 5) { prev = out; iprev = iout; }
        else { prev = nullptr; iprev = -1; }
    }
    return FV_OK;
}

}  // extern "C"
