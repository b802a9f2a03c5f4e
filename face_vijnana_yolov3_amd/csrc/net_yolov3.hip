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
constexpr float BN_MOMENTUM = 0.99f;
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
            for (int bm64 = 0; bm64 < 2; ++bm64) {       // either setting of option "conv_bm64"
                const int ks = fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32, bm64 != 0);
                if (ks > 1 && ks * rows * d.cout > max_slab) max_slab = ks * rows * d.cout;
            }
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

// ---------------------------------------------------------------------------------------------- training plan
// Everything the backward pass needs is kept: per BN layer z (pre-BN) and a (activated), the two concatenated
// tensors, the three raw head outputs and their padded gradients.
struct YTrain {
    std::vector<float*> z, a, mean, invstd, scale, shift, wt;
    std::vector<double*> slots, bslots;
    size_t slots_bytes = 0;
    float *w0p, *cat61, *cat36, *y[3], *dy[3], *GA, *GB, *DZ, *DZ2, *gs61, *gs36, *r79, *r91, *loss_part, *colsum_part, *tail;
    size_t tail_floats = 0, bytes = 0;
    int cpad = 0;
};

YTrain ytrain_plan(void* base, const YNet& N, int B, int S, int out_ch) {
    YTrain p{};
    Carver c(base);
    const int nl = (int)N.L.size();
    p.z.resize(nl); p.a.resize(nl); p.mean.resize(nl); p.invstd.resize(nl); p.scale.resize(nl); p.shift.resize(nl); p.wt.resize(nl);
    p.slots.resize(nl); p.bslots.resize(nl);
    p.cpad = (out_ch + 31) / 32 * 32;
    size_t max_act = 0, slot_tot = 0;
    for (int l = 0; l < nl; ++l) {
        const auto& d = N.L[l].d;
        const size_t elems = (size_t)B * (S / d.out_div) * (S / d.out_div) * d.cout;
        if (d.has_bn) {
            p.z[l] = c.take(elems); p.a[l] = c.take(elems);
            p.mean[l] = c.take(d.cout); p.invstd[l] = c.take(d.cout); p.scale[l] = c.take(d.cout); p.shift[l] = c.take(d.cout);
            slot_tot += (size_t)fv_ew_bn_stat_slots(d.cout) * 2 * d.cout;
            if (elems > max_act) max_act = elems;
        }
        if (l > 0) p.wt[l] = c.take((size_t)d.cin * d.ksize * d.ksize * (d.has_bn ? d.cout : p.cpad));
    }
    {
        double* bs = (double*)c.take(slot_tot * 4);
        p.slots_bytes = 2 * slot_tot * sizeof(double);
        size_t off = 0;
        for (int l = 0; l < nl; ++l) {
            const auto& d = N.L[l].d;
            if (!d.has_bn) continue;
            p.slots[l] = bs ? bs + off : nullptr; p.bslots[l] = bs ? bs + slot_tot + off : nullptr;
            off += (size_t)fv_ew_bn_stat_slots(d.cout) * 2 * d.cout;
        }
    }
    p.w0p = c.take(32 * 32);
    const size_t cat61 = (size_t)B * (S / 16) * (S / 16) * 768, cat36 = (size_t)B * (S / 8) * (S / 8) * 384;
    p.cat61 = c.take(cat61); p.cat36 = c.take(cat36);
    if (cat61 > max_act) max_act = cat61;
    if (cat36 > max_act) max_act = cat36;
    for (int s = 0; s < 3; ++s) {
        const size_t rows = (size_t)B * (S / (32 >> s)) * (S / (32 >> s));
        p.y[s] = c.take(rows * out_ch); p.dy[s] = c.take(rows * p.cpad);
    }
    p.GA = c.take(max_act); p.GB = c.take(max_act); p.DZ = c.take(max_act); p.DZ2 = c.take(max_act);
    p.gs61 = c.take((size_t)B * (S / 16) * (S / 16) * 512); p.gs36 = c.take((size_t)B * (S / 8) * (S / 8) * 256);
    p.r79 = c.take((size_t)B * (S / 32) * (S / 32) * 512); p.r91 = c.take((size_t)B * (S / 16) * (S / 16) * 256);
    p.loss_part = c.take(2 * 3 * 1024 + 64);
    p.colsum_part = c.take(2 * 64 * (size_t)p.cpad);
    {
        long long need = 0;
        for (int l = 1; l < nl; ++l) {
            const auto& d = N.L[l].d;
            const int Hi = S / d.in_div, Ho = Hi / d.stride;
            int tf, full; long long n;
            fv_conv_tail_plan(B * Ho * Ho, d.cout, d.ksize * d.ksize * d.cin / 32, &tf, &full, &n);
            if (n > need) need = n;
            if (d.stride == 1) {
                fv_conv_tail_plan(B * Hi * Hi, d.cin, d.ksize * d.ksize * (d.has_bn ? d.cout : p.cpad) / 32, &tf, &full, &n);
                if (n > need) need = n;
            }
        }
        p.tail_floats = (size_t)need;
        p.tail = need ? c.take((size_t)need) : nullptr;
    }
    p.bytes = c.off;
    return p;
}

int find_base(const YNet& N, int darknet_idx) {
    for (int l = 0; l < N.nbase; ++l) if (N.L[l].d.darknet_index == darknet_idx) return l;
    return -1;
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
        const int ks = d.darknet_index == 0 ? 1 : (ctx->conv_small && d.cin % 32 == 0 && fv_conv_small_plan((int)rows, d.cout, d.cin, d.ksize * d.ksize)) ? 1
                       : fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32, ctx->conv_bm64);
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

size_t fv_yolov3_train_workspace_bytes(int batch, int image_size, int out_channels) {
    if (batch < 1 || image_size < 32 || image_size % 32 || out_channels < 1 || out_channels % 3) return 0;
    return ytrain_plan(nullptr, ynet(out_channels), batch, image_size, out_channels).bytes;
}

int fv_yolov3_train_workspace_tensor(int batch, int image_size, int out_channels, int layer, int which, size_t* offset_bytes,
                                     int64_t* count) {
    if (!offset_bytes || !count || batch < 1 || image_size < 32 || image_size % 32 || out_channels < 18 || out_channels % 3) return FV_ERR_INVALID;
    const YNet& N = ynet(out_channels);
    if (layer < 0 || layer >= (int)N.L.size() || which < 0 || which > 5 || !N.L[layer].d.has_bn) return FV_ERR_INVALID;
    char* const base = (char*)(uintptr_t)65536;
    YTrain p = ytrain_plan(base, N, batch, image_size, out_channels);
    const auto& d = N.L[layer].d;
    const int Ho = image_size / d.out_div;
    const float* t = which == 0 ? p.z[layer] : which == 1 ? p.a[layer] : which == 2 ? p.mean[layer]
                   : which == 3 ? p.invstd[layer] : which == 4 ? p.scale[layer] : p.shift[layer];
    *offset_bytes = (size_t)((const char*)t - base);
    *count = which <= 1 ? (int64_t)batch * Ho * Ho * d.cout : d.cout;
    return FV_OK;
}

int fv_yolov3_train_step(fv_ctx* ctx, const float* params, float* bn_state, const float* x, const float* yt13, const float* yt26,
                         const float* yt52, int batch, int image_size, int out_channels, void* workspace, size_t workspace_bytes,
                         float* grads, float* loss, double loss_weight, fv_bucket_fn on_bucket, void* user) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, loss_weight > 0.0 && loss_weight <= 1.0, "yolov3_train_step: loss_weight must be in (0, 1]");
    FV_REQUIRE(ctx, params && bn_state && x && yt13 && yt26 && yt52 && workspace && grads && loss, "yolov3_train_step: NULL buffer");
    FV_REQUIRE(ctx, batch >= 1 && image_size >= 32 && image_size % 32 == 0 && out_channels >= 18 && out_channels % 3 == 0,
               "yolov3_train_step: bad shape (out_channels = 3*(5+classes))");
    FV_REQUIRE(ctx, (long long)batch * image_size * image_size * 32 < (1ll << 29), "yolov3_train_step: batch too large");
    const YNet& N = ynet(out_channels);
    const int S = image_size, B = batch, nl = (int)N.L.size(), nb = N.nbase;
    const int ncls = out_channels / 3 - 5;
    YTrain p = ytrain_plan(workspace, N, B, S, out_channels);
    if (p.bytes > workspace_bytes) return fv_fail(ctx, FV_ERR_WORKSPACE, "yolov3_train_step: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    float* const prev_tail = ctx->tail_slab; const long long prev_tail_floats = ctx->tail_slab_floats;
    struct Restore { fv_ctx* c; float* s; long long n; ~Restore() { c->tail_slab = s; c->tail_slab_floats = n; } } restore{ctx, prev_tail, prev_tail_floats};
    ctx->tail_slab = ctx->tail_split ? p.tail : nullptr; ctx->tail_slab_floats = ctx->tail_split ? (long long)p.tail_floats : 0;
    struct EmaReset { fv_ctx* c; ~EmaReset() { c->bn_ema_step = 0; } } ema_reset{ctx};   // the zero-debias step applies to this call only

    FV_HIP(ctx, hipMemsetAsync(grads, 0, (size_t)N.nparam * sizeof(float), ctx->stream));
    FV_HIP(ctx, hipMemsetAsync(p.slots[0], 0, p.slots_bytes, ctx->stream));
    if (int rc = fv_ew_pad_rows(ctx, params + N.L[0].d.w_off, p.w0p, 32, 27, 32)) return rc;
    for (int l0 = 1; l0 < nl; l0 += 60) {   // transposed kernels for the data-gradients (table of 64 entries per launch)
        long long so[64], dof[64]; int tn[64], tt[64], tc[64], tp[64];
        const int cnt = nl - l0 < 60 ? nl - l0 : 60;
        for (int i = 0; i < cnt; ++i) {
            const auto& d = N.L[l0 + i].d;
            so[i] = d.w_off; dof[i] = p.wt[l0 + i] - p.wt[l0];
            tn[i] = d.cout; tt[i] = d.ksize * d.ksize; tc[i] = d.cin; tp[i] = d.has_bn ? d.cout : p.cpad;
        }
        if (int rc = fv_ew_transpose_all(ctx, params, p.wt[l0], cnt, so, dof, tn, tt, tc, tp)) return rc;
    }
    const int l35 = find_base(N, 35), l60 = find_base(N, 60);
    const int P79 = nb + 4, P80 = nb + 5, D13 = nb + 6, P84 = nb + 7, P87 = nb + 8, P91 = nb + 12, P92 = nb + 13, D26 = nb + 14, P96 = nb + 15,
              P99 = nb + 16, P104 = nb + 21, D52 = nb + 22;
    FV_REQUIRE(ctx, l35 >= 0 && l60 >= 0 && D52 == nl - 1 && N.L[P79].d.darknet_index == 79 && N.L[P91].d.darknet_index == 91 &&
                        N.L[P84].d.darknet_index == 84 && N.L[P96].d.darknet_index == 96 && N.L[P104].d.darknet_index == 104,
               "yolov3_train_step: unexpected layer table");

    // ------------------------------------------------------------------ forward (training-mode BN everywhere)
    auto input_of = [&](int l) -> const float* {
        if (l == 0) return x;
        switch (N.L[l].src) {
            case BASE: return p.a[nb - 1];
            case ROUTE79: return p.a[P79];
            case ROUTE91: return p.a[P91];
            case CAT61: return p.cat61;
            case CAT36: return p.cat36;
            default: return p.a[l - 1];
        }
    };
    const float* skip = nullptr;
    for (int l = 0; l < nl; ++l) {
        const auto& d = N.L[l].d;
        const int H = S / d.in_div, Ho = S / d.out_div;
        const long long rows = (long long)B * Ho * Ho;
        if (N.L[l].src == CAT61) { if (int rc = fv_ew_upsample_concat(ctx, p.a[P84], p.a[l60], p.cat61, B, S / 32, S / 32, 256, 512)) return rc; }
        if (N.L[l].src == CAT36) { if (int rc = fv_ew_upsample_concat(ctx, p.a[P96], p.a[l35], p.cat36, B, S / 16, S / 16, 128, 256)) return rc; }
        const float* in = input_of(l);
        if (d.role == 1) skip = in;
        const float* w = l == 0 ? p.w0p : params + d.w_off;
        if (d.has_bn) {
            const int ns = fv_ew_bn_stat_slots(d.cout);
            if (int rc = fv_op_conv_forward(ctx, in, w, B, H, H, d.cin, d.cout, d.ksize, d.stride, FV_EPI_STATS, nullptr, nullptr, 0.f, nullptr,
                                            p.z[l], nullptr, nullptr, 1, p.slots[l], ns)) return rc;
            if (int rc = fv_ew_bn_act_stats(ctx, p.z[l], p.slots[l], ns, (double)rows, params + d.gamma_off, params + d.beta_off, BN_EPS,
                                            BN_MOMENTUM, p.mean[l], p.invstd[l], p.scale[l], p.shift[l], bn_state + d.mean_off,
                                            bn_state + d.var_off, d.role == 2 ? skip : nullptr, p.a[l], rows, d.cout, LEAKY)) return rc;
        } else {
            const int sidx = d.in_div == 32 ? 0 : (d.in_div == 16 ? 1 : 2);
            if (int rc = fv_op_conv_forward(ctx, in, w, B, H, H, d.cin, d.cout, d.ksize, d.stride, FV_EPI_AFFINE, nullptr, params + d.beta_off,
                                            0.f, nullptr, p.y[sidx], nullptr, nullptr)) return rc;
        }
    }
    // ------------------------------------------------------------------ loss of the three scales, its gradient, bias gradients
    const float* yt[3] = {yt13, yt26, yt52};
    const int det[3] = {D13, D26, D52};
    long long cells[3];
    double* lpart = (double*)p.loss_part;
    {
        int off = 0;
        for (int s = 0; s < 3; ++s) {
            cells[s] = (long long)B * (S / (32 >> s)) * (S / (32 >> s));
            if (int rc = fv_ew_yolo_loss_part(ctx, p.y[s], yt[s], cells[s], ncls, 3, p.cpad, p.dy[s], lpart + off, loss_weight)) return rc;
            off += fv_ew_yolo_loss_blocks(cells[s] * 3);
            if (int rc = fv_ew_colsum(ctx, p.dy[s], cells[s], out_channels, p.cpad, (double*)p.colsum_part, grads + N.L[det[s]].d.beta_off)) return rc;
        }
        if (int rc = fv_ew_yolo_loss_finish(ctx, lpart, cells, 3, loss)) return rc;
    }
    // ------------------------------------------------------------------ backward
    FvBnRed bnr;
    auto bnred = [&](int l) -> const FvBnRed* {
        if (l < 0) return nullptr;
        bnr = FvBnRed{p.z[l], p.scale[l], p.shift[l], p.mean[l], p.invstd[l], p.bslots[l], fv_ew_bn_stat_slots(N.L[l].d.cout), LEAKY};
        return &bnr;
    };
    // Weight-gradients run on the low-priority side stream, as in fv_train_step (net.hip): wgrad(l) needs only dz(l) and a saved
    // forward activation, so the main stream goes on with dgrad(l) and the next layer's BN-backward.  dz alternates between two
    // buffers; a buffer is rewritten only after the weight-gradient that read it has signalled ev_wg[slot].
    const bool ov = ctx->overlap && ctx->side;
    hipStream_t main_stream = ctx->stream;
    float* const DZs[2] = {p.DZ, p.DZ2};
    // A layer's gradient range [w_off, + kernel + (gamma, beta | bias)) is complete once its weight-gradient has finished: it is
    // handed to the bucket callback when ev_wg[slot] has been waited for (the slots alternate strictly, so ranges are reported
    // in issue order = reverse execution order = descending offsets, the protocol of fv_train_step).
    // fv_set_bucket_on_side: the callback fires when the weight-gradient is in the side stream's queue and works on that stream.
    const bool early = ov && ctx->bucket_on_side;
    struct Pending { bool on; int64_t off, cnt; } pend[2] = {{false, 0, 0}, {false, 0, 0}};
    int slot = 0;
    auto join = [&](int s) -> int {
        if (!pend[s].on) return FV_OK;
        FV_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_wg[s], 0));
        if (on_bucket && !early) on_bucket(user, pend[s].off, pend[s].cnt);
        pend[s].on = false;
        return FV_OK;
    };
    auto wgrad = [&](int s, int l, const float* xin, const float* dyv, int H, int ndy, float* dw) -> int {
        const auto& d = N.L[l].d;
        const int64_t cnt = (int64_t)d.cout * d.ksize * d.ksize * d.cin + (d.has_bn ? 2 : 1) * d.cout;
        if (!ov) {
            if (int rc = fv_op_conv_wgrad(ctx, xin, dyv, B, H, H, d.cin, d.cout, ndy, d.ksize, d.stride, dw)) return rc;
            if (on_bucket) on_bucket(user, d.w_off, cnt);
            return FV_OK;
        }
        FV_HIP(ctx, hipEventRecord(ctx->ev_dz[s], main_stream));
        FV_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_dz[s], 0));
        ctx->stream = ctx->side;
        const int rc = fv_op_conv_wgrad(ctx, xin, dyv, B, H, H, d.cin, d.cout, ndy, d.ksize, d.stride, dw);
        ctx->stream = main_stream;
        if (rc) return rc;
        // recorded BEFORE an early callback: the event guards the reuse of the dz buffer, which needs the weight-gradient alone
        // (after the callback it would make the compute stream wait for the collective the callback enqueued; net.hip)
        FV_HIP(ctx, hipEventRecord(ctx->ev_wg[s], ctx->side));
        if (early && on_bucket) on_bucket(user, d.w_off, cnt);      // may enqueue a collective on the side stream
        pend[s] = Pending{true, d.w_off, cnt};
        return FV_OK;
    };
    // detection conv: dy (padded) -> dW, and g of its input layer `lin` (with that layer's d-beta/d-gamma reduction)
    auto det_bwd = [&](int l, int sidx, int lin, float* g_out) -> int {
        const auto& d = N.L[l].d;
        const int H = S / d.in_div;
        if (int rc = join(slot)) return rc;
        if (int rc = wgrad(slot, l, p.a[lin], p.dy[sidx], H, p.cpad, grads + d.w_off)) return rc;
        slot ^= 1;
        return fv_op_conv_dgrad(ctx, p.dy[sidx], p.wt[l], B, H, H, d.cin, p.cpad, d.ksize, 1, nullptr, g_out, bnred(lin));
    };
    // BN layer l: g (its d-beta/d-gamma already in the slots unless !reduced) -> dz -> dW; data-gradient into g_out (+ addend),
    // reducing for layer `lred` (-1: none: the gradient of that tensor is not complete yet, or it is a concatenation)
    auto bn_bwd = [&](int l, const float* g, bool reduced, const float* xin, float* g_out, const float* addend, int lred) -> int {
        const auto& d = N.L[l].d;
        const int H = S / d.in_div, Ho = S / d.out_div;
        const long long rows = (long long)B * Ho * Ho;
        if (int rc = join(slot)) return rc;
        float* dz = DZs[slot];
        if (int rc = fv_ew_bn_bwd(ctx, g, p.z[l], p.scale[l], p.shift[l], p.mean[l], p.invstd[l], rows, d.cout, LEAKY, nullptr, nullptr,
                                  grads + d.beta_off, grads + d.gamma_off, dz, p.bslots[l], fv_ew_bn_stat_slots(d.cout), reduced)) return rc;
        if (int rc = wgrad(slot, l, xin, dz, H, d.cout, grads + d.w_off)) return rc;
        slot ^= 1;
        if (!g_out) return FV_OK;
        return fv_op_conv_dgrad(ctx, dz, p.wt[l], B, H, H, d.cin, d.cout, d.ksize, d.stride, addend, g_out, bnred(lred));
    };
    float *ga = p.GA, *gb = p.GB;
    auto swap = [&]() { float* t = ga; ga = gb; gb = t; };
    // 52x52 head
    if (int rc = det_bwd(D52, 2, P104, ga)) return rc;
    for (int l = P104; l > P99; --l) { if (int rc = bn_bwd(l, ga, true, p.a[l - 1], gb, nullptr, l - 1)) return rc; swap(); }
    if (int rc = bn_bwd(P99, ga, true, p.cat36, gb, nullptr, -1)) return rc;
    if (int rc = fv_ew_upsample_concat_bwd(ctx, gb, ga, p.gs36, B, S / 16, S / 16, 128, 256)) return rc;       // ga = g(a96), gs36 = g(skip_36) part
    if (int rc = bn_bwd(P96, ga, false, p.a[P91], p.r91, nullptr, -1)) return rc;
    // 26x26 head
    if (int rc = det_bwd(D26, 1, P92, ga)) return rc;
    if (int rc = bn_bwd(P92, ga, true, p.a[P91], gb, p.r91, P91)) return rc;
    swap();
    for (int l = P91; l > P87; --l) { if (int rc = bn_bwd(l, ga, true, p.a[l - 1], gb, nullptr, l - 1)) return rc; swap(); }
    if (int rc = bn_bwd(P87, ga, true, p.cat61, gb, nullptr, -1)) return rc;
    if (int rc = fv_ew_upsample_concat_bwd(ctx, gb, ga, p.gs61, B, S / 32, S / 32, 256, 512)) return rc;
    if (int rc = bn_bwd(P84, ga, false, p.a[P79], p.r79, nullptr, -1)) return rc;
    // 13x13 head
    if (int rc = det_bwd(D13, 0, P80, ga)) return rc;
    if (int rc = bn_bwd(P80, ga, true, p.a[P79], gb, p.r79, P79)) return rc;
    swap();
    for (int l = P79; l > nb; --l) { if (int rc = bn_bwd(l, ga, true, p.a[l - 1], gb, nullptr, l - 1)) return rc; swap(); }
    if (int rc = bn_bwd(nb, ga, true, p.a[nb - 1], gb, nullptr, nb - 1)) return rc;     // conv_75 reads the base output
    swap();
    // base: the chain of fv_train_step, with the two routed gradients joining where the forward pass branched off
    {
        float* G2[2] = {ga, gb};
        int ig = 0, ires = -1;
        for (int l = nb - 1; l >= 0; --l) {
            const auto& d = N.L[l].d;
            if (d.role == 2) ires = ig;
            const float* xin = l == 0 ? x : p.a[l - 1];
            if (l == 0) { if (int rc = bn_bwd(0, G2[ig], true, xin, nullptr, nullptr, -1)) return rc; break; }
            const int iout = (ig == ires) ? 1 - ig : ig;
            const float* addend = d.role == 1 ? G2[ires] : nullptr;
            if (l - 1 == l60) addend = p.gs61;       // conv_62 reads the tensor that was also routed to the 26x26 head
            if (l - 1 == l35) addend = p.gs36;       // conv_37 reads the tensor that was also routed to the 52x52 head
            if (int rc = bn_bwd(l, G2[ig], true, xin, G2[iout], addend, l - 1)) return rc;
            ig = iout;
            if (d.role == 1) ires = -1;
        }
    }
    if (int rc = join(slot)) return rc;          // `slot` now names the older of the two outstanding weight-gradients
    return join(slot ^ 1);
}

}  // extern "C"
