// Network-level schedule: the FaceDetector model (Darknet-53 base as wired by reference
// face_detection.py:404-593 over yolov3_detect.py:221-267, head face_detection.py:348-352),
// inference forward and the training step (forward, MSE, backward) as a fixed sequence of kernel
// launches on one HIP stream.  The layer table is derived here from the stage structure
// (filters, residual blocks) rather than transcribed.
#include <cstring>
#include <vector>
#include "conv.h"
#include "elementwise.h"
#include "ops.h"

namespace {

constexpr float BN_EPS = 1e-3f;       // yd.py:212
constexpr float BN_MOMENTUM = 0.99f;  // Keras BatchNormalization default
constexpr float LEAKY = 0.1f;         // yd.py:213
constexpr int HEAD_C = 6;             // nn_arch.bb_info_c_size
constexpr int HEAD_PAD = 32;

struct Net {
    std::vector<fv_layer_desc> L;
    int64_t nparam = 0, nstate = 0;
    Net() {
        auto add = [&](int idx, int k, int s, int cin, int cout, int role, int in_div) {
            fv_layer_desc d{};
            d.darknet_index = idx; d.ksize = k; d.stride = s; d.cin = cin; d.cout = cout; d.has_bn = 1; d.role = role;
            d.in_div = in_div; d.out_div = in_div * s;
            d.w_off = nparam; nparam += (int64_t)cout * k * k * cin;
            d.gamma_off = nparam; nparam += cout;
            d.beta_off = nparam; nparam += cout;
            d.mean_off = nstate; nstate += cout;
            d.var_off = nstate; nstate += cout;
            L.push_back(d);
        };
        int idx = 0, div = 1, cin = 32;
        add(idx++, 3, 1, 3, 32, 0, div);
        const int stages[5][2] = {{64, 1}, {128, 2}, {256, 8}, {512, 8}, {1024, 4}};
        for (auto& st : stages) {
            const int cout = st[0];
            add(idx++, 3, 2, cin, cout, 0, div);
            div *= 2;
            for (int b = 0; b < st[1]; ++b) {
                add(idx++, 1, 1, cout, cout / 2, 1, div);
                add(idx++, 3, 1, cout / 2, cout, 2, div);
                ++idx;  // the Darknet shortcut layer owns an index
            }
            cin = cout;
        }
        fv_layer_desc h{};
        h.darknet_index = -1; h.ksize = 3; h.stride = 1; h.cin = 1024; h.cout = HEAD_C; h.has_bn = 0; h.role = 3;
        h.in_div = div; h.out_div = div;
        h.w_off = nparam; nparam += (int64_t)HEAD_C * 9 * 1024;
        h.gamma_off = -1; h.beta_off = nparam; nparam += HEAD_C;
        h.mean_off = h.var_off = -1;
        L.push_back(h);
    }
};
const Net& net() { static Net n; return n; }

struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    float* take(size_t floats) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += (floats * sizeof(float) + 255) & ~(size_t)255;
        return p;
    }
};

struct Plan {
    int B, S, nl;
    std::vector<float*> z, a, mean, invstd, scale, shift, wt;
    std::vector<double*> slots, bslots;   // per layer [nslot][2][cout] fp64 accumulators, forward statistics and
                                          // backward d-beta/d-gamma (one contiguous range over all layers)
    size_t slots_bytes;
    float *w0p, *yhat, *dyp, *G[4], *loss, *slab, *tail, *mse_part, *head_slab;
    int head_ks;
    size_t tail_floats;
    size_t bytes;
};

// Lends the plan's tail-split scratch to the conv launcher for the duration of one network call.
struct TailLend {
    fv_ctx* ctx;
    float* prev; long long prev_floats;
    TailLend(fv_ctx* c, const Plan& p) : ctx(c), prev(c->tail_slab), prev_floats(c->tail_slab_floats) {
        c->tail_slab = nullptr; c->tail_slab_floats = 0;
        if (c->tail_split && p.tail) { c->tail_slab = p.tail; c->tail_slab_floats = (long long)p.tail_floats; }
    }
    ~TailLend() { ctx->tail_slab = prev; ctx->tail_slab_floats = prev_floats; }
};


// Carve the workspace (base == NULL: size query only).
Plan make_plan(void* base, int B, int S, bool training) {
    const Net& N = net();
    Plan p{};
    p.B = B; p.S = S; p.nl = (int)N.L.size();
    Carver c(base);
    const int nb = p.nl - 1;
    p.z.resize(nb); p.a.resize(nb); p.mean.resize(nb); p.invstd.resize(nb); p.scale.resize(nb); p.shift.resize(nb);
    p.wt.resize(p.nl);
    p.slots.resize(nb); p.bslots.resize(nb);
    size_t max_act = 0;
    for (int l = 0; l < nb; ++l) {
        const auto& d = N.L[l];
        size_t hw = (size_t)(S / d.out_div) * (S / d.out_div);
        size_t rows = (size_t)B * hw, elems = rows * d.cout;
        if (elems > max_act) max_act = elems;
    }
    {   // per-channel BN vectors of all layers are contiguous (channel offset = mean_off / 2)
        float* sc_all = c.take((size_t)N.nstate / 2);
        float* sh_all = c.take((size_t)N.nstate / 2);
        for (int l = 0; l < nb; ++l) {
            const auto& d = N.L[l];
            p.scale[l] = sc_all ? sc_all + d.mean_off / 2 : nullptr;
            p.shift[l] = sh_all ? sh_all + d.mean_off / 2 : nullptr;
            if (training) { p.mean[l] = c.take(d.cout); p.invstd[l] = c.take(d.cout); }
        }
    }
    p.w0p = c.take(32 * 32);
    const int G = S / 32;
    p.yhat = c.take((size_t)B * G * G * HEAD_C);
    if (training) {
        for (int l = 0; l < nb; ++l) {
            const auto& d = N.L[l];
            size_t elems = (size_t)B * (S / d.out_div) * (S / d.out_div) * d.cout;
            p.z[l] = c.take(elems); p.a[l] = c.take(elems);
        }
        for (int l = 1; l < p.nl; ++l) {
            const auto& d = N.L[l];
            int cp = d.has_bn ? d.cout : HEAD_PAD;
            p.wt[l] = c.take((size_t)d.cin * d.ksize * d.ksize * cp);
        }
        {   // accumulator slots of all layers in one range (zeroed by one memset per step)
            size_t tot = 0;
            for (int l = 0; l < nb; ++l) tot += (size_t)fv_ew_bn_stat_slots(N.L[l].cout) * 2 * N.L[l].cout;
            double* base_s = (double*)c.take(tot * 4);
            p.slots_bytes = 2 * tot * sizeof(double);
            size_t off = 0;
            for (int l = 0; l < nb; ++l) {
                p.slots[l] = base_s ? base_s + off : nullptr;
                p.bslots[l] = base_s ? base_s + tot + off : nullptr;
                off += (size_t)fv_ew_bn_stat_slots(N.L[l].cout) * 2 * N.L[l].cout;
            }
        }
        p.dyp = c.take((size_t)B * G * G * HEAD_PAD);
        // G[0], G[1]: activation gradients g(l) = dL/d a(l) (the one being consumed / produced and the kept block gradient of a
        // residual pair); G[2], G[3]: dz of the even / odd layers (a weight-gradient on the side stream may still read one)
        for (int i = 0; i < 4; ++i) p.G[i] = c.take(max_act);
        p.loss = c.take(64);
        p.mse_part = c.take(fv_ew_mse_scratch_floats());
        // the head conv has 6 output channels: 53 tiles of 288 K steps -- K-split it like the batch-1 inference path
        p.head_ks = fv_conv_choose_ksplit(B * G * G, HEAD_C, 9 * 1024 / 32);
        p.head_slab = p.head_ks > 1 ? c.take((size_t)p.head_ks * B * G * G * HEAD_C) : nullptr;
    } else {
        for (int i = 0; i < 3; ++i) p.G[i] = c.take(max_act);
        // K-split partial slabs of the small-M layers (batch-1 latency path)
        size_t max_slab = 0;
        for (int l = 1; l < p.nl; ++l) {
            const auto& d = N.L[l];
            size_t rows = (size_t)B * (S / d.out_div) * (S / d.out_div);
            for (int bm64 = 0; bm64 < 2; ++bm64) {       // either setting of option "conv_bm64"
                const int ks = fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32, bm64 != 0);
                if (ks > 1 && ks * rows * d.cout > max_slab) max_slab = ks * rows * d.cout;
            }
        }
        p.slab = max_slab ? c.take(max_slab) : nullptr;
    }
    {   // tail-split scratch: largest need over the forward and (training) stride-1 data-gradient launches
        long long need = 0;
        for (int l = 1; l < p.nl; ++l) {
            const auto& d = N.L[l];
            const int Hi = S / d.in_div, Ho = Hi / d.stride;
            int tf, full; long long n;
            fv_conv_tail_plan(B * Ho * Ho, d.cout, d.ksize * d.ksize * d.cin / 32, &tf, &full, &n);
            if (n > need) need = n;
            if (training && d.stride == 1) {
                const int cp = d.has_bn ? d.cout : HEAD_PAD;
                fv_conv_tail_plan(B * Hi * Hi, d.cin, d.ksize * d.ksize * cp / 32, &tf, &full, &n);
                if (n > need) need = n;
            }
        }
        p.tail_floats = (size_t)need;
        p.tail = need ? c.take((size_t)need) : nullptr;
    }
    p.bytes = c.off;
    return p;
}

int check_shape(fv_ctx* ctx, int batch, int S) {
    FV_REQUIRE(ctx, batch >= 1 && S >= 32 && S % 32 == 0, "image_size must be a positive multiple of 32 (got %d), batch >= 1", S);
    FV_REQUIRE(ctx, (long long)batch * S * S * 32 < (1ll << 31), "batch*S*S*32 exceeds 2^31 elements; reduce the per-GPU batch");
    return FV_OK;
}

}  // namespace

extern "C" {

int fv_num_layers(void) { return (int)net().L.size(); }
int fv_layer(int i, fv_layer_desc* out) {
    if (!out || i < 0 || i >= (int)net().L.size()) return FV_ERR_INVALID;
    *out = net().L[i];
    return FV_OK;
}
int64_t fv_param_count(void) { return net().nparam; }
int64_t fv_state_count(void) { return net().nstate; }

size_t fv_workspace_bytes(int batch, int image_size, int training) {
    if (batch < 1 || image_size < 32 || image_size % 32) return 0;
    return make_plan(nullptr, batch, image_size, training != 0).bytes;
}

int fv_train_workspace_tensor(int batch, int image_size, int layer, int which, size_t* offset_bytes, int64_t* count) {
    if (!offset_bytes || !count || batch < 1 || image_size < 32 || image_size % 32) return FV_ERR_INVALID;
    const Net& N = net();
    const int nb = (int)N.L.size() - 1;
    if (layer < 0 || layer >= nb || which < 0 || which > 5) return FV_ERR_INVALID;
    char* const base = (char*)(uintptr_t)65536;   // any non-null base: only differences are used
    Plan p = make_plan(base, batch, image_size, true);
    const auto& d = N.L[layer];
    const int Ho = image_size / d.out_div;
    const float* t = which == 0 ? p.z[layer] : which == 1 ? p.a[layer] : which == 2 ? p.mean[layer]
                   : which == 3 ? p.invstd[layer] : which == 4 ? p.scale[layer] : p.shift[layer];
    *offset_bytes = (size_t)((const char*)t - base);
    *count = which <= 1 ? (int64_t)batch * Ho * Ho * d.cout : d.cout;
    return FV_OK;
}

// inference forward: the 52 base layers (the last one into `feat` when given), then the head into `y` when given
static int forward_impl(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch, int image_size,
                        void* workspace, size_t workspace_bytes, float* feat, float* y) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, params && bn_state && x && workspace && (y || feat), "forward_infer: NULL buffer");
    if (int rc = check_shape(ctx, batch, image_size)) return rc;
    Plan p = make_plan(workspace, batch, image_size, false);
    if (p.bytes > workspace_bytes) return fv_fail(ctx, FV_ERR_WORKSPACE, "forward_infer: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    TailLend lend(ctx, p);
    const Net& N = net();
    const int nb = p.nl - 1;
    {   // fold the moving statistics of all 52 BN layers into scale/shift with one launch
        int chb[64]; long long go[64], bo[64], mo[64], vo[64];
        for (int l = 0; l < nb; ++l) {
            const auto& d = N.L[l];
            chb[l] = (int)(d.mean_off / 2); go[l] = d.gamma_off; bo[l] = d.beta_off; mo[l] = d.mean_off; vo[l] = d.var_off;
        }
        if (int rc = fv_ew_bn_fold_all(ctx, params, bn_state, nb, chb, go, bo, mo, vo, BN_EPS, (int)(N.nstate / 2), p.scale[0], p.shift[0])) return rc;
    }
    if (int rc = fv_ew_pad_rows(ctx, params + N.L[0].w_off, p.w0p, 32, 27, 32)) return rc;
    // rotating buffers: cur (input), skip (kept while a residual block runs), out
    const float* cur = x;
    int icur = -1, iskip = -1;
    const float* skip = nullptr;
    for (int l = 0; l < nb; ++l) {
        const auto& d = N.L[l];
        const int H = image_size / d.in_div;
        if (d.role == 1) { skip = cur; iskip = icur; }
        int iout = 0;
        while (iout == icur || (iout == iskip && (d.role == 1 || d.role == 2))) ++iout;
        float* out = (feat && l == nb - 1) ? feat : p.G[iout];
        const float* w = l == 0 ? p.w0p : params + d.w_off;
        const long long rows = (long long)batch * (H / d.stride) * (H / d.stride);
        const int ks = l == 0 ? 1 : (ctx->conv_small && d.cin % 32 == 0 && fv_conv_small_plan((int)rows, d.cout, d.cin, d.ksize * d.ksize)) ? 1
                       : fv_conv_choose_ksplit((int)rows, d.cout, d.ksize * d.ksize * d.cin / 32, ctx->conv_bm64);
        if (ks > 1) {
            // small M (batch-1 latency): K-split partial slabs, summed in fixed order by the finish kernel
            if (int rc = fv_op_conv_forward(ctx, cur, w, batch, H, H, d.cin, d.cout, d.ksize, d.stride, 0, nullptr, nullptr, 0.f,
                                            nullptr, p.slab, nullptr, nullptr, ks)) return rc;
            if (int rc = fv_ew_splitk_finish(ctx, p.slab, ks, rows * d.cout, p.scale[l], p.shift[l], d.role == 2 ? skip : nullptr,
                                             out, rows * d.cout, d.cout, LEAKY, 1)) return rc;
        } else {
            int epi = FV_EPI_AFFINE | FV_EPI_LEAKY | (d.role == 2 ? FV_EPI_ADD : 0);
            if (int rc = fv_op_conv_forward(ctx, cur, w, batch, H, H, d.cin, d.cout, d.ksize, d.stride, epi, p.scale[l], p.shift[l],
                                            LEAKY, d.role == 2 ? skip : nullptr, out, nullptr, nullptr)) return rc;
        }
        cur = out; icur = iout;
        if (d.role == 2) { skip = nullptr; iskip = -1; }
    }
    if (!y) return FV_OK;
    const auto& h = N.L[nb];
    const int G = image_size / h.in_div;
    const long long hrows = (long long)batch * G * G;
    const int hks = fv_conv_choose_ksplit((int)hrows, h.cout, 9 * h.cin / 32);
    if (hks > 1) {
        if (int rc = fv_op_conv_forward(ctx, cur, params + h.w_off, batch, G, G, h.cin, h.cout, 3, 1, 0, nullptr, nullptr, 0.f,
                                        nullptr, p.slab, nullptr, nullptr, hks)) return rc;
        return fv_ew_splitk_finish(ctx, p.slab, hks, hrows * h.cout, nullptr, params + h.beta_off, nullptr, y, hrows * h.cout,
                                   h.cout, 0.f, 0);
    }
    return fv_op_conv_forward(ctx, cur, params + h.w_off, batch, G, G, h.cin, h.cout, 3, 1, FV_EPI_AFFINE, nullptr,
                              params + h.beta_off, 0.f, nullptr, y, nullptr, nullptr);
}

int fv_forward_infer(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch, int image_size,
                     void* workspace, size_t workspace_bytes, float* y) {
    if (ctx && !y) return fv_fail(ctx, FV_ERR_INVALID, "forward_infer: NULL buffer");
    return forward_impl(ctx, params, bn_state, x, batch, image_size, workspace, workspace_bytes, nullptr, y);
}

int fv_forward_base(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch, int image_size,
                    void* workspace, size_t workspace_bytes, float* feat, float* y) {
    if (ctx && !feat) return fv_fail(ctx, FV_ERR_INVALID, "forward_base: NULL buffer");
    return forward_impl(ctx, params, bn_state, x, batch, image_size, workspace, workspace_bytes, feat, y);
}

int fv_train_step(fv_ctx* ctx, const float* params, float* bn_state, const float* x, const float* y_true, int batch,
                  int image_size, void* workspace, size_t workspace_bytes, float* grads, float* loss, double loss_weight,
                  fv_bucket_fn on_bucket, void* user) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, params && bn_state && x && y_true && workspace && grads && loss, "train_step: NULL buffer");
    FV_REQUIRE(ctx, loss_weight > 0.0 && loss_weight <= 1.0, "train_step: loss_weight must be in (0, 1] (the slice's share of the merged batch)");
    if (int rc = check_shape(ctx, batch, image_size)) return rc;
    Plan p = make_plan(workspace, batch, image_size, true);
    if (p.bytes > workspace_bytes) return fv_fail(ctx, FV_ERR_WORKSPACE, "train_step: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    TailLend lend(ctx, p);
    // fv_set_bn_zero_debias_step applies to ONE training step: later per-operator BN calls on this context use their own momentum
    struct EmaReset { fv_ctx* c; ~EmaReset() { c->bn_ema_step = 0; } } ema_reset{ctx};
    const Net& N = net();
    const int nb = p.nl - 1;
    const int S = image_size;

    FV_HIP(ctx, hipMemsetAsync(grads, 0, (size_t)N.nparam * sizeof(float), ctx->stream));
    FV_HIP(ctx, hipMemsetAsync(p.slots[0], 0, p.slots_bytes, ctx->stream));   // forward and backward accumulators
    // weight images for this step: packed first layer, transposed kernels for the data-gradients
    if (int rc = fv_ew_pad_rows(ctx, params + N.L[0].w_off, p.w0p, 32, 27, 32)) return rc;
    {
        long long so[64], dof[64]; int tn[64], tt[64], tc[64], tp[64];
        for (int l = 1; l < p.nl; ++l) {
            const auto& d = N.L[l];
            so[l - 1] = d.w_off; dof[l - 1] = p.wt[l] - p.wt[1];
            tn[l - 1] = d.cout; tt[l - 1] = d.ksize * d.ksize; tc[l - 1] = d.cin; tp[l - 1] = d.has_bn ? d.cout : HEAD_PAD;
        }
        if (int rc = fv_ew_transpose_all(ctx, params, p.wt[1], p.nl - 1, so, dof, tn, tt, tc, tp)) return rc;
    }

    // ---------------- forward (training-mode BN)
    const float* cur = x;
    const float* skip = nullptr;
    for (int l = 0; l < nb; ++l) {
        const auto& d = N.L[l];
        const int H = S / d.in_div, Ho = S / d.out_div;
        const long long rows = (long long)batch * Ho * Ho;
        if (d.role == 1) skip = cur;
        const float* w = l == 0 ? p.w0p : params + d.w_off;
        // the column sums go to fp64 accumulator slots and the normalise pass reduces them itself: two
        // launches per layer (the per-operator API keeps the partial-row form + fv_bn_finalize)
        const int ns = fv_ew_bn_stat_slots(d.cout);
        if (int rc = fv_op_conv_forward(ctx, cur, w, batch, H, H, d.cin, d.cout, d.ksize, d.stride, FV_EPI_STATS, nullptr, nullptr,
                                        0.f, nullptr, p.z[l], nullptr, nullptr, 1, p.slots[l], ns)) return rc;
        if (int rc = fv_ew_bn_act_stats(ctx, p.z[l], p.slots[l], ns, (double)rows, params + d.gamma_off, params + d.beta_off,
                                        BN_EPS, BN_MOMENTUM, p.mean[l], p.invstd[l], p.scale[l], p.shift[l],
                                        bn_state + d.mean_off, bn_state + d.var_off, d.role == 2 ? skip : nullptr, p.a[l], rows,
                                        d.cout, LEAKY)) return rc;
        cur = p.a[l];
    }
    const auto& h = N.L[nb];
    const int G = S / h.in_div;
    const int hrows = batch * G * G;
    if (p.head_ks > 1) {
        if (int rc = fv_op_conv_forward(ctx, cur, params + h.w_off, batch, G, G, h.cin, h.cout, 3, 1, 0, nullptr, nullptr, 0.f,
                                        nullptr, p.head_slab, nullptr, nullptr, p.head_ks)) return rc;
        if (int rc = fv_ew_splitk_finish(ctx, p.head_slab, p.head_ks, (long long)hrows * h.cout, nullptr, params + h.beta_off, nullptr,
                                         p.yhat, (long long)hrows * h.cout, h.cout, 0.f, 0)) return rc;
    } else {
        if (int rc = fv_op_conv_forward(ctx, cur, params + h.w_off, batch, G, G, h.cin, h.cout, 3, 1, FV_EPI_AFFINE, nullptr,
                                        params + h.beta_off, 0.f, nullptr, p.yhat, nullptr, nullptr)) return rc;
    }
    // ---------------- loss + its gradient (fd.py:381 'mse')
    if (int rc = fv_ew_mse(ctx, p.yhat, y_true, hrows, HEAD_C, HEAD_PAD, loss, p.dyp, grads + h.beta_off, (double*)p.mse_part,
                          loss_weight)) return rc;

    // ---------------- backward
    // every data-gradient also reduces d-beta / d-gamma of the layer whose output gradient it produces (conv.h
    // FV_EPI_BNRED): the BN-backward of that layer then is the apply pass alone (measured for every layer, also the
    // 32/64-channel ones: fusing all of them 59.4 ms per step, none 61.0).
    auto bnred = [&](int l, FvBnRed& b) -> const FvBnRed* {
        const auto& d = N.L[l];
        b = FvBnRed{p.z[l], p.scale[l], p.shift[l], p.mean[l], p.invstd[l], p.bslots[l], fv_ew_bn_stat_slots(d.cout), LEAKY};
        return &b;
    };
    // A layer's gradient range is handed to the bucket callback once ev_wg[parity] of its weight-gradient (side stream) has been
    // waited for -- or, with fv_set_bucket_on_side, as soon as that weight-gradient is in the side stream's queue (the callback
    // then works on the side stream).  EVERY weight-gradient, the head's included, runs on the side stream: a range is never
    // reported from a stream other than the one its gradient was made on (round 3 ran the head's on the compute stream and
    // reported it at once: with a bucket smaller than the head's 221 KB a side-stream collective could have overtaken it).
    const bool ov = ctx->overlap && ctx->side;
    hipStream_t main_stream = ctx->stream;
    const bool early = ov && ctx->bucket_on_side;      // the callback fires at enqueue time and works on the side stream
    struct Pending { bool on; int64_t off, cnt; } pend[2] = {{false, 0, 0}, {false, 0, 0}};
    auto join = [&](int par) -> int {
        if (!pend[par].on) return FV_OK;
        FV_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_wg[par], 0));
        if (on_bucket && !early) on_bucket(user, pend[par].off, pend[par].cnt);
        pend[par].on = false;
        return FV_OK;
    };
    // weight-gradient of layer l (dy rows of `ndy` floats) on the side stream behind ev_dz[par]; the range [off, off + cnt) is its
    // kernel + (gamma, beta | bias).  ev_wg[par] is recorded BEFORE an early callback: it guards the reuse of the dz buffer, which
    // needs the weight-gradient alone -- recorded after the callback it would make the compute stream wait for the collective the
    // callback enqueued (the host joins the side stream once, before Adam).
    auto wgrad_side = [&](int par, const float* xin, const float* dyv, int H, const fv_layer_desc& d, int ndy, int64_t off, int64_t cnt) -> int {
        if (!ov) {
            if (int rc = fv_op_conv_wgrad(ctx, xin, dyv, batch, H, H, d.cin, d.cout, ndy, d.ksize, d.stride, grads + d.w_off)) return rc;
            if (on_bucket) on_bucket(user, off, cnt);
            return FV_OK;
        }
        FV_HIP(ctx, hipEventRecord(ctx->ev_dz[par], main_stream));
        FV_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_dz[par], 0));
        ctx->stream = ctx->side;
        const int rc = fv_op_conv_wgrad(ctx, xin, dyv, batch, H, H, d.cin, d.cout, ndy, d.ksize, d.stride, grads + d.w_off);
        ctx->stream = main_stream;
        if (rc) return rc;
        FV_HIP(ctx, hipEventRecord(ctx->ev_wg[par], ctx->side));
        if (early && on_bucket) on_bucket(user, off, cnt);      // may enqueue a collective on the side stream
        pend[par] = Pending{true, off, cnt};
        return FV_OK;
    };
    // head: its bias gradient was written by the loss kernel above (compute stream, before ev_dz is recorded)
    if (int rc = wgrad_side(nb & 1, p.a[nb - 1], p.dyp, G, h, HEAD_PAD, h.w_off, (int64_t)HEAD_C * 9 * 1024 + HEAD_C)) return rc;
    FvBnRed bnr;
    if (int rc = fv_op_conv_dgrad(ctx, p.dyp, p.wt[nb], batch, G, G, h.cin, HEAD_PAD, 3, 1, nullptr, p.G[0], bnred(nb - 1, bnr))) return rc;
    {
    // G[ig]: gradient w.r.t. the current layer's (post-add) output; G[ires]: kept block gradient.
    // dz(l) -> D[l&1].  The weight-gradient of layer l only needs dz(l) and the saved forward
    // activation, so it runs on the side stream while this stream continues with dgrad(l) and
    // bn_bwd(l-1); D[l&1] is reused by layer l-2 only after wgrad(l) has signalled ev_wg[l&1].
    float* const D[2] = {p.G[2], p.G[3]};
    int ig = 0, ires = -1;
    for (int l = nb - 1; l >= 0; --l) {
        const auto& d = N.L[l];
        const int H = S / d.in_div, Ho = S / d.out_div;
        const long long rows = (long long)batch * Ho * Ho;
        const int par = l & 1;
        if (int rc = join(par)) return rc;
        float* dz = D[par];
        if (d.role == 2) ires = ig;  // add(skip, x): the same gradient also reaches the skip input
        if (int rc = fv_ew_bn_bwd(ctx, p.G[ig], p.z[l], p.scale[l], p.shift[l], p.mean[l], p.invstd[l], rows, d.cout, LEAKY,
                                  nullptr, nullptr, grads + d.beta_off, grads + d.gamma_off, dz, p.bslots[l], fv_ew_bn_stat_slots(d.cout),
                                  true)) return rc;
        const float* xin = l == 0 ? x : p.a[l - 1];
        const int64_t cnt = (int64_t)d.cout * d.ksize * d.ksize * d.cin + 2 * d.cout;
        if (int rc = wgrad_side(par, xin, dz, H, d, d.cout, d.w_off, cnt)) return rc;
        if (l == 0) break;
        // dgrad overwrites the consumed gradient buffer G[ig] unless that is the kept block gradient
        const int iout = (ig == ires) ? 1 - ig : ig;
        const float* addend = d.role == 1 ? p.G[ires] : nullptr;
        if (int rc = fv_op_conv_dgrad(ctx, dz, p.wt[l], batch, H, H, d.cin, d.cout, d.ksize, d.stride, addend, p.G[iout],
                                      bnred(l - 1, bnr))) return rc;
        ig = iout;
        if (d.role == 1) ires = -1;
    }
    }
    if (int rc = join(1)) return rc;   // layer 1, then layer 0: ranges stay in descending order
    if (int rc = join(0)) return rc;
    return FV_OK;
}

}  // extern "C"
