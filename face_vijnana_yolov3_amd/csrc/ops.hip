// Operator-level entry points of the C ABI: translate Keras-style conv descriptions into the
// gather-convolution tap lists of conv.h.
#include "conv.h"
#include "elementwise.h"
#include "ops.h"

static void fwd_taps(int ksize, FvTaps& t) {
    if (ksize == 1) { t.n = 1; t.dh[0] = t.dw[0] = 0; t.wslot[0] = 0; return; }
    t.n = 9;
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) { int i = r * 3 + q; t.dh[i] = r - 1; t.dw[i] = q - 1; t.wslot[i] = i; }
}

int fv_op_conv_forward(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout, int ksize,
                       int stride, int epi, const float* scale, const float* shift, float leaky, const float* addend,
                       float* out, float* psum, float* psq, int ksplit, double* stat_slots, int stat_nslot) {
    FV_REQUIRE(ctx, (ksize == 1 && stride == 1) || (ksize == 3 && (stride == 1 || stride == 2)), "conv: unsupported k=%d s=%d", ksize, stride);
    FV_REQUIRE(ctx, H % stride == 0 && W % stride == 0, "conv: H,W must be divisible by the stride");
    FvConvArgs a{};
    a.x = x; a.w = w; a.out = out; a.addend = addend; a.scale = scale; a.shift = shift; a.psum = psum; a.psq = psq;
    a.B = B; a.Hin = H; a.Win = W; a.Cin = cin;
    a.Hl = H / stride; a.Wl = W / stride; a.Hout = a.Hl; a.Wout = a.Wl; a.Nout = cout;
    a.is = stride; a.os = 1; a.Tw = ksize * ksize; a.M = B * a.Hl * a.Wl;
    a.epi = epi; a.leaky = leaky; a.nclass = 1;
    fwd_taps(ksize, a.taps[0]);
    if (cin % 32 != 0) a.Tw = 1;  // packed [cout][32] first-layer weights
    a.alg_flops = 2.0 * a.M * cout * (double)(ksize * ksize * cin);
    a.ksplit = ksplit; a.split_stride = (long long)a.M * cout;
    a.stat_slots = stat_slots; a.stat_nslot = stat_nslot;
    // inference epilogues only: the training forward keeps one tiling whatever the batch (its statistics are per-tile sums)
    a.small = (ctx->conv_small && ksplit <= 1 && !(epi & FV_EPI_STATS) && cin % 32 == 0) ? fv_conv_small_plan(a.M, cout, cin, ksize * ksize) : 0;
    a.narrow = (!a.small && ksize == 1 && ksplit <= 1 && !(epi & FV_EPI_STATS) && cin % 32 == 0 && fv_conv_narrow(a.M, cout, cin / 32)) ? 1 : 0;
    a.bm64 = (ctx->conv_bm64 && !a.small && !a.narrow && !(epi & FV_EPI_STATS) && cin % 32 == 0 && fv_conv_bm64(a.M, cout, ksize * ksize * cin / 32)) ? 1 : 0;
    return fv_conv_launch(ctx, a);
}

int fv_op_conv_dgrad(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin, int cout_pad, int ksize,
                     int stride, const float* addend, float* dx, const FvBnRed* bn) {
    FV_REQUIRE(ctx, (ksize == 1 && stride == 1) || (ksize == 3 && (stride == 1 || stride == 2)), "dgrad: unsupported k=%d s=%d", ksize, stride);
    FV_REQUIRE(ctx, cout_pad % 32 == 0, "dgrad: cout_pad must be a multiple of 32");
    FV_REQUIRE(ctx, H % stride == 0 && W % stride == 0, "dgrad: H,W must be divisible by the stride");
    FvConvArgs a{};
    a.x = dy; a.w = w_t; a.out = dx; a.addend = addend;
    a.B = B; a.Hin = H / stride; a.Win = W / stride; a.Cin = cout_pad;
    a.Hout = H; a.Wout = W; a.Nout = cin;
    a.is = 1; a.Tw = ksize * ksize;
    a.epi = addend ? FV_EPI_ADD : 0; a.leaky = 0.f;
    if (bn) {
        a.epi |= FV_EPI_BNRED;
        a.bn_z = bn->z; a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_mean = bn->mean; a.bn_invstd = bn->invstd;
        a.bn_slots = bn->slots; a.bn_nslot = bn->nslot; a.bn_leaky = bn->leaky;
    }
    if (stride == 1) {
        a.Hl = H; a.Wl = W; a.os = 1; a.nclass = 1;
        FvTaps& t = a.taps[0];
        if (ksize == 1) { t.n = 1; t.dh[0] = t.dw[0] = 0; t.wslot[0] = 0; }
        else {
            t.n = 9;   // dx[h] = sum_r dz[h + 1 - r] w[r]
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) { int i = r * 3 + q; t.dh[i] = 1 - r; t.dw[i] = 1 - q; t.wslot[i] = i; }
        }
    } else {
        // forward: z[oh] reads x[2 oh - 1 + r].  Output pixel h = 2a+ph receives from r with
        // (h+1-r) even: ph=0 -> r=1 (oh=a); ph=1 -> r=0 (oh=a+1), r=2 (oh=a).  One class per (ph,pw).
        FV_REQUIRE(ctx, H % 2 == 0 && W % 2 == 0, "dgrad: stride 2 needs even H, W");
        a.Hl = H / 2; a.Wl = W / 2; a.os = 2; a.nclass = 4;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                // class = blockIdx.z, dispatched in ascending order: the 4-tap class (ph = pw = 1) goes
                // first and the 1-tap class last, so the longest tiles are not left for the tail
                int c = 3 - (ph * 2 + pw);
                a.oph[c] = ph; a.opw[c] = pw;
                FvTaps& t = a.taps[c];
                t.n = 0;
                for (int r = 0; r < 3; ++r) {
                    if ((ph + 1 - r) % 2 != 0) continue;
                    for (int q = 0; q < 3; ++q) {
                        if ((pw + 1 - q) % 2 != 0) continue;
                        t.dh[t.n] = (ph + 1 - r) / 2; t.dw[t.n] = (pw + 1 - q) / 2; t.wslot[t.n] = r * 3 + q;
                        ++t.n;
                    }
                }
            }
    }
    a.M = B * a.Hl * a.Wl;
    // same MACs as the forward conv it differentiates (cout_real unknown here: padded channels are zeros)
    a.alg_flops = 2.0 * (double)B * (H / stride) * (W / stride) * cin * (double)(ksize * ksize * cout_pad);
    return fv_conv_launch(ctx, a);
}

int fv_op_conv_wgrad(fv_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int cin, int cout, int dy_stride,
                     int ksize, int stride, float* dw) {
    FV_REQUIRE(ctx, (ksize == 1 && stride == 1) || (ksize == 3 && (stride == 1 || stride == 2)), "wgrad: unsupported k=%d s=%d", ksize, stride);
    FvWgradArgs a{};
    a.x = x; a.dy = dy; a.dw = dw;
    a.B = B; a.Hin = H; a.Win = W; a.Cin = cin;
    a.Hl = H / stride; a.Wl = W / stride; a.N = cout; a.Ndy = dy_stride;
    a.is = stride; a.Tw = ksize * ksize; a.M = B * a.Hl * a.Wl;
    fwd_taps(ksize, a.taps);
    a.alg_flops = 2.0 * a.M * cout * (double)(ksize * ksize * cin);
    return fv_wgrad_launch(ctx, a);
}

extern "C" {

int fv_conv2d_stat_rows(int64_t out_pixels) { return fv_conv_mtiles((int)out_pixels, 0); }

int fv_conv2d_forward(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout, int ksize,
                      int stride, const float* scale, const float* shift, float leaky, const float* addend, float* out,
                      float* psum, float* psq) {
    if (!ctx) return FV_ERR_INVALID;
    int epi = 0;
    if (psum) {
        FV_REQUIRE(ctx, !scale && !shift && !addend && leaky < 0.f, "conv2d_forward: statistics mode stores the raw result");
        epi = FV_EPI_STATS;
    } else {
        if (scale || shift) epi |= FV_EPI_AFFINE;
        if (leaky >= 0.f) epi |= FV_EPI_LEAKY;
        if (addend) epi |= FV_EPI_ADD;
    }
    return fv_op_conv_forward(ctx, x, w, B, H, W, cin, cout, ksize, stride, epi, scale, shift, leaky, addend, out, psum, psq, 1);
}

int fv_conv2d_dgrad(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin, int cout_pad, int ksize,
                    int stride, const float* addend, float* dx) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_op_conv_dgrad(ctx, dy, w_t, B, H, W, cin, cout_pad, ksize, stride, addend, dx, nullptr);
}

int fv_conv2d_wgrad(fv_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int cin, int cout, int dy_stride,
                    int ksize, int stride, float* dw) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_op_conv_wgrad(ctx, x, dy, B, H, W, cin, cout, dy_stride, ksize, stride, dw);
}

int fv_transpose_weights(fv_ctx* ctx, const float* w, int cout, int taps, int cin, int cout_pad, float* w_t) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, cout_pad >= cout, "transpose_weights: cout_pad < cout");
    return fv_ew_transpose_ntc(ctx, w, w_t, cout, taps, cin, cout_pad);
}

int fv_pack_first_layer(fv_ctx* ctx, const float* w, int cout, int k_elems, float* w_packed) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, k_elems <= 32, "pack_first_layer: k_elems > 32");
    return fv_ew_pad_rows(ctx, w, w_packed, cout, k_elems, 32);
}

int fv_bn_finalize(fv_ctx* ctx, const float* psum, const float* psq, int stat_rows, int C, int64_t count, const float* gamma,
                   const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                   float* moving_mean, float* moving_var) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_ew_bn_finalize(ctx, psum, psq, stat_rows, C, (double)count, gamma, beta, eps, momentum, mean, invstd, scale, shift,
                             moving_mean, moving_var);
}

int fv_bn_act(fv_ctx* ctx, const float* z, const float* scale, const float* shift, const float* skip, float* out, int64_t rows,
              int C, float leaky) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_ew_bn_act(ctx, z, scale, shift, skip, out, rows, C, leaky);
}

int64_t fv_bn_bwd_scratch_floats(int64_t rows, int C) { return (int64_t)fv_ew_bn_bwd_chunks(rows, C) * C; }

int fv_bn_bwd(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift, const float* mean,
              const float* invstd, int64_t rows, int C, float leaky, float* scratch, float* dbeta, float* dgamma, float* dz) {
    if (!ctx) return FV_ERR_INVALID;
    int64_t half = fv_bn_bwd_scratch_floats(rows, C);
    return fv_ew_bn_bwd(ctx, g, z, scale, shift, mean, invstd, rows, C, leaky, scratch, scratch + half, dbeta, dgamma, dz);
}

int fv_bn_stat_slots(int C) { return fv_ew_bn_stat_slots(C); }

static int slots_ok(fv_ctx* ctx, const double* slots, int nslot, int C, const char* who) {
    FV_REQUIRE(ctx, slots && nslot >= 1 && C % 4 == 0 && C <= 1024 && (C >= 256 ? true : 256 % C == 0),
               "%s: needs accumulator slots, C %% 4 == 0, C <= 1024 and C dividing or divided by 256 (C=%d)", who, C);
    return FV_OK;
}

int fv_conv2d_forward_slots(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout, int ksize,
                            int stride, float* z, double* slots, int nslot) {
    if (!ctx) return FV_ERR_INVALID;
    if (int rc = slots_ok(ctx, slots, nslot, cout, "conv2d_forward_slots")) return rc;
    return fv_op_conv_forward(ctx, x, w, B, H, W, cin, cout, ksize, stride, FV_EPI_STATS, nullptr, nullptr, 0.f, nullptr, z,
                              nullptr, nullptr, 1, slots, nslot);
}

int fv_bn_act_slots(fv_ctx* ctx, const float* z, const double* slots, int nslot, int64_t rows, int C, const float* gamma,
                    const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                    float* moving_mean, float* moving_var, const float* skip, float* out, float leaky) {
    if (!ctx) return FV_ERR_INVALID;
    if (int rc = slots_ok(ctx, slots, nslot, C, "bn_act_slots")) return rc;
    FV_REQUIRE(ctx, z && gamma && beta && mean && invstd && scale && shift && out && rows > 0, "bn_act_slots: NULL buffer");
    return fv_ew_bn_act_stats(ctx, z, slots, nslot, (double)rows, gamma, beta, eps, momentum, mean, invstd, scale, shift,
                              moving_mean, moving_var, skip, out, rows, C, leaky);
}

int fv_conv2d_dgrad_bnred(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin, int cout_pad, int ksize,
                          int stride, const float* addend, float* dx, const float* bn_z, const float* scale, const float* shift,
                          const float* mean, const float* invstd, float leaky, double* slots, int nslot) {
    if (!ctx) return FV_ERR_INVALID;
    if (int rc = slots_ok(ctx, slots, nslot, cin, "conv2d_dgrad_bnred")) return rc;
    FvBnRed b{bn_z, scale, shift, mean, invstd, slots, nslot, leaky};
    return fv_op_conv_dgrad(ctx, dy, w_t, B, H, W, cin, cout_pad, ksize, stride, addend, dx, &b);
}

int fv_bn_bwd_slots(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift, const float* mean,
                    const float* invstd, int64_t rows, int C, float leaky, double* slots, int nslot, int reduced, float* dbeta,
                    float* dgamma, float* dz) {
    if (!ctx) return FV_ERR_INVALID;
    if (int rc = slots_ok(ctx, slots, nslot, C, "bn_bwd_slots")) return rc;
    FV_REQUIRE(ctx, g && z && scale && shift && mean && invstd && dbeta && dgamma && dz && rows > 0, "bn_bwd_slots: NULL buffer");
    return fv_ew_bn_bwd(ctx, g, z, scale, shift, mean, invstd, rows, C, leaky, nullptr, nullptr, dbeta, dgamma, dz, slots, nslot,
                        reduced != 0);
}

int fv_mse_loss_grad(fv_ctx* ctx, const float* yp, const float* yt, int rows, int C, int c_pad, float* loss, float* dy,
                     float* dbias) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_ew_mse(ctx, yp, yt, rows, C, c_pad, loss, dy, dbias);
}

int fv_fd_loss_grad(fv_ctx* ctx, const float* yp, const float* yt, int cells, int c_pad, float* loss, float* dy) {
    if (!ctx) return FV_ERR_INVALID;
    return fv_ew_fd_loss(ctx, yp, yt, cells, c_pad, loss, dy);
}

int fv_adam_step(fv_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n, int64_t iteration, double lr,
                 double beta_1, double beta_2, double eps, double decay) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, params && grads && m && v && n > 0, "adam: NULL buffer");
    if (decay > 0.0) lr = lr * (1.0 / (1.0 + decay * (double)iteration));
    double t = (double)iteration + 1.0;
    double lr_t = lr * (sqrt(1.0 - pow(beta_2, t)) / (1.0 - pow(beta_1, t)));
    return fv_ew_adam(ctx, params, grads, m, v, n, (float)lr_t, (float)beta_1, (float)beta_2, (float)eps);
}

int fv_scale(fv_ctx* ctx, float* v, int64_t n, double alpha) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, v && n >= 0, "scale: NULL buffer");
    return fv_ew_scale(ctx, v, n, (float)alpha);
}

}  // extern "C"
