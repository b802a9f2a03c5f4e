// HBM-bound companions of the conv kernels: training-mode BatchNorm statistics / normalise /
// LeakyReLU / residual add (reference yolov3_detect.py:212-215), their backward, the MSE loss and
// its gradient (reference face_detection.py:381), the Keras-formula Adam update
// (face_detection.py:376-379) and the weight-layout transforms the data-gradient needs.
// All tensors NHWC float32; 16-byte vector accesses; reductions in fixed order, except the BN accumulator
// slots of the training step, which are filled with fp64 atomics (fp32 partials are exact in fp64; only
// the order of the fp64 additions varies, far below fp32 resolution -- see DESIGN.md 4.3).
#include <string>
#include "elementwise.h"

namespace {


// ---------------------------------------------------------------- BN finalize (forward, training)
// partial sums [mtiles][C] (from the conv epilogue) -> mean, invstd, scale, shift, moving stats.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psq,
                                                           int mtiles, int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float ema_old, float ema_new,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                           float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                           float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    // block = 8 channels x 128 row lanes (many blocks even for 32-channel layers with 27k partial rows)
    __shared__ double ssum[128][9], ssq[128][9];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int r = rl; r < mtiles; r += 128) { s += (double)psum[(size_t)r * C + c]; q += (double)psq[(size_t)r * C + c]; }
    }
    ssum[rl][cl] = s; ssq[rl][cl] = q;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {   // fixed-order tree: deterministic
        if (rl < st) { ssum[rl][cl] += ssum[rl + st][cl]; ssq[rl][cl] += ssq[rl + st][cl]; }
        __syncthreads();
    }
    if (rl == 0 && c < C) {
        s = ssum[0][cl]; q = ssq[0][cl];
        double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma[c];
        float sc = g * invstd;
        mean_out[c] = (float)mean; invstd_out[c] = invstd;
        scale_out[c] = sc; shift_out[c] = beta[c] - (float)mean * sc;
        if (moving_mean) {
            // Keras 2.2.4 BatchNormalization: EMA of batch mean and of var * n/(n-(1+eps))
            double corr = count / (count - (1.0 + (double)eps));
            moving_mean[c] = ema_old * moving_mean[c] + ema_new * (float)mean;
            moving_var[c] = ema_old * moving_var[c] + ema_new * (float)(var * corr);
        }
    }
}

// inference: fold moving statistics into scale/shift
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float sc = gamma[c] / sqrtf(var[c] + eps);
        scale[c] = sc; shift[c] = beta[c] - mean[c] * sc;
    }
}

// all BN layers at once: thread = one channel of the whole network (channel offsets are mean_off/2)
struct FoldTable { int n; int ch_begin[64]; int gamma_off[64]; int beta_off[64]; int mean_off[64]; int var_off[64]; };
__global__ void bn_fold_all_kernel(const float* __restrict__ params, const float* __restrict__ state, FoldTable t, float eps,
                                   int total, float* __restrict__ scale, float* __restrict__ shift) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (t.ch_begin[mid] <= g) lo = mid; else hi = mid - 1; }
    const int c = g - t.ch_begin[lo];
    float sc = params[t.gamma_off[lo] + c] / sqrtf(state[t.var_off[lo] + c] + eps);
    scale[g] = sc;
    shift[g] = params[t.beta_off[lo] + c] - state[t.mean_off[lo] + c] * sc;
}

// ---------------------------------------------------------------- y = leaky(z*scale+shift) (+ skip)
__global__ __launch_bounds__(256) void bn_act_kernel(const float4* __restrict__ z, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float4* __restrict__ skip,
                                                     float4* __restrict__ out, long long n4, int C, float leaky) {
    const int c4n = C >> 2;
    // channel group of element i, advanced incrementally (one 64-bit modulo per thread, not per element)
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cstep = (int)(stride % c4n);
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    int cg = (int)(i % c4n);
    for (; i < n4; i += stride, cg += cstep, cg -= cg >= c4n ? c4n : 0) {
        const int c = cg << 2;
        float4 v = z[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        v.x = v.x > 0.f ? v.x : v.x * leaky; v.y = v.y > 0.f ? v.y : v.y * leaky;
        v.z = v.z > 0.f ? v.z : v.z * leaky; v.w = v.w > 0.f ? v.w : v.w * leaky;
        if (skip) { float4 s = skip[i]; v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
        out[i] = v;
    }
}

// The same pass fed by the conv epilogue's accumulator slots (conv.h stat_slots): every workgroup first
// turns the slots into scale/shift for all C channels (fixed summation order over the slots, fp64,
// 16 loads per thread) and keeps them in LDS; workgroup 0 also publishes mean / invstd / scale / shift
// for the backward pass and updates the moving statistics -- no finalize launch in between.
__global__ __launch_bounds__(256) void bn_act_stats_kernel(const float4* __restrict__ z, const double* __restrict__ slots, int nslot,
                                                           double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float ema_old, float ema_new, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                           float* __restrict__ shift_out, float* __restrict__ moving_mean,
                                                           float* __restrict__ moving_var, const float4* __restrict__ skip,
                                                           float4* __restrict__ out, long long n4, int C, float leaky) {
    __shared__ __attribute__((aligned(16))) float s_sc[1024], s_sh[1024];
    __shared__ double s_part[2][256];
    const int tid = threadIdx.x;
    auto finish = [&](int c, double s, double q) {
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd, sh = beta[c] - (float)mean * sc;
        s_sc[c] = sc; s_sh[c] = sh;
        if (blockIdx.x == 0) {
            mean_out[c] = (float)mean; invstd_out[c] = invstd; scale_out[c] = sc; shift_out[c] = sh;
            if (moving_mean) {   // Keras 2.2.4 BatchNormalization: EMA of batch mean and of var * n/(n-(1+eps))
                const double corr = count / (count - (1.0 + (double)eps));
                moving_mean[c] = ema_old * moving_mean[c] + ema_new * (float)mean;
                moving_var[c] = ema_old * moving_var[c] + ema_new * (float)(var * corr);
            }
        }
    };
    if (C >= 256) {
        for (int c = tid; c < C; c += 256) {
            double s = 0.0, q = 0.0;
            for (int k = 0; k < nslot; ++k) { s += slots[(size_t)(2 * k) * C + c]; q += slots[(size_t)(2 * k + 1) * C + c]; }
            finish(c, s, q);
        }
    } else {
        // C < 256 (a power of two >= 32 here): 256 / C thread groups share the slots of a channel
        const int G = 256 / C, g = tid / C, c = tid % C;
        double s = 0.0, q = 0.0;
        if (g < G)
            for (int k = g; k < nslot; k += G) { s += slots[(size_t)(2 * k) * C + c]; q += slots[(size_t)(2 * k + 1) * C + c]; }
        s_part[0][tid] = s; s_part[1][tid] = q;
        __syncthreads();
        if (tid < C) {
            s = 0.0; q = 0.0;
            for (int j = 0; j < G; ++j) { s += s_part[0][j * C + tid]; q += s_part[1][j * C + tid]; }
            finish(tid, s, q);
        }
    }
    __syncthreads();
    const int c4n = C >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cstep = (int)(stride % c4n);
    long long i = blockIdx.x * (long long)blockDim.x + tid;
    int cg = (int)(i % c4n);
    for (; i < n4; i += stride, cg += cstep, cg -= cg >= c4n ? c4n : 0) {
        const int c = cg << 2;
        float4 v = z[i];
        const float4 sc = *reinterpret_cast<const float4*>(s_sc + c), sh = *reinterpret_cast<const float4*>(s_sh + c);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        v.x = v.x > 0.f ? v.x : v.x * leaky; v.y = v.y > 0.f ? v.y : v.y * leaky;
        v.z = v.z > 0.f ? v.z : v.z * leaky; v.w = v.w > 0.f ? v.w : v.w * leaky;
        if (skip) { float4 sk = skip[i]; v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w; }
        out[i] = v;
    }
}

// ---------------------------------------------------------------- BN + leaky backward
// pass 1: per-channel partial sums of gy and gy*xhat over a chunk of rows; gy = g * leaky'(y)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            long long M, int C, int rows_per_block, float leaky,
                                                            float* __restrict__ pdb, float* __restrict__ pdg,
                                                            double* __restrict__ slots, int nslot) {
    // thread layout: tpr = min(C/4, 256) threads per row, 256/tpr rows in flight
    __shared__ float4 sdb[256], sdg[256];
    const int c4n = C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rpi = 256 / tpr;
    const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int cb = cl; cb < c4n; cb += tpr) {
        const int c = cb << 2;
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        float4 db = make_float4(0.f, 0.f, 0.f, 0.f), dg = db;
#pragma unroll 4
        for (long long r = r0 + rl; r < r1; r += rpi) {
            const float4 gv = *reinterpret_cast<const float4*>(g + r * C + c);
            const float4 zv = *reinterpret_cast<const float4*>(z + r * C + c);
            float gy;
            gy = (zv.x * sc.x + sh.x) > 0.f ? gv.x : gv.x * leaky; db.x += gy; dg.x += gy * ((zv.x - mu.x) * is.x);
            gy = (zv.y * sc.y + sh.y) > 0.f ? gv.y : gv.y * leaky; db.y += gy; dg.y += gy * ((zv.y - mu.y) * is.y);
            gy = (zv.z * sc.z + sh.z) > 0.f ? gv.z : gv.z * leaky; db.z += gy; dg.z += gy * ((zv.z - mu.z) * is.z);
            gy = (zv.w * sc.w + sh.w) > 0.f ? gv.w : gv.w * leaky; db.w += gy; dg.w += gy * ((zv.w - mu.w) * is.w);
        }
        sdb[threadIdx.x] = db; sdg[threadIdx.x] = dg;
        __syncthreads();
        if (rl == 0) {
            for (int k = 1; k < rpi; ++k) {
                float4 b = sdb[k * tpr + cl], d = sdg[k * tpr + cl];
                db.x += b.x; db.y += b.y; db.z += b.z; db.w += b.w;
                dg.x += d.x; dg.y += d.y; dg.z += d.z; dg.w += d.w;
            }
            if (slots) {   // [nslot][2][C] fp64 accumulators; the apply pass sums them itself (no finalize launch)
                double* sl = slots + (size_t)(blockIdx.x % nslot) * 2 * C + c;
                unsafeAtomicAdd(sl + 0, (double)db.x); unsafeAtomicAdd(sl + 1, (double)db.y);
                unsafeAtomicAdd(sl + 2, (double)db.z); unsafeAtomicAdd(sl + 3, (double)db.w);
                unsafeAtomicAdd(sl + C + 0, (double)dg.x); unsafeAtomicAdd(sl + C + 1, (double)dg.y);
                unsafeAtomicAdd(sl + C + 2, (double)dg.z); unsafeAtomicAdd(sl + C + 3, (double)dg.w);
            } else {
                *reinterpret_cast<float4*>(pdb + (size_t)blockIdx.x * C + c) = db;
                *reinterpret_cast<float4*>(pdg + (size_t)blockIdx.x * C + c) = dg;
            }
        }
        __syncthreads();
    }
}

// pass 1b: reduce the chunk partials in double -> dbeta, dgamma (written into the flat grad vector)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ pdb, const float* __restrict__ pdg,
                                                               int chunks, int C, float* __restrict__ dbeta, float* __restrict__ dgamma) {
    __shared__ double s1[128][9], s2[128][9];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int r = rl; r < chunks; r += 128) { a += (double)pdb[(size_t)r * C + c]; b += (double)pdg[(size_t)r * C + c]; }
    }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (rl < st) { s1[rl][cl] += s1[rl + st][cl]; s2[rl][cl] += s2[rl + st][cl]; }
        __syncthreads();
    }
    if (rl == 0 && c < C) { dbeta[c] = (float)s1[0][cl]; dgamma[c] = (float)s2[0][cl]; }
}

// pass 2: dz = scale * (gy - dbeta/M - xhat * dgamma/M)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4* __restrict__ g, const float4* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                           float inv_count, long long n4, int C, float leaky, float4* __restrict__ dz) {
    const int c4n = C >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cstep = (int)(stride % c4n);
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    int cg = (int)(i % c4n);
    for (; i < n4; i += stride, cg += cstep, cg -= cg >= c4n ? c4n : 0) {
        const int c = cg << 2;
        const float4 gv = g[i], zv = z[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 db = *reinterpret_cast<const float4*>(dbeta + c), dg = *reinterpret_cast<const float4*>(dgamma + c);
        float4 o;
        float gy;
        gy = (zv.x * sc.x + sh.x) > 0.f ? gv.x : gv.x * leaky; o.x = sc.x * (gy - db.x * inv_count - (zv.x - mu.x) * is.x * (dg.x * inv_count));
        gy = (zv.y * sc.y + sh.y) > 0.f ? gv.y : gv.y * leaky; o.y = sc.y * (gy - db.y * inv_count - (zv.y - mu.y) * is.y * (dg.y * inv_count));
        gy = (zv.z * sc.z + sh.z) > 0.f ? gv.z : gv.z * leaky; o.z = sc.z * (gy - db.z * inv_count - (zv.z - mu.z) * is.z * (dg.z * inv_count));
        gy = (zv.w * sc.w + sh.w) > 0.f ? gv.w : gv.w * leaky; o.w = sc.w * (gy - db.w * inv_count - (zv.w - mu.w) * is.w * (dg.w * inv_count));
        dz[i] = o;
    }
}

// pass 2 fed by accumulator slots: every workgroup sums the slots of all channels first (as
// bn_act_stats_kernel does); workgroup 0 writes d-beta / d-gamma into the gradient vector
__global__ __launch_bounds__(256) void bn_bwd_apply_slots_kernel(const float4* __restrict__ g, const float4* __restrict__ z,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const double* __restrict__ slots, int nslot,
                                                                 float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                 float inv_count, long long n4, int C, float leaky, float4* __restrict__ dz) {
    __shared__ __attribute__((aligned(16))) float s_db[1024], s_dg[1024];
    __shared__ double s_part[2][256];
    const int tid = threadIdx.x;
    auto finish = [&](int c, double a, double b) {
        s_db[c] = (float)a; s_dg[c] = (float)b;
        if (blockIdx.x == 0) { dbeta[c] = (float)a; dgamma[c] = (float)b; }
    };
    if (C >= 256) {
        for (int c = tid; c < C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < nslot; ++k) { a += slots[(size_t)(2 * k) * C + c]; b += slots[(size_t)(2 * k + 1) * C + c]; }
            finish(c, a, b);
        }
    } else {
        const int G = 256 / C, gi = tid / C, c = tid % C;
        double a = 0.0, b = 0.0;
        for (int k = gi; k < nslot; k += G) { a += slots[(size_t)(2 * k) * C + c]; b += slots[(size_t)(2 * k + 1) * C + c]; }
        s_part[0][tid] = a; s_part[1][tid] = b;
        __syncthreads();
        if (tid < C) {
            a = 0.0; b = 0.0;
            for (int j = 0; j < G; ++j) { a += s_part[0][j * C + tid]; b += s_part[1][j * C + tid]; }
            finish(tid, a, b);
        }
    }
    __syncthreads();
    const int c4n = C >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cstep = (int)(stride % c4n);
    long long i = blockIdx.x * (long long)blockDim.x + tid;
    int cg = (int)(i % c4n);
    for (; i < n4; i += stride, cg += cstep, cg -= cg >= c4n ? c4n : 0) {
        const int c = cg << 2;
        const float4 gv = g[i], zv = z[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 db = *reinterpret_cast<const float4*>(s_db + c), dg = *reinterpret_cast<const float4*>(s_dg + c);
        float4 o;
        float gy;
        gy = (zv.x * sc.x + sh.x) > 0.f ? gv.x : gv.x * leaky; o.x = sc.x * (gy - db.x * inv_count - (zv.x - mu.x) * is.x * (dg.x * inv_count));
        gy = (zv.y * sc.y + sh.y) > 0.f ? gv.y : gv.y * leaky; o.y = sc.y * (gy - db.y * inv_count - (zv.y - mu.y) * is.y * (dg.y * inv_count));
        gy = (zv.z * sc.z + sh.z) > 0.f ? gv.z : gv.z * leaky; o.z = sc.z * (gy - db.z * inv_count - (zv.z - mu.z) * is.z * (dg.z * inv_count));
        gy = (zv.w * sc.w + sh.w) > 0.f ? gv.w : gv.w * leaky; o.w = sc.w * (gy - db.w * inv_count - (zv.w - mu.w) * is.w * (dg.w * inv_count));
        dz[i] = o;
    }
}

// ---------------------------------------------------------------- MSE loss + gradient (single block, deterministic)
// y_pred/y_true [rows][C]; dy padded [rows][Cpad] with zeros beyond C; bias gradient db[C] = column sums of dy.
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ yp, const float* __restrict__ yt, int rows, int C,
                                                   int Cpad, float grad_scale, float* __restrict__ loss, float* __restrict__ dy,
                                                   float* __restrict__ dbias) {
    __shared__ double sred[1024];
    __shared__ double scol[1024];
    const long long n = (long long)rows * C;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < (long long)rows * Cpad; i += 1024) {
        int r = (int)(i / Cpad), c = (int)(i - (long long)r * Cpad);
        float d = 0.f;
        if (c < C) { float e = yp[(size_t)r * C + c] - yt[(size_t)r * C + c]; acc += (double)e * (double)e; d = e * grad_scale; }
        dy[i] = d;
    }
    sred[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(sred[0] / (double)n);
    if (dbias) {
        // column sums: thread = (row lane, column); C <= 32
        const int cl = threadIdx.x % 32, rl = threadIdx.x / 32;
        double s = 0.0;
        if (cl < C) for (int r = rl; r < rows; r += 32) s += (double)((yp[(size_t)r * C + cl] - yt[(size_t)r * C + cl]) * grad_scale);
        scol[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0 && cl < C) { for (int k = 1; k < 32; ++k) s += scol[k * 32 + cl]; dbias[cl] = (float)s; }
    }
}

// The same over many workgroups (the training step's form): a workgroup = 8 row lanes x 32 columns owns the
// rows b*8 + lane, stepping by 8*gridDim.x; it writes its rows of dy and leaves one partial (sum of squares, column
// sums, all in double) in part[b][0..32]; mse_finish_kernel adds the partials in workgroup order: deterministic.
constexpr int MSE_G = 64;
__global__ __launch_bounds__(256) void mse_part_kernel(const float* __restrict__ yp, const float* __restrict__ yt, int rows, int C,
                                                       int Cpad, float grad_scale, float* __restrict__ dy, double* __restrict__ part) {
    __shared__ double s_col[8][33];
    __shared__ double s_acc[4];
    const int tid = threadIdx.x, c = tid & 31, rl = tid >> 5;
    double acc = 0.0, col = 0.0;
    for (int r = blockIdx.x * 8 + rl; r < rows; r += 8 * gridDim.x) {
        float d = 0.f;
        if (c < C) {
            const float e = yp[(size_t)r * C + c] - yt[(size_t)r * C + c];
            acc += (double)e * (double)e;
            d = e * grad_scale;
            col += (double)d;
        }
        for (int cc = c; cc < Cpad; cc += 32) dy[(size_t)r * Cpad + cc] = cc < C ? d : 0.f;
    }
    // wave reduction of the squared error (64 lanes = 2 row lanes x 32 columns), then the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) s_acc[tid >> 6] = acc;
    s_col[rl][c] = col;
    __syncthreads();
    if (tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += s_col[k][tid];
        part[(size_t)blockIdx.x * 33 + 1 + tid] = s;
        if (tid == 0) part[(size_t)blockIdx.x * 33] = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
    }
}
__global__ __launch_bounds__(64) void mse_finish_kernel(const double* __restrict__ part, int nblk, int C, double inv_n,
                                                        float* __restrict__ loss, float* __restrict__ dbias) {
    const int t = threadIdx.x;
    if (t > 32) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 33 + t];
    if (t == 0) *loss = (float)(s * inv_n);
    else if (t - 1 < C && dbias) dbias[t - 1] = (float)s;
}

// ---------------------------------------------------------------- fd_loss (reference fd.py:59-64; defined there, never used)
// per cell: (BCE(y0,p0) + mean_{1..4} sqrt((y-p)^2) + BCE(y5,p5)) / 3 with Keras' probability-space BCE
// (p clipped to [1e-7, 1-1e-7]); loss = mean over cells.  Gradient: (p-y)/(p(1-p)) inside the clip range, 0 outside;
// -sign(y-p)/4 for the box terms (0 at equality, where TF's sqrt gradient is nan).  Single block, deterministic.
__global__ __launch_bounds__(1024) void fd_loss_kernel(const float* __restrict__ yp, const float* __restrict__ yt, int cells, int Cpad,
                                                       float* __restrict__ loss, float* __restrict__ dy) {
    __shared__ double sred[1024];
    const double eps = 1e-7;
    const double gs = 1.0 / (3.0 * (double)cells);
    double acc = 0.0;
    for (int i = threadIdx.x; i < cells; i += 1024) {
        const float* p = yp + (size_t)i * 6;
        const float* t = yt + (size_t)i * 6;
        float* g = dy + (size_t)i * Cpad;
        double cell = 0.0;
        for (int k = 0; k < 6; k += 5) {
            double raw = (double)p[k], y = (double)t[k];
            double pc = raw < eps ? eps : (raw > 1.0 - eps ? 1.0 - eps : raw);
            cell += -(y * log(pc) + (1.0 - y) * log1p(-pc));
            bool inside = raw >= eps && raw <= 1.0 - eps;
            g[k] = inside ? (float)(gs * (pc - y) / (pc * (1.0 - pc))) : 0.0f;
        }
        double l1 = 0.0;
        for (int k = 1; k < 5; ++k) {
            double d = (double)t[k] - (double)p[k];
            l1 += fabs(d);
            g[k] = (float)(gs * 0.25 * (d > 0.0 ? -1.0 : (d < 0.0 ? 1.0 : 0.0)));
        }
        cell += 0.25 * l1;
        for (int k = 6; k < Cpad; ++k) g[k] = 0.0f;
        acc += cell / 3.0;
    }
    sred[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(sred[0] / (double)cells);
}

// ---------------------------------------------------------------- Adam (Keras 2.2.4 formula)
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                   float4* __restrict__ v, long long n4, float lr_t, float b1, float b2, float eps) {
    const float ob1 = 1.0f - b1, ob2 = 1.0f - b2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pv = p[i], mv = m[i], vv = v[i];
        const float4 gv = g[i];
        mv.x = b1 * mv.x + ob1 * gv.x; mv.y = b1 * mv.y + ob1 * gv.y; mv.z = b1 * mv.z + ob1 * gv.z; mv.w = b1 * mv.w + ob1 * gv.w;
        vv.x = b2 * vv.x + ob2 * (gv.x * gv.x); vv.y = b2 * vv.y + ob2 * (gv.y * gv.y);
        vv.z = b2 * vv.z + ob2 * (gv.z * gv.z); vv.w = b2 * vv.w + ob2 * (gv.w * gv.w);
        pv.x -= lr_t * mv.x / (sqrtf(vv.x) + eps); pv.y -= lr_t * mv.y / (sqrtf(vv.y) + eps);
        pv.z -= lr_t * mv.z / (sqrtf(vv.z) + eps); pv.w -= lr_t * mv.w / (sqrtf(vv.w) + eps);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}
__global__ void adam_tail_kernel(float* p, const float* g, float* m, float* v, long long begin, long long n, float lr_t, float b1,
                                 float b2, float eps) {
    long long i = begin + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) {
        float mv = b1 * m[i] + (1.0f - b1) * g[i];
        float vv = b2 * v[i] + (1.0f - b2) * (g[i] * g[i]);
        p[i] -= lr_t * mv / (sqrtf(vv) + eps);
        m[i] = mv; v[i] = vv;
    }
}

// ---------------------------------------------------------------- weight layout transforms
// src [N][T][C] -> dst [C][T][Npad] (zero padded), tiled through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void transpose_ntc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int T,
                                                            int C, int Npad) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        int n = n0 + k, c = c0 + tx;
        tile[k][tx] = (n < N && c < C) ? src[((size_t)n * T + t) * C + c] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int c = c0 + k, n = n0 + tx;
        if (c < C && n < Npad) dst[((size_t)c * T + t) * Npad + n] = tile[tx][k];
    }
}

// every layer's [N][T][C] -> [C][T][Npad] in one launch: a block finds its layer in the table
struct TransposeTable { int n; int blk_begin[64]; long long src_off[64]; long long dst_off[64]; int N[64]; int T[64]; int C[64]; int Npad[64]; };
__global__ __launch_bounds__(256) void transpose_all_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base, TransposeTable tb) {
    __shared__ float tile[32][33];
    int lo = 0, hi = tb.n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (tb.blk_begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const int N = tb.N[lo], T = tb.T[lo], C = tb.C[lo], Npad = tb.Npad[lo];
    const float* __restrict__ src = src_base + tb.src_off[lo];
    float* __restrict__ dst = dst_base + tb.dst_off[lo];
    int b = blockIdx.x - tb.blk_begin[lo];
    const int cb = (C + 31) / 32, nb = (Npad + 31) / 32;
    const int c0 = (b % cb) * 32; b /= cb;
    const int n0 = (b % nb) * 32; const int t = b / nb;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        int n = n0 + k, c = c0 + tx;
        tile[k][tx] = (n < N && c < C) ? src[((size_t)n * T + t) * C + c] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int c = c0 + k, n = n0 + tx;
        if (c < C && n < Npad) dst[((size_t)c * T + t) * Npad + n] = tile[tx][k];
    }
}

// first-layer kernel [N][K] (K = 27) -> [N][32] zero padded (the gather conv's B operand)
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K, int Kpad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * Kpad) { int n = i / Kpad, k = i - n * Kpad; dst[i] = k < K ? src[n * K + k] : 0.0f; }
}

// out[rows][C] = in[rows][Cpad][:C]
__global__ void slice_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int C, int Cpad) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < rows * C) { long long r = i / C; int c = (int)(i - r * C); dst[i] = src[r * Cpad + c]; }
}

// out = leaky(scale * (sum over K-split partial slabs, fixed order) + shift) (+ skip); any C
// The slabs are summed in slab order (deterministic) but LOADED eight at a time: with one dependent load per addition the
// batch-1 detect path spent 0.45 of its 1.5 ms here (47 launches of ~10 us, all memory latency).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int ksplit, long long stride,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ skip, float* __restrict__ out, long long n,
                                                            int C, float leaky, int do_leaky) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        int k = 0;
        for (; k + 8 <= ksplit; k += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = slabs[(k + j) * stride + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) v += t[j];
        }
        for (; k < ksplit; ++k) v += slabs[k * stride + i];
        const int c = (int)(i % C);
        if (scale) v *= scale[c];
        if (shift) v += shift[c];
        if (do_leaky) v = v > 0.f ? v : v * leaky;
        if (skip) v += skip[i];
        out[i] = v;
    }
}
// the same on float4 pieces (n, stride and C multiples of 4; every pointer 16-byte aligned).  One piece per thread, the slabs
// loaded SIXTEEN at a time (all of a <= 16-way split in one round trip; the sum order stays slab order): the batch-1 detect
// path is one dependent chain of ~100 launches and this kernel is memory LATENCY, not bandwidth.
__global__ __launch_bounds__(256) void splitk_finish4_kernel(const float4* __restrict__ slabs, int ksplit, long long stride4,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float4* __restrict__ skip, float4* __restrict__ out, long long n4,
                                                             int C, float leaky, int do_leaky) {
    const int c4n = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
        if (skip) sk = skip[i];                        // in flight with the first round of slabs
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 16 <= ksplit; k += 16) {
            float4 t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = slabs[(k + j) * stride4 + i];
#pragma unroll
            for (int j = 0; j < 16; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
        {   // the remaining < 16 slabs, again in one round (out-of-range slots re-read the last slab and are not added)
            float4 t[16];
            const int rem = ksplit - k;
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = slabs[(k + (j < rem ? j : (rem > 0 ? rem - 1 : -k))) * stride4 + i];
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j < rem) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
        const int c = (int)(i % c4n) << 2;
        if (scale) { const float4 s = *reinterpret_cast<const float4*>(scale + c); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
        if (shift) { const float4 s = *reinterpret_cast<const float4*>(shift + c); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
        if (do_leaky) {
            v.x = v.x > 0.f ? v.x : v.x * leaky; v.y = v.y > 0.f ? v.y : v.y * leaky;
            v.z = v.z > 0.f ? v.z : v.z * leaky; v.w = v.w > 0.f ? v.w : v.w * leaky;
        }
        if (skip) { v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w; }
        out[i] = v;
    }
}

inline int grid_for(long long n, int block, int cap = 256 * 8) {
    long long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// Coefficients of moving <- ema_old * moving + ema_new * batch.  Plain EMA: (momentum, 1 - momentum).  With
// fv_set_bn_zero_debias_step(t >= 1): the update Keras 2.2.4 performs through TF 1.x
// moving_averages.assign_moving_average(..., zero_debias=True) (reference yd.py:212 BatchNormalization): a zero-initialised biased
// accumulator b_t = m b_{t-1} + (1 - m) x_t and moving_t = b_t / (1 - m^t); with b_{t-1} = moving_{t-1} (1 - m^{t-1}) that is
// ema_old = m (1 - m^{t-1}) / (1 - m^t), ema_new = (1 - m) / (1 - m^t) -- the first update replaces the stored value outright.
static inline void bn_ema_coeff(const fv_ctx* ctx, float momentum, float* c_old, float* c_new) {
    if (ctx->bn_ema_step <= 0) { *c_old = momentum; *c_new = 1.0f - momentum; return; }
    const double m = (double)momentum, t = (double)ctx->bn_ema_step;
    const double den = 1.0 - pow(m, t);
    *c_old = (float)(m * (1.0 - pow(m, t - 1.0)) / den);
    *c_new = (float)((1.0 - m) / den);
}

int fv_ew_bn_finalize(fv_ctx* ctx, const float* psum, const float* psq, int mtiles, int C, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                      float* moving_mean, float* moving_var) {
    FvProfScope ps(ctx, "bn_finalize_kernel", 0.0, 8.0 * mtiles * C);
    float ema_old, ema_new;
    bn_ema_coeff(ctx, momentum, &ema_old, &ema_new);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, ctx->stream, psum, psq, mtiles, C, count, gamma,
                       beta, eps, ema_old, ema_new, mean, invstd, scale, shift, moving_mean, moving_var);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_bn_fold(fv_ctx* ctx, const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                  float* scale, float* shift) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, gamma, beta, mean, var, eps, C, scale, shift);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_bn_act(fv_ctx* ctx, const float* z, const float* scale, const float* shift, const float* skip, float* out,
                 long long rows, int C, float leaky) {
    FV_REQUIRE(ctx, C % 4 == 0, "bn_act: C must be a multiple of 4");
    long long n4 = rows * C / 4;
    FvProfScope ps(ctx, "bn_act_kernel", 0.0, 4.0 * rows * C * (skip ? 3 : 2));
    hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(n4, 256, 256 * 16)), dim3(256), 0, ctx->stream, (const float4*)z, scale, shift,
                       (const float4*)skip, (float4*)out, n4, C, leaky);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_bn_stat_slots(int C) {
    // slots x channels = 2048 accumulator pairs (16 fp64 loads per thread in the consumer's prologue)
    int n = 2048 / (C < 1 ? 1 : C);
    return n < 2 ? 2 : (n > 64 ? 64 : n);
}

int fv_ew_bn_act_stats(fv_ctx* ctx, const float* z, const double* slots, int nslot, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                       float* moving_mean, float* moving_var, const float* skip, float* out, long long rows, int C, float leaky) {
    FV_REQUIRE(ctx, C % 4 == 0 && C <= 1024 && (C >= 256 || 256 % C == 0), "bn_act_stats: C must be a multiple of 4, <= 1024, and divide 256 when below it");
    FV_REQUIRE(ctx, nslot >= 1 && slots, "bn_act_stats: no accumulator slots");
    long long n4 = rows * C / 4;
    FvProfScope ps(ctx, "bn_act_stats_kernel", 0.0, 4.0 * rows * C * (skip ? 3 : 2));
    // 4 workgroups per CU: the slot reduction in front of the stream is paid once per workgroup (measured:
    // 4096 / 2048 / 1024 / 512 workgroups -> 2.78 / 2.57 / 2.47 / 3.03 ms per step over the 52 layers)
    float ema_old, ema_new;
    bn_ema_coeff(ctx, momentum, &ema_old, &ema_new);
    hipLaunchKernelGGL(bn_act_stats_kernel, dim3(grid_for(n4, 256, 256 * 4)), dim3(256), 0, ctx->stream, (const float4*)z, slots, nslot,
                       count, gamma, beta, eps, ema_old, ema_new, mean, invstd, scale, shift, moving_mean, moving_var, (const float4*)skip,
                       (float4*)out, n4, C, leaky);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_bn_bwd_chunks(long long rows, int C) {
    // ~2048 blocks, at least 64 rows each
    long long rpb = (rows + 2047) / 2048;
    if (rpb < 64) rpb = 64;
    return (int)((rows + rpb - 1) / rpb);
}

int fv_ew_bn_bwd(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift, const float* mean,
                 const float* invstd, long long rows, int C, float leaky, float* pdb, float* pdg, float* dbeta, float* dgamma,
                 float* dz, double* slots, int nslot, bool reduced) {
    FV_REQUIRE(ctx, C % 4 == 0, "bn_bwd: C must be a multiple of 4");
    FV_REQUIRE(ctx, !reduced || slots, "bn_bwd: a reduction done elsewhere must have gone to accumulator slots");
    FV_REQUIRE(ctx, !slots || (nslot >= 1 && C <= 1024 && (C >= 256 || 256 % C == 0)), "bn_bwd: accumulator slots need C <= 1024 dividing or divided by 256");
    long long rpb = (rows + 2047) / 2048;
    if (rpb < 64) rpb = 64;
    int chunks = (int)((rows + rpb - 1) / rpb);
    if (!reduced) {
        FvProfScope ps(ctx, "bn_bwd_reduce_kernel", 0.0, 8.0 * rows * C);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(chunks), dim3(256), 0, ctx->stream, g, z, scale, shift, mean, invstd, rows, C,
                           (int)rpb, leaky, pdb, pdg, slots, nslot);
    }
    FV_LAUNCH_CHECK(ctx);
    if (slots) {
        long long n4s = rows * C / 4;
        FvProfScope ps(ctx, "bn_bwd_apply_slots_kernel", 0.0, 12.0 * rows * C);
        hipLaunchKernelGGL(bn_bwd_apply_slots_kernel, dim3(grid_for(n4s, 256, 256 * 4)), dim3(256), 0, ctx->stream, (const float4*)g,
                           (const float4*)z, scale, shift, mean, invstd, slots, nslot, dbeta, dgamma, (float)(1.0 / (double)rows), n4s, C,
                           leaky, (float4*)dz);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, ctx->stream, pdb, pdg, chunks, C, dbeta, dgamma);
    FV_LAUNCH_CHECK(ctx);
    long long n4 = rows * C / 4;
    FvProfScope ps(ctx, "bn_bwd_apply_kernel", 0.0, 12.0 * rows * C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(n4, 256, 256 * 16)), dim3(256), 0, ctx->stream, (const float4*)g, (const float4*)z,
                       scale, shift, mean, invstd, dbeta, dgamma, (float)(1.0 / (double)rows), n4, C, leaky, (float4*)dz);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_mse_scratch_floats() { return 2 * MSE_G * 33; }

int fv_ew_mse(fv_ctx* ctx, const float* yp, const float* yt, int rows, int C, int Cpad, float* loss, float* dy, float* dbias,
              double* part, double grad_weight) {
    FV_REQUIRE(ctx, C <= 32 && Cpad >= C, "mse: C must be <= 32");
    // grad_weight: this rank's share n_r / N of a merged data-parallel batch -- it scales dy (and with it every gradient of the
    // step, all linear in dy); the loss value stays this slice's own mean
    float gs = (float)(2.0 * grad_weight / ((double)rows * C));
    if (part) {
        const int g = (rows + 7) / 8 < MSE_G ? (rows + 7) / 8 : MSE_G;
        hipLaunchKernelGGL(mse_part_kernel, dim3(g), dim3(256), 0, ctx->stream, yp, yt, rows, C, Cpad, gs, dy, part);
        FV_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, part, g, C, 1.0 / ((double)rows * C), loss, dbias);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    }
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, ctx->stream, yp, yt, rows, C, Cpad, gs, loss, dy, dbias);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

namespace {
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ v, long long n, float alpha) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) v[i] *= alpha;
}
}  // namespace
int fv_ew_scale(fv_ctx* ctx, float* v, long long n, float alpha) {
    if (n <= 0) return FV_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, v, n, alpha);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_adam(fv_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2, float eps) {
    long long n4 = n / 4;
    FvProfScope ps(ctx, "adam_kernel", 0.0, 28.0 * n);
    if (n4 > 0) {
        hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4, 256, 256 * 16)), dim3(256), 0, ctx->stream, (float4*)p, (const float4*)g,
                           (float4*)m, (float4*)v, n4, lr_t, b1, b2, eps);
        FV_LAUNCH_CHECK(ctx);
    }
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(adam_tail_kernel, dim3(1), dim3(64), 0, ctx->stream, p, g, m, v, n4 * 4, n, lr_t, b1, b2, eps);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

int fv_ew_transpose_ntc(fv_ctx* ctx, const float* src, float* dst, int N, int T, int C, int Npad) {
    FvProfScope ps(ctx, "transpose_ntc_kernel", 0.0, 4.0 * T * C * ((double)N + Npad));
    hipLaunchKernelGGL(transpose_ntc_kernel, dim3((C + 31) / 32, (Npad + 31) / 32, T), dim3(256), 0, ctx->stream, src, dst, N, T, C, Npad);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_transpose_all(fv_ctx* ctx, const float* src_base, float* dst_base, int nlayers, const long long* src_off,
                        const long long* dst_off, const int* N, const int* T, const int* C, const int* Npad) {
    FV_REQUIRE(ctx, nlayers >= 1 && nlayers <= 64, "transpose_all: 1..64 layers");
    TransposeTable tb{};
    tb.n = nlayers;
    int blocks = 0;
    double bytes = 0.0;
    for (int l = 0; l < nlayers; ++l) {
        tb.blk_begin[l] = blocks; tb.src_off[l] = src_off[l]; tb.dst_off[l] = dst_off[l];
        tb.N[l] = N[l]; tb.T[l] = T[l]; tb.C[l] = C[l]; tb.Npad[l] = Npad[l];
        blocks += ((C[l] + 31) / 32) * ((Npad[l] + 31) / 32) * T[l];
        bytes += 4.0 * T[l] * C[l] * ((double)N[l] + Npad[l]);
    }
    FvProfScope ps(ctx, "transpose_all_kernel", 0.0, bytes);
    hipLaunchKernelGGL(transpose_all_kernel, dim3(blocks), dim3(256), 0, ctx->stream, src_base, dst_base, tb);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_pad_rows(fv_ctx* ctx, const float* src, float* dst, int N, int K, int Kpad) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3((N * Kpad + 255) / 256), dim3(256), 0, ctx->stream, src, dst, N, K, Kpad);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_slice_cols(fv_ctx* ctx, const float* src, float* dst, long long rows, int C, int Cpad) {
    hipLaunchKernelGGL(slice_cols_kernel, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, rows, C, Cpad);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_splitk_finish(fv_ctx* ctx, const float* slabs, int ksplit, long long stride, const float* scale, const float* shift,
                        const float* skip, float* out, long long n, int C, float leaky, int do_leaky) {
    FvProfScope ps(ctx, "splitk_finish_kernel", "M" + std::to_string(n / C) + " N" + std::to_string(C) + " ks" + std::to_string(ksplit), 0.0,
                   4.0 * n * (ksplit + 1 + (skip ? 1 : 0)));
    const bool al16 = (((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)skip | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0;
    if ((n & 3) == 0 && (stride & 3) == 0 && (C & 3) == 0 && al16) {
        hipLaunchKernelGGL(splitk_finish4_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, ctx->stream, (const float4*)slabs, ksplit, stride / 4,
                           scale, shift, (const float4*)skip, (float4*)out, n / 4, C, leaky, do_leaky);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    }
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, slabs, ksplit, stride, scale, shift,
                       skip, out, n, C, leaky, do_leaky);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_bn_fold_all(fv_ctx* ctx, const float* params, const float* state, int nlayers, const int* ch_begin, const long long* gamma_off,
                      const long long* beta_off, const long long* mean_off, const long long* var_off, float eps, int total,
                      float* scale, float* shift) {
    FV_REQUIRE(ctx, nlayers <= 64, "bn_fold_all: too many layers");
    FoldTable t{};
    t.n = nlayers;
    for (int i = 0; i < nlayers; ++i) {
        t.ch_begin[i] = ch_begin[i]; t.gamma_off[i] = (int)gamma_off[i]; t.beta_off[i] = (int)beta_off[i];
        t.mean_off[i] = (int)mean_off[i]; t.var_off[i] = (int)var_off[i];
    }
    hipLaunchKernelGGL(bn_fold_all_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, params, state, t, eps, total, scale, shift);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_fd_loss(fv_ctx* ctx, const float* yp, const float* yt, int cells, int Cpad, float* loss, float* dy) {
    FV_REQUIRE(ctx, Cpad >= 6, "fd_loss: Cpad must be >= 6");
    hipLaunchKernelGGL(fd_loss_kernel, dim3(1), dim3(1024), 0, ctx->stream, yp, yt, cells, Cpad, loss, dy);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// UpSampling2D(2) (nearest) of src [B][Hs][Ws][C1] concatenated in front of skip [B][2Hs][2Ws][C2]
// (reference yolov3_detect.py:282-283, 298-299) -> out [B][2Hs][2Ws][C1+C2]
namespace {
__global__ __launch_bounds__(256) void upsample_concat_kernel(const float4* __restrict__ src, const float4* __restrict__ skip,
                                                              float4* __restrict__ out, int B, int Hs, int Ws, int C1, int C2) {
    const int c4 = (C1 + C2) >> 2, c14 = C1 >> 2, c24 = C2 >> 2;
    const long long n4 = (long long)B * (2 * Hs) * (2 * Ws) * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        const long long pix = i / c4;
        const int w = (int)(pix % (2 * Ws)), h = (int)((pix / (2 * Ws)) % (2 * Hs)), b = (int)(pix / ((long long)4 * Hs * Ws));
        out[i] = c < c14 ? src[(((long long)b * Hs + (h >> 1)) * Ws + (w >> 1)) * c14 + c] : skip[pix * c24 + (c - c14)];
    }
}
}  // namespace

int fv_ew_upsample_concat(fv_ctx* ctx, const float* src, const float* skip, float* out, int B, int Hs, int Ws, int C1, int C2) {
    FV_REQUIRE(ctx, C1 % 4 == 0 && C2 % 4 == 0, "upsample_concat: channels must be multiples of 4");
    const long long n4 = (long long)B * 4 * Hs * Ws * ((C1 + C2) / 4);
    FvProfScope ps(ctx, "upsample_concat_kernel", 0.0, 4.0 * B * Hs * Ws * (C1 + 8.0 * (C1 + C2) / 2 + 4.0 * C2));
    long long g = (n4 + 255) / 256;
    hipLaunchKernelGGL(upsample_concat_kernel, dim3((unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g))), dim3(256), 0, ctx->stream,
                       (const float4*)src, (const float4*)skip, (float4*)out, B, Hs, Ws, C1, C2);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}


// ---------------------------------------------------------------- three-scale training
// Backward of UpSampling2D(2) + concatenate (yd.py:282-283, 298-299): g [B][2Hs][2Ws][C1+C2] ->
//   g_up [B][Hs][Ws][C1] = sum of the 2x2 block of the first C1 channels,  g_skip [B][2Hs][2Ws][C2] = the other C2.
namespace {
__global__ __launch_bounds__(256) void upsample_concat_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ g_up,
                                                                  float4* __restrict__ g_skip, int B, int Hs, int Ws, int C1, int C2) {
    const int c4 = (C1 + C2) >> 2, c14 = C1 >> 2, c24 = C2 >> 2;
    const long long n_up = (long long)B * Hs * Ws * c14, n_sk = (long long)B * 4 * Hs * Ws * c24;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_up + n_sk; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_up) {
            const int c = (int)(i % c14);
            const long long pix = i / c14;
            const int w = (int)(pix % Ws), h = (int)((pix / Ws) % Hs), b = (int)(pix / ((long long)Hs * Ws));
            const long long r0 = (((long long)b * 2 * Hs + 2 * h) * 2 * Ws + 2 * w) * c4 + c, r1 = r0 + (long long)2 * Ws * c4;
            const float4 p = g[r0], q = g[r0 + c4], r = g[r1], t = g[r1 + c4];   // fixed order: deterministic
            float4 o;
            o.x = (p.x + q.x) + (r.x + t.x); o.y = (p.y + q.y) + (r.y + t.y); o.z = (p.z + q.z) + (r.z + t.z); o.w = (p.w + q.w) + (r.w + t.w);
            g_up[i] = o;
        } else {
            const long long j = i - n_up;
            const int c = (int)(j % c24);
            const long long pix = j / c24;
            g_skip[j] = g[pix * c4 + c14 + c];
        }
    }
}

// column sums of dy [rows][Cpad] over the first C columns -> out[C] (bias gradient of a detection conv); two stages, fixed order
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ dy, long long rows, int C, int Cpad, double* __restrict__ part) {
    __shared__ double s[8][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    double acc = 0.0;
    if (c < C)
        for (long long r = blockIdx.y * 8 + rl; r < rows; r += 8 * gridDim.y) acc += (double)dy[r * Cpad + c];
    s[rl][threadIdx.x & 31] = acc;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s[k][threadIdx.x];
        part[(size_t)blockIdx.y * C + c] = t;
    }
}
__global__ void colsum_finish_kernel(const double* __restrict__ part, int nchunk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (int k = 0; k < nchunk; ++k) t += part[(size_t)k * C + c];
    out[c] = (float)t;
}

// Detection loss of one scale -- the reference defines none for its three-scale graph (yd.py:217-311 is inference only); this
// generalises its fd_loss (fd.py:59-64: (BCE(objectness) + mean |box error| + BCE(class)) / 3 per cell) to 3 anchors per cell
// and `ncls` classes, with the cross-entropies taken on LOGITS (the head is linear; fd_loss as written feeds a linear output
// to a probability-space BCE):  per (cell, anchor)  ( bce(t4, y4) + mean_{k<4} |t_k - y_k| + mean_c bce(t_{5+c}, y_{5+c}) ) / 3,
// bce(t, y) = max(t, 0) - t*y + log1p(exp(-|t|));  scale loss = mean over cells x anchors.  dy = its gradient, zero padded.
__device__ __forceinline__ double bce_logit(double t, double y) { return fmax(t, 0.0) - t * y + log1p(exp(-fabs(t))); }
__global__ __launch_bounds__(256) void yolo_loss_part_kernel(const float* __restrict__ t, const float* __restrict__ y, long long nbox,
                                                             int ncls, int A, int Cpad, double grad_weight, float* __restrict__ dy,
                                                             double* __restrict__ part) {
    // one wave per (cell, anchor) box: lanes stride over the 5 + ncls entries, wave-reduce the box loss
    __shared__ double s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = 5 + ncls;
    const double gs = grad_weight / (3.0 * (double)nbox);      // grad_weight: fv_ew_mse
    double acc = 0.0;
    for (long long bx = (long long)blockIdx.x * 4 + wave; bx < nbox; bx += (long long)gridDim.x * 4) {
        const long long cell = bx / A;
        const int an = (int)(bx - cell * A);
        const float* tp = t + cell * (long long)(A * E) + (long long)an * E;
        const float* yp = y + cell * (long long)(A * E) + (long long)an * E;
        float* gp = dy + cell * Cpad + (long long)an * E;
        double l = 0.0;
        for (int e = lane; e < E; e += 64) {
            const double tv = (double)tp[e], yv = (double)yp[e];
            double g;
            if (e < 4) { const double d = tv - yv; l += 0.25 * fabs(d); g = 0.25 * (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0)); }
            else {
                const double w = e == 4 ? 1.0 : 1.0 / (double)ncls;
                l += w * bce_logit(tv, yv);
                g = w * (1.0 / (1.0 + exp(-tv)) - yv);
            }
            gp[e] = (float)(g * gs);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        acc += l / 3.0;
        if (an == A - 1) for (int e = A * E + lane; e < Cpad; e += 64) dy[cell * Cpad + e] = 0.f;   // padding columns
    }
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ void yolo_loss_finish_kernel(const double* __restrict__ part, int n0, int n1, int n2, double inv0, double inv1, double inv2,
                                        float* __restrict__ loss) {
    if (threadIdx.x || blockIdx.x) return;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = 0; i < n0; ++i) a += part[i];
    for (int i = 0; i < n1; ++i) b += part[n0 + i];
    for (int i = 0; i < n2; ++i) c += part[n0 + n1 + i];
    *loss = (float)(a * inv0 + b * inv1 + c * inv2);
}
}  // namespace

int fv_ew_upsample_concat_bwd(fv_ctx* ctx, const float* g, float* g_up, float* g_skip, int B, int Hs, int Ws, int C1, int C2) {
    FV_REQUIRE(ctx, C1 % 4 == 0 && C2 % 4 == 0, "upsample_concat_bwd: channels must be multiples of 4");
    const long long n4 = (long long)B * Hs * Ws * (C1 / 4) + (long long)B * 4 * Hs * Ws * (C2 / 4);
    FvProfScope ps(ctx, "upsample_concat_bwd_kernel", 0.0, 4.0 * B * Hs * Ws * (5.0 * C1 + 8.0 * C2));
    long long gr = (n4 + 255) / 256;
    hipLaunchKernelGGL(upsample_concat_bwd_kernel, dim3((unsigned)(gr > 4096 ? 4096 : (gr < 1 ? 1 : gr))), dim3(256), 0, ctx->stream,
                       (const float4*)g, (float4*)g_up, (float4*)g_skip, B, Hs, Ws, C1, C2);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_colsum_chunks(long long rows) { long long n = (rows + 511) / 512; return (int)(n < 1 ? 1 : (n > 64 ? 64 : n)); }
int fv_ew_colsum(fv_ctx* ctx, const float* dy, long long rows, int C, int Cpad, double* part, float* out) {
    const int nchunk = fv_ew_colsum_chunks(rows);
    hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 31) / 32, nchunk), dim3(256), 0, ctx->stream, dy, rows, C, Cpad, part);
    FV_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, part, nchunk, C, out);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_ew_yolo_loss_blocks(long long nbox) { long long n = (nbox + 3) / 4; return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n)); }
int fv_ew_yolo_loss_part(fv_ctx* ctx, const float* t, const float* y, long long cells, int ncls, int A, int Cpad, float* dy, double* part,
                         double grad_weight) {
    FV_REQUIRE(ctx, Cpad >= A * (5 + ncls), "yolo_loss: Cpad too small");
    const long long nbox = cells * A;
    hipLaunchKernelGGL(yolo_loss_part_kernel, dim3(fv_ew_yolo_loss_blocks(nbox)), dim3(256), 0, ctx->stream, t, y, nbox, ncls, A, Cpad, grad_weight, dy, part);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
int fv_ew_yolo_loss_finish(fv_ctx* ctx, const double* part, const long long* cells3, int A, float* loss) {
    const int n0 = fv_ew_yolo_loss_blocks(cells3[0] * A), n1 = fv_ew_yolo_loss_blocks(cells3[1] * A), n2 = fv_ew_yolo_loss_blocks(cells3[2] * A);
    hipLaunchKernelGGL(yolo_loss_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, part, n0, n1, n2, 1.0 / ((double)cells3[0] * A),
                       1.0 / ((double)cells3[1] * A), 1.0 / ((double)cells3[2] * A), loss);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
