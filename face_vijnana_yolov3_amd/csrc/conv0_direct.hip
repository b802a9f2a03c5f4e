// First layer of Darknet-53 (reference yolov3_detect.py:221 conv_0: 3x3, stride 1, 3 -> 32 channels, fd.py:408) as a
// direct vector-FMA convolution.
//
// With K = 27 this layer has 12 GFLOP at batch 40 but writes 886 MB: it is HBM-bound (floor ~0.18 ms), and the
// matrix-core gather kernel (conv_mfma.hip, GATHER) spends its time building the 27-wide K slab element by element
// (16 scalar loads + integer divisions per thread): 0.62 ms = 1.6 TB/s.  Here a workgroup owns 8 x 32 output pixels:
//  * the 10 x 34 x 3 input halo is staged in LDS with coalesced row loads (zero outside the image),
//  * a thread computes 4 consecutive pixels x 8 output channels: 15 ds_read_b128 of inputs, 54 of weights ([27][32] in
//    LDS, same address across the strips: broadcast), 864 v_fma_f32 in the SAME order as the fmaf chain of the matrix-core
//    kernel (k = tap * 3 + channel, visited 0,4,1,5,2,6,3,7 inside every group of eight: conv_mfma.hip feeds k = 4h + j
//    of a group to MFMA j from lane half h), so the result is bit-identical to the gather kernel's,
//  * the halo of the workgroup's NEXT tile is fetched into registers before the current tile is computed and written to
//    the second LDS buffer afterwards: the global-load latency hides behind the FMAs,
//  * the four channel-group lanes of a pixel write 128 contiguous bytes; stores are 16 bytes wide,
//  * training mode: per-thread column sums / sums of squares are carried over all tiles of a workgroup (persistent grid),
//    reduced through LDS once and added to the layer's fp64 accumulator slots (conv.h stat_slots);
//    inference mode: affine (+ LeakyReLU) applied on the way out.
#include "conv.h"

namespace {

constexpr int TH = 8, TW = 32;            // output tile
constexpr int HR = TH + 2, HC = (TW + 2) * 3;   // halo rows, floats per halo row (102)
constexpr int HCP = 104;                  // padded halo row (16-byte aligned rows)

template <int epi>   // FV_EPI_STATS (training) | FV_EPI_AFFINE [| FV_EPI_LEAKY] (inference) | 0: compile-time, to keep registers down
__global__ __launch_bounds__(256, 3) void conv0_direct_kernel(const float* __restrict__ x, const float* __restrict__ w32 /*[32][32], k-major per n*/,
                                                           float* __restrict__ out, int B, int H, int W,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, float leaky,
                                                           double* __restrict__ slots, int nslot) {
    __shared__ __attribute__((aligned(16))) float halo2[2][HR * HCP];
    __shared__ __attribute__((aligned(16))) float wk[27 * 32];        // [k][n]
    __shared__ double red[2][64][33];   // (only the statistics instantiation keeps it)
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * 32; i += 256) { const int k = i >> 5, n = i & 31; wk[i] = w32[n * 32 + k]; }
    const int strip = tid >> 2, q = tid & 3;          // strip of 4 pixels, channel group of 8
    const int sr = strip >> 3, sc = (strip & 7) * 4;
    const int tiles_w = W / TW, tiles_h = H / TH;
    const long long ntiles = (long long)B * tiles_h * tiles_w;
    // statistics: fp32 over the four pixels of a tile, fp64 across the tiles of this persistent workgroup (43 K values per
    // channel and workgroup at batch 256: an fp32 chain of that length would cost the mean its last digits)
    double ssum[8], ssq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ssum[c] = 0.0; ssq[c] = 0.0; }
    float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sh0;
    if (epi & FV_EPI_AFFINE) {
        if (scale) { sc0 = *reinterpret_cast<const float4*>(scale + 8 * q); sc1 = *reinterpret_cast<const float4*>(scale + 8 * q + 4); }
        if (shift) { sh0 = *reinterpret_cast<const float4*>(shift + 8 * q); sh1 = *reinterpret_cast<const float4*>(shift + 8 * q + 4); }
    }
    // halo element e of this thread (e = tid + 256 j, j < 4) of a tile: global offset, or -1 outside the image
    auto halo_fetch = [&](long long tile, float (&hv)[4]) {
        const int tw = (int)(tile % tiles_w), th = (int)((tile / tiles_w) % tiles_h), b = (int)(tile / ((long long)tiles_w * tiles_h));
        const int h0 = th * TH, w0 = tw * TW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const int r = i / HC, cc = i - r * HC;
            const int ih = h0 - 1 + r, iw3 = (w0 - 1) * 3 + cc;
            float v = 0.f;
            if (i < HR * HC && (unsigned)ih < (unsigned)H && iw3 >= 0 && iw3 < W * 3) v = x[((size_t)b * H + ih) * W * 3 + iw3];
            hv[j] = v;
        }
    };
    auto halo_store = [&](float* h, const float (&hv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            if (i < HR * HC) { const int r = i / HC; h[r * HCP + (i - r * HC)] = hv[j]; }
        }
    };
    float hv[4];
    if ((long long)blockIdx.x < ntiles) { halo_fetch(blockIdx.x, hv); halo_store(halo2[0], hv); }
    __syncthreads();                                   // halo of the first tile and wk are visible
    int cur = 0;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const int tw = (int)(tile % tiles_w), th = (int)((tile / tiles_w) % tiles_h), b = (int)(tile / ((long long)tiles_w * tiles_h));
        const int h0 = th * TH, w0 = tw * TW;
        const float* halo = halo2[cur];
        const bool more = tile + gridDim.x < ntiles;
        if (more) halo_fetch(tile + gridDim.x, hv);    // in flight while this tile computes
        float in[3][20];   // 18 used
#pragma unroll
        for (int dr = 0; dr < 3; ++dr)
#pragma unroll
            for (int v4 = 0; v4 < 5; ++v4) {
                const float4 t = *reinterpret_cast<const float4*>(&halo[(sr + dr) * HCP + sc * 3 + v4 * 4]);
                in[dr][v4 * 4] = t.x; in[dr][v4 * 4 + 1] = t.y; in[dr][v4 * 4 + 2] = t.z; in[dr][v4 * 4 + 3] = t.w;
            }
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
        // weights of step k+1 are read while step k computes; the sched_barrier keeps that shape (left alone, the
        // scheduler hoists all 54 weight reads to the top: 256 VGPRs, one wave per SIMD)
        float4 wa = *reinterpret_cast<const float4*>(&wk[8 * q]), wb = *reinterpret_cast<const float4*>(&wk[8 * q + 4]);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            // position kk of the chain -> k: inside each group of eight the order is 0,4,1,5,2,6,3,7 (see the header)
            constexpr int ORD[8] = {0, 4, 1, 5, 2, 6, 3, 7};
            const int k = (kk & ~7) + ORD[kk & 7];
            int kn = 32;                                   // next k < 27 in chain order
#pragma unroll
            for (int t = 31; t > kk; --t) { const int c = (t & ~7) + ORD[t & 7]; if (c < 27) kn = c; }
            if (k >= 27) continue;                         // zero-padded K slots of the matrix-core kernel: fma(0, 0, acc) = acc
            const int tp = k / 3, ci = k - tp * 3, dr = tp / 3, dc = tp - dr * 3;
            float4 na = wa, nb = wb;
            if (kn < 27) {
                na = *reinterpret_cast<const float4*>(&wk[kn * 32 + 8 * q]);
                nb = *reinterpret_cast<const float4*>(&wk[kn * 32 + 8 * q + 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = in[dr][(j + dc) * 3 + ci];
                acc[j][0] = fmaf(xv, wa.x, acc[j][0]); acc[j][1] = fmaf(xv, wa.y, acc[j][1]);
                acc[j][2] = fmaf(xv, wa.z, acc[j][2]); acc[j][3] = fmaf(xv, wa.w, acc[j][3]);
                acc[j][4] = fmaf(xv, wb.x, acc[j][4]); acc[j][5] = fmaf(xv, wb.y, acc[j][5]);
                acc[j][6] = fmaf(xv, wb.z, acc[j][6]); acc[j][7] = fmaf(xv, wb.w, acc[j][7]);
            }
            __builtin_amdgcn_sched_barrier(0);
            wa = na; wb = nb;
        }
        float* op = out + (((size_t)b * H + h0 + sr) * W + w0 + sc) * 32 + 8 * q;
        if (epi & FV_EPI_STATS) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float ts = 0.f, tq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ts += acc[j][c]; tq += acc[j][c] * acc[j][c]; }
                ssum[c] += (double)ts; ssq[c] += (double)tq;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v0 = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]), v1 = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
            if (epi & FV_EPI_AFFINE) {
                v0.x = v0.x * sc0.x + sh0.x; v0.y = v0.y * sc0.y + sh0.y; v0.z = v0.z * sc0.z + sh0.z; v0.w = v0.w * sc0.w + sh0.w;
                v1.x = v1.x * sc1.x + sh1.x; v1.y = v1.y * sc1.y + sh1.y; v1.z = v1.z * sc1.z + sh1.z; v1.w = v1.w * sc1.w + sh1.w;
            }
            if (epi & FV_EPI_LEAKY) {
                v0.x = v0.x > 0.f ? v0.x : v0.x * leaky; v0.y = v0.y > 0.f ? v0.y : v0.y * leaky; v0.z = v0.z > 0.f ? v0.z : v0.z * leaky; v0.w = v0.w > 0.f ? v0.w : v0.w * leaky;
                v1.x = v1.x > 0.f ? v1.x : v1.x * leaky; v1.y = v1.y > 0.f ? v1.y : v1.y * leaky; v1.z = v1.z > 0.f ? v1.z : v1.z * leaky; v1.w = v1.w > 0.f ? v1.w : v1.w * leaky;
            }
            *reinterpret_cast<float4*>(op + j * 32) = v0;
            *reinterpret_cast<float4*>(op + j * 32 + 4) = v1;
        }
        if (more) halo_store(halo2[cur ^ 1], hv);
        __syncthreads();                               // next halo visible; this tile's readers are done
    }
    if (epi & FV_EPI_STATS) {
        // 64 strips x 4 channel groups -> 32 columns: one LDS pass, then fp64 atomics to this workgroup's slot
#pragma unroll
        for (int c = 0; c < 8; ++c) { red[0][strip][8 * q + c] = ssum[c]; red[1][strip][8 * q + c] = ssq[c]; }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, n = tid & 31;
            double t = 0.0;
#pragma unroll 8
            for (int s = 0; s < 64; ++s) t += red[which][s][n];
            double* sl = slots + (size_t)(blockIdx.x % nslot) * 2 * 32;
            unsafeAtomicAdd(sl + which * 32 + n, t);
        }
    }
}

}  // namespace

// Eligible problems: Cin 3, Cout 32, 3x3 stride 1, W % 32 == 0, H % 8 == 0, statistics through slots (or none).
bool fv_conv0_direct_ok(const FvConvArgs& a) {
    return a.Cin == 3 && a.Nout == 32 && a.taps[0].n == 9 && a.is == 1 && a.os == 1 && a.nclass == 1 && a.Hl == a.Hin && a.Wl == a.Win &&
           a.Win % TW == 0 && a.Hin % TH == 0 && a.ksplit <= 1 && !(a.epi & (FV_EPI_ADD | FV_EPI_BNRED)) &&
           (!(a.epi & FV_EPI_STATS) || (a.stat_slots && a.stat_nslot >= 1)) && !((a.epi & FV_EPI_STATS) && (a.epi & (FV_EPI_AFFINE | FV_EPI_LEAKY)));
}

int fv_conv0_direct_launch(fv_ctx* ctx, const FvConvArgs& a) {
    const long long ntiles = (long long)a.B * (a.Hin / TH) * (a.Win / TW);
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);     // 4 workgroups per CU, persistent over the tiles
    FvProfScope ps(ctx, "conv0_direct_kernel", a.alg_flops, 4.0 * ((double)a.B * a.Hin * a.Win * (3 + 32)));
#define FV_C0(E) hipLaunchKernelGGL(conv0_direct_kernel<E>, dim3(grid), dim3(256), 0, ctx->stream, a.x, a.w, a.out, a.B, a.Hin, a.Win, a.scale, \
                                    a.shift, a.leaky, a.stat_slots, a.stat_nslot)
    if (a.epi & FV_EPI_STATS) FV_C0(FV_EPI_STATS);
    else if ((a.epi & FV_EPI_AFFINE) && (a.epi & FV_EPI_LEAKY)) FV_C0(FV_EPI_AFFINE | FV_EPI_LEAKY);
    else if (a.epi & FV_EPI_AFFINE) FV_C0(FV_EPI_AFFINE);
    else if (a.epi & FV_EPI_LEAKY) FV_C0(FV_EPI_LEAKY);
    else FV_C0(0);
#undef FV_C0
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
