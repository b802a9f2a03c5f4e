// Letterbox preprocessing on the device: uint8 HxWx3 image -> float32 SxSx3 in [0,1]
// (bicubic resize of the /255 image to (w_p, h_p), zero padding to S x S).
//
// Replaces the cv2.resize(INTER_CUBIC) + cv2.copyMakeBorder pair of the reference
// (face_detection.py:112-147 train, 656-694 evaluate, 798-835 test).  Geometry (target size, pad
// split with the odd row/column at the bottom/right) is exact; pixel values follow OpenCV's
// bicubic kernel (a = -0.75, half-pixel centres, replicated border) in float32 -- "parity
// unpinned" against cv2 itself, which is not installed here.
// HBM-bound: reads each source pixel ~(scale^2 x 16) times through L1/L2, writes 12 B per pixel.
#include "common.h"

namespace {

__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float a = -0.75f;
    w[0] = ((a * (t + 1.f) - 5.f * a) * (t + 1.f) + 8.f * a) * (t + 1.f) - 4.f * a;
    w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
    w[2] = ((a + 2.f) * (1.f - t) - (a + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void letterbox_kernel(const unsigned char* __restrict__ src, int h, int w, int S, int w_p,
                                                        int h_p, int pad_t, int pad_l, float* __restrict__ dst) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= S || y >= S) return;
    float r = 0.f, g = 0.f, b = 0.f;
    const int xi = x - pad_l, yi = y - pad_t;
    if (xi >= 0 && xi < w_p && yi >= 0 && yi < h_p) {
        // source coordinates in double: at 1080p an fp32 coordinate already costs 5e-5 in the weights
        const double fx = (xi + 0.5) * ((double)w / (double)w_p) - 0.5;
        const double fy = (yi + 0.5) * ((double)h / (double)h_p) - 0.5;
        const int sx = (int)floor(fx), sy = (int)floor(fy);
        float wx[4], wy[4];
        cubic_w((float)(fx - sx), wx);
        cubic_w((float)(fy - sy), wy);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(sy - 1 + j, 0), h - 1);
            float rr = 0.f, gg = 0.f, bb = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = min(max(sx - 1 + i, 0), w - 1);
                const unsigned char* p = src + ((size_t)yy * w + xx) * 3;
                rr += wx[i] * (float)p[0]; gg += wx[i] * (float)p[1]; bb += wx[i] * (float)p[2];
            }
            r += wy[j] * rr; g += wy[j] * gg; b += wy[j] * bb;
        }
        r *= (1.0f / 255.0f); g *= (1.0f / 255.0f); b *= (1.0f / 255.0f);
    }
    float* o = dst + ((size_t)y * S + x) * 3;
    o[0] = r; o[1] = g; o[2] = b;
}

// the same for a whole batch in ONE launch: blockIdx.z = image; the images' bytes sit back to back in one
// buffer (one host-to-device copy per batch), their geometry comes in the kernel arguments
constexpr int LB_MAX = 64;
struct LbTable { long long off[LB_MAX]; int h[LB_MAX], w[LB_MAX], w_p[LB_MAX], h_p[LB_MAX], pad_t[LB_MAX], pad_l[LB_MAX]; };

__global__ __launch_bounds__(256) void letterbox_batch_kernel(const unsigned char* __restrict__ packed, LbTable t, int S,
                                                              float* __restrict__ dst) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= S || y >= S) return;
    const unsigned char* __restrict__ src = packed + t.off[b];
    const int h = t.h[b], w = t.w[b], w_p = t.w_p[b], h_p = t.h_p[b];
    float r = 0.f, g = 0.f, bl = 0.f;
    const int xi = x - t.pad_l[b], yi = y - t.pad_t[b];
    if (xi >= 0 && xi < w_p && yi >= 0 && yi < h_p) {
        const double fx = (xi + 0.5) * ((double)w / (double)w_p) - 0.5;
        const double fy = (yi + 0.5) * ((double)h / (double)h_p) - 0.5;
        const int sx = (int)floor(fx), sy = (int)floor(fy);
        float wx[4], wy[4];
        cubic_w((float)(fx - sx), wx);
        cubic_w((float)(fy - sy), wy);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(sy - 1 + j, 0), h - 1);
            float rr = 0.f, gg = 0.f, bb = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = min(max(sx - 1 + i, 0), w - 1);
                const unsigned char* p = src + ((size_t)yy * w + xx) * 3;
                rr += wx[i] * (float)p[0]; gg += wx[i] * (float)p[1]; bb += wx[i] * (float)p[2];
            }
            r += wy[j] * rr; g += wy[j] * gg; bl += wy[j] * bb;
        }
        r *= (1.0f / 255.0f); g *= (1.0f / 255.0f); bl *= (1.0f / 255.0f);
    }
    float* o = dst + (((size_t)b * S + y) * S + x) * 3;
    o[0] = r; o[1] = g; o[2] = bl;
}

bool lb_geometry(int h, int w, int S, int* g) {
    int w_p, h_p, pad_t = 0, pad_b = 0, pad_l = 0, pad_r = 0;
    if (w >= h) {   // face_detection.py:120-133
        w_p = S; h_p = (int)((double)h / (double)w * S);
        int pad = S - h_p; pad_t = pad / 2; pad_b = pad - pad_t;
    } else {        // face_detection.py:134-147
        h_p = S; w_p = (int)((double)w / (double)h * S);
        int pad = S - w_p; pad_l = pad / 2; pad_r = pad - pad_l;
    }
    g[0] = w_p; g[1] = h_p; g[2] = pad_t; g[3] = pad_b; g[4] = pad_l; g[5] = pad_r;
    return w_p >= 1 && h_p >= 1;
}

}  // namespace

extern "C" int fv_letterbox_batch(fv_ctx* ctx, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                                  int image_size, float* dst, int32_t* geom) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, packed && offsets && hw && dst && n >= 1 && image_size >= 1, "letterbox_batch: bad arguments");
    const int S = image_size;
    for (int b0 = 0; b0 < n; b0 += LB_MAX) {
        const int nb = n - b0 < LB_MAX ? n - b0 : LB_MAX;
        LbTable t{};
        double bytes = 0.0;
        for (int i = 0; i < nb; ++i) {
            const int h = hw[2 * (b0 + i)], w = hw[2 * (b0 + i) + 1];
            int g[6];
            FV_REQUIRE(ctx, h >= 1 && w >= 1 && offsets[b0 + i] >= 0, "letterbox_batch: bad image %d", b0 + i);
            FV_REQUIRE(ctx, lb_geometry(h, w, S, g), "letterbox_batch: image %d too elongated for image_size %d", b0 + i, S);
            t.off[i] = offsets[b0 + i]; t.h[i] = h; t.w[i] = w; t.w_p[i] = g[0]; t.h_p[i] = g[1]; t.pad_t[i] = g[2]; t.pad_l[i] = g[4];
            if (geom) for (int k = 0; k < 6; ++k) geom[6 * (b0 + i) + k] = g[k];
            bytes += (double)h * w * 3 + 12.0 * S * S;
        }
        FvProfScope ps(ctx, "letterbox_batch_kernel", 0.0, bytes);
        hipLaunchKernelGGL(letterbox_batch_kernel, dim3((S + 15) / 16, (S + 15) / 16, nb), dim3(256), 0, ctx->stream, packed, t, S,
                           dst + (size_t)b0 * S * S * 3);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

extern "C" int fv_letterbox(fv_ctx* ctx, const uint8_t* src, int h, int w, int image_size, float* dst, int32_t* geom) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, src && dst && h >= 1 && w >= 1 && image_size >= 1, "letterbox: bad arguments");
    const int S = image_size;
    int w_p, h_p, pad_t = 0, pad_b = 0, pad_l = 0, pad_r = 0;
    if (w >= h) {   // face_detection.py:120-133
        w_p = S; h_p = (int)((double)h / (double)w * S);
        int pad = S - h_p; pad_t = pad / 2; pad_b = pad - pad_t;
    } else {        // face_detection.py:134-147
        h_p = S; w_p = (int)((double)w / (double)h * S);
        int pad = S - w_p; pad_l = pad / 2; pad_r = pad - pad_l;
    }
    FV_REQUIRE(ctx, w_p >= 1 && h_p >= 1, "letterbox: image too elongated for image_size %d", S);
    if (geom) { geom[0] = w_p; geom[1] = h_p; geom[2] = pad_t; geom[3] = pad_b; geom[4] = pad_l; geom[5] = pad_r; }
    FvProfScope ps(ctx, "letterbox_kernel", 0.0, (double)h * w * 3 + 12.0 * S * S);
    hipLaunchKernelGGL(letterbox_kernel, dim3((S + 15) / 16, (S + 15) / 16), dim3(256), 0, ctx->stream, src, h, w, S, w_p, h_p,
                       pad_t, pad_l, dst);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
