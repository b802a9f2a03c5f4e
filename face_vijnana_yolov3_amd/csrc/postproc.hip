// Detect-path post-processing on gfx950: sigmoid + threshold + box decode + greedy NMS +
// ascending top-k, one 256-thread workgroup (4 waves) per image, everything in LDS.
//
// Replaces the NumPy/Python tail of FaceDetector.detect (reference face_detection.py:900-947,
// do_nms_v2 yolov3_detect.py:446-458, bbox_iou yolov3_detect.py:165-194).
//
// Structure per image (n = candidates, P = padded cell count 256/512):
//   1. decode every cell (double arithmetic exactly as the reference's float64 promotion)
//   2. LDS bitonic sort of 64-bit keys (score bits | ~cell)  -> descending score, ties: lower cell
//   3. suppression matrix as bitmasks: wave w owns rows w, w+4, ...; one __ballot per 64 columns
//   4. wave 0 walks the rows in order keeping the alive mask in registers (no barriers)
//   5. second bitonic sort of the survivors (ascending score) and output of the first num_cands
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_ref(float x) {
    // correctly-rounded float32 exp, then IEEE float32 add / divide (see oracle/postproc_oracle.c)
    float e = (float)exp(-(double)x);
    return 1.0f / (1.0f + e);
}

__device__ __forceinline__ int interval_overlap(int x1, int x2, int x3, int x4) {
    if (x3 < x1) {
        if (x4 < x1) return 0;
        return min(x2, x4) - x1;
    } else {
        if (x2 < x3) return 0;
        return min(x2, x4) - x3;
    }
}

// iou(a,b) >= th with the reference's semantics (0/0 = nan compares false).
__device__ __forceinline__ bool iou_ge(const int4& a, const int4& b, double th) {
    long long iw = interval_overlap(a.x, a.z, b.x, b.z);
    long long ih = interval_overlap(a.y, a.w, b.y, b.w);
    long long inter = iw * ih;
    long long uni = (long long)(a.z - a.x) * (a.w - a.y) + (long long)(b.z - b.x) * (b.w - b.y) - inter;
    if (inter == 0) return uni != 0 && 0.0 >= th;  // == (0.0 / uni >= th), without the divide
    return (double)inter / (double)uni >= th;
}

template <int P, bool DESC>
__device__ __forceinline__ void bitonic_sort(unsigned long long* keys, int tid) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += 256) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i + j;
                unsigned long long a = keys[i], b = keys[l];
                bool up = ((i & k) == 0) != DESC;  // ascending run?
                if (up ? (a > b) : (a < b)) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
}

template <int P>
__global__ __launch_bounds__(256) void decode_nms_kernel(const float* __restrict__ head, int grid, int S,
                                                         double conf_th, double iou_th, int num_cands,
                                                         int* __restrict__ boxes, int* __restrict__ cell_out,
                                                         float* __restrict__ obj_out, float* __restrict__ score_out,
                                                         int* __restrict__ count_out) {
    constexpr int NW = P / 64;
    __shared__ unsigned long long keys[P];
    __shared__ int4 sbox[P];
    __shared__ float sobj[P];
    __shared__ unsigned long long mask[P * NW];
    __shared__ unsigned long long alive[NW];
    __shared__ int s_n;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x;
    const int ncell = grid * grid;
    const int cs = S / grid;
    const float* h0 = head + (size_t)img * ncell * 6;

    if (tid == 0) s_n = 0;
    // ---- 1. decode
    for (int c = tid; c < P; c += 256) {
        unsigned long long key = 0;
        if (c < ncell) {
            const float* h = h0 + c * 6;
            float o = sigmoid_ref(h[0]);
            float s = o * sigmoid_ref(h[5]);
            if (o > 0.0f && (double)s >= conf_th) {
                int i = c / grid, j = c - i * grid;
                double bx = fmax((double)h[1], 0.0), by = fmax((double)h[2], 0.0);
                double bw = fmax((double)h[3], 0.0), bh = fmax((double)h[4], 0.0);
                double fx = bx * cs, fy = by * cs;
                int ix = fx >= (double)cs ? cs - 1 : (int)fx;
                int iy = fy >= (double)cs ? cs - 1 : (int)fy;
                int px = ix + cs * j, py = iy + cs * i;
                double pw = fmin(bw * S, (double)S), ph = fmin(bh * S, (double)S);
                int hw = (int)(pw / 2), hh = (int)(ph / 2);
                sbox[c] = make_int4(max(px - hw, 0), max(py - hh, 0), min(px + hw, S - 1), min(py + hh, S - 1));
                sobj[c] = o;
                key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)c);
            }
        }
        keys[c] = key;
    }
    __syncthreads();

    // ---- 2. descending sort (invalid keys = 0 sink to the end)
    bitonic_sort<P, true>(keys, tid);
    for (int p = tid; p < P; p += 256)
        if (keys[p] != 0 && (p == P - 1 || keys[p + 1] == 0)) s_n = p + 1;
    __syncthreads();
    const int n = s_n;
    const int nwn = (n + 63) >> 6;

    // ---- 3. suppression bitmasks: bit j of row i  <=>  j > i and IoU(i,j) >= th
    for (int i = wave; i < n; i += 4) {
        const int4 bi = sbox[0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull)];
        for (int c = i >> 6; c < nwn; ++c) {
            int j = (c << 6) + lane;
            bool pred = false;
            if (j > i && j < n) pred = iou_ge(bi, sbox[0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull)], iou_th);
            unsigned long long word = __ballot(pred);
            if (lane == 0) mask[i * NW + c] = word;
        }
    }
    __syncthreads();

    // ---- 4. greedy sweep, alive mask in registers of wave 0 (lane c holds word c)
    if (wave == 0) {
        unsigned long long al = 0;
        if (lane < nwn) {
            for (int b = 0; b < 64; ++b) {
                int p = (lane << 6) + b;
                if (p < n && (keys[p] >> 32) != 0) al |= 1ull << b;  // score == 0 never suppresses / survives
            }
        }
        for (int i = 0; i < n; ++i) {
            unsigned long long aw = __shfl(al, i >> 6);
            if ((aw >> (i & 63)) & 1ull) {
                if (lane < nwn && lane >= (i >> 6)) al &= ~mask[i * NW + lane];
            }
        }
        if (lane < NW) alive[lane] = al;
    }
    __syncthreads();

    // ---- 5. survivors, ascending score (ties: lower cell), first num_cands
    unsigned long long nk[P / 256];
#pragma unroll
    for (int r = 0; r < P / 256; ++r) {
        int p = tid + r * 256;
        unsigned long long k = keys[p];
        bool keep = p < n && ((alive[p >> 6] >> (p & 63)) & 1ull);
        unsigned cellidx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        nk[r] = keep ? ((k & 0xFFFFFFFF00000000ull) | cellidx) : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < P / 256; ++r) keys[tid + r * 256] = nk[r];
    __syncthreads();
    bitonic_sort<P, false>(keys, tid);

    int m = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) m += __popcll(alive[w]);
    const int cnt = min(m, num_cands);
    if (tid == 0) count_out[img] = cnt;
    for (int k = tid; k < num_cands; k += 256) {
        size_t o = (size_t)img * num_cands + k;
        if (k < cnt) {
            unsigned long long key = keys[k];
            unsigned c = (unsigned)(key & 0xFFFFFFFFull);
            float s = __uint_as_float((unsigned)(key >> 32));
            int4 b = sbox[c];
            reinterpret_cast<int4*>(boxes)[o] = b;
            cell_out[o] = (int)c;
            obj_out[o] = sobj[c];
            score_out[o] = fminf(s, 1.0f);
        } else {
            reinterpret_cast<int4*>(boxes)[o] = make_int4(-1, -1, -1, -1);
            cell_out[o] = -1;
            obj_out[o] = 0.0f;
            score_out[o] = 0.0f;
        }
    }
}

}  // namespace

extern "C" int fv_decode_nms(fv_ctx* ctx, const float* head, int nimg, int grid, int image_size, double conf_th,
                             double iou_th, int num_cands, int32_t* boxes, int32_t* cell, float* obj,
                             float* score, int32_t* count) {
    if (!ctx) return FV_ERR_INVALID;
    if (nimg == 0) return FV_OK;
    FV_REQUIRE(ctx, head && boxes && cell && obj && score && count, "fv_decode_nms: NULL buffer");
    FV_REQUIRE(ctx, nimg >= 0 && grid >= 1 && grid <= 22, "fv_decode_nms: grid %d unsupported (1..22)", grid);
    FV_REQUIRE(ctx, image_size >= grid, "fv_decode_nms: image_size %d < grid %d", image_size, grid);
    FV_REQUIRE(ctx, num_cands >= 1 && num_cands <= 512, "fv_decode_nms: num_cands %d unsupported (1..512)", num_cands);
    FV_REQUIRE(ctx, ((uintptr_t)boxes & 15) == 0, "fv_decode_nms: boxes must be 16-byte aligned");
    const int ncell = grid * grid;
    FvProfScope ps(ctx, "decode_nms_kernel", 0.0, (double)nimg * (ncell * 24.0 + num_cands * 28.0 + 4.0));
    if (ncell <= 256 && num_cands <= 256)
        hipLaunchKernelGGL(decode_nms_kernel<256>, dim3(nimg), dim3(256), 0, ctx->stream, head, grid, image_size,
                           conf_th, iou_th, num_cands, boxes, cell, obj, score, count);
    else
        hipLaunchKernelGGL(decode_nms_kernel<512>, dim3(nimg), dim3(256), 0, ctx->stream, head, grid, image_size,
                           conf_th, iou_th, num_cands, boxes, cell, obj, score, count);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}


// ---------------------------------------------------------------------------------------------
// Batched bbox_iou for the accuracy metric (reference evaluate.py:46-75 calls yd.py:183-194 bbox_iou on every
// (ground truth, detection) pair of an image, with float64 corner coordinates read from the csv files): one thread per
// pair, float64 arithmetic in the reference's operation order and branch structure -- every operation is a single IEEE
// double operation, so the result is bit-identical to the Python floats.  A zero union gives nan (0/0) or +-inf, as NumPy does.
namespace {
__device__ __forceinline__ double interval_overlap_f64(double x1, double x2, double x3, double x4) {
    if (x3 < x1) {
        if (x4 < x1) return 0.0;
        return fmin(x2, x4) - x1;
    } else {
        if (x2 < x3) return 0.0;
        return fmin(x2, x4) - x3;
    }
}
__global__ __launch_bounds__(256) void bbox_iou_pairs_kernel(const double* __restrict__ a, const double* __restrict__ b, long long n,
                                                             double* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double ax0 = a[4 * i], ay0 = a[4 * i + 1], ax1 = a[4 * i + 2], ay1 = a[4 * i + 3];
        const double bx0 = b[4 * i], by0 = b[4 * i + 1], bx1 = b[4 * i + 2], by1 = b[4 * i + 3];
        const double iw = interval_overlap_f64(ax0, ax1, bx0, bx1), ih = interval_overlap_f64(ay0, ay1, by0, by1);
        const double inter = iw * ih;
        const double uni = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter;
        out[i] = inter / uni;
    }
}
}  // namespace

extern "C" int fv_bbox_iou_pairs(fv_ctx* ctx, const double* boxes_a, const double* boxes_b, int64_t npairs, double* iou) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, npairs >= 0 && (npairs == 0 || (boxes_a && boxes_b && iou)), "bbox_iou_pairs: NULL buffer");
    if (npairs == 0) return FV_OK;
    long long g = (npairs + 255) / 256;
    hipLaunchKernelGGL(bbox_iou_pairs_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, ctx->stream, boxes_a, boxes_b,
                       (long long)npairs, iou);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}
