// Secondary path: three-scale anchor decode + per-class NMS on the device.
//
// Replaces decode_netout (reference yolov3_detect.py:335-387, including its hard-coded anchor skip
// list yd.py:354-362), correct_yolo_boxes (yd.py:389-404) and do_nms (yd.py:426-444), which the
// reference reaches only from yolov3_detect.py:_main_ (COCO demo).  One image per call, as there.
//  * decode kernel: one 1024-thread workgroup walks the candidate slots in the reference's list
//    order (scale 0,1,2; cell row-major; kept anchors ascending) and compacts the survivors with
//    ballot prefix sums, so the output order equals the reference's `boxes` list
//  * NMS kernel: one workgroup per class; LDS bitonic sort of (prob, index) keys (ties: lower
//    index, the reference's argsort is unstable there), then the greedy sweep with the whole
//    workgroup testing IoU against the current survivor
// float32 arithmetic in the reference's operation order (NumPy 2 scalar rules); exp is the
// correctly rounded float32 exponential (NumPy's SIMD exp may differ by 1 ulp).
#include "common.h"

namespace {

__device__ __forceinline__ float sigf(float x) { return 1.0f / (1.0f + (float)exp(-(double)x)); }
__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }

struct YoloDecodeArgs {
    const float* y[3];
    int grid0, nclass;
    float anchors[18];
    float obj_thresh;
    int net_h, net_w, image_h, image_w;
    float x_off, x_sc, y_off, y_sc;   // correct_yolo_boxes constants (float32-rounded python floats)
    int capacity;
    int* boxes; float* objness; float* classes; int* count;
    long long ystride[3];             // floats between the images of a batch (blockIdx.x = image)
};

__global__ __launch_bounds__(1024) void yolo_decode_kernel(const YoloDecodeArgs a) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    const int C = 5 + a.nclass;
    const int img = blockIdx.x;
    int* const boxes = a.boxes + (size_t)img * a.capacity * 4;
    float* const objness = a.objness + (size_t)img * a.capacity;
    float* const classes = a.classes + (size_t)img * a.capacity * a.nclass;
    for (int s = 0; s < 3; ++s) {
        const int g = a.grid0 << s;
        const int nkeep = s == 1 ? 2 : 1;               // skip list: keep b=1 | b=0,2 | b=1
        const int nslots = g * g * nkeep;
        for (int s0 = 0; s0 < nslots; s0 += 1024) {
            const int slot = s0 + tid;
            bool keep = false;
            int cell = 0, b = 0;
            const float* t = nullptr;
            float conf = 0.f;
            if (slot < nslots) {
                cell = slot / nkeep;
                b = s == 1 ? (slot - cell * nkeep) * 2 : 1;
                t = a.y[s] + (size_t)img * a.ystride[s] + ((size_t)cell * 3 + b) * C;
                conf = sigf(t[4]);
                keep = !(conf < a.obj_thresh);
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int pre = base;
            for (int w = 0; w < wave; ++w) pre += wave_cnt[w];
            const int pos = pre + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && pos < a.capacity) {
                const int row = cell / g, col = cell - row * g;
                float x = ((float)col + sigf(t[0])) / (float)g;
                float y = ((float)row + sigf(t[1])) / (float)g;
                float w = a.anchors[s * 6 + 2 * b] * expf_cr(t[2]) / (float)a.net_w;
                float h = a.anchors[s * 6 + 2 * b + 1] * expf_cr(t[3]) / (float)a.net_h;
                float xmin = x - w / 2.f, ymin = y - h / 2.f, xmax = x + w / 2.f, ymax = y + h / 2.f;
                int4 bx;
                bx.x = (int)((xmin - a.x_off) / a.x_sc * (float)a.image_w);
                bx.z = (int)((xmax - a.x_off) / a.x_sc * (float)a.image_w);
                bx.y = (int)((ymin - a.y_off) / a.y_sc * (float)a.image_h);
                bx.w = (int)((ymax - a.y_off) / a.y_sc * (float)a.image_h);
                reinterpret_cast<int4*>(boxes)[pos] = bx;
                objness[pos] = conf;
                for (int c = 0; c < a.nclass; ++c) classes[(size_t)pos * a.nclass + c] = sigf(t[5 + c]);
            }
            __syncthreads();
            if (tid == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wave_cnt[w]; base += tot; }
            __syncthreads();
        }
    }
    if (tid == 0) a.count[img] = base < a.capacity ? base : a.capacity;
}

__device__ __forceinline__ int ovl(int x1, int x2, int x3, int x4) {
    if (x3 < x1) { if (x4 < x1) return 0; return min(x2, x4) - x1; }
    if (x2 < x3) return 0;
    return min(x2, x4) - x3;
}

template <int P>
__global__ __launch_bounds__(1024) void yolo_nms_kernel(const int* __restrict__ boxes_all, float* __restrict__ classes_all,
                                                        const int* __restrict__ count_all, int nclass, double nms_thresh, int capacity) {
    const int* __restrict__ boxes = boxes_all + (size_t)blockIdx.y * capacity * 4;          // blockIdx.y = image of the batch
    float* __restrict__ classes = classes_all + (size_t)blockIdx.y * capacity * nclass;
    const int* __restrict__ count = count_all + blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);      // [P]
    float* pv = reinterpret_cast<float*>(smem + (size_t)P * 8);                  // [P] current class prob by candidate
    const int c = blockIdx.x, tid = threadIdx.x;
    const int n = min(*count, P);
    for (int i = tid; i < P; i += 1024) {
        float p = i < n ? classes[(size_t)i * nclass + c] : 0.f;
        pv[i] = p;
        // descending prob, ties -> lower index; non-positive probs (incl. padding) sink to the end
        keys[i] = (i < n && p > 0.f) ? (((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += 1024) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                unsigned long long x = keys[i], y = keys[l];
                bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[l] = x; }
            }
            __syncthreads();
        }
    // greedy sweep over the positive-prob prefix
    for (int a = 0; a < n; ++a) {
        const unsigned long long ka = keys[a];
        if (ka == 0ull) break;                                   // uniform: rest has prob <= 0
        const int ia = (int)(0xFFFFFFFFu - (unsigned)(ka & 0xFFFFFFFFull));
        if (pv[ia] != 0.f) {                                     // uniform (LDS value, read after a barrier)
            const int4 A = reinterpret_cast<const int4*>(boxes)[ia];
            for (int b = a + 1 + tid; b < n; b += 1024) {
                const unsigned long long kb = keys[b];
                if (kb == 0ull) break;
                const int ib = (int)(0xFFFFFFFFu - (unsigned)(kb & 0xFFFFFFFFull));
                const int4 Bx = reinterpret_cast<const int4*>(boxes)[ib];
                long long inter = (long long)ovl(A.x, A.z, Bx.x, Bx.z) * ovl(A.y, A.w, Bx.y, Bx.w);
                long long uni = (long long)(A.z - A.x) * (A.w - A.y) + (long long)(Bx.z - Bx.x) * (Bx.w - Bx.y) - inter;
                // python ints: union == 0 raises ZeroDivisionError in the reference; here: no suppression
                if (uni != 0 && (double)inter / (double)uni >= nms_thresh) pv[ib] = 0.f;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) classes[(size_t)i * nclass + c] = pv[i];
}

}  // namespace

extern "C" int fv_yolo_decode_nms_batch(fv_ctx* ctx, const float* y13, const float* y26, const float* y52, int nimg, int grid0, int nclass,
                                        const float* anchors18, float obj_thresh, double nms_thresh, int net_h, int net_w, int image_h,
                                        int image_w, int capacity, int32_t* boxes, float* objness, float* classes, int32_t* count) {
    if (!ctx) return FV_ERR_INVALID;
    FV_REQUIRE(ctx, y13 && y26 && y52 && anchors18 && boxes && objness && classes && count, "yolo_decode_nms: NULL buffer");
    FV_REQUIRE(ctx, nimg >= 1 && nimg <= 65535 && grid0 >= 1 && nclass >= 1 && capacity >= 1 && capacity <= 8192,
               "yolo_decode_nms: capacity must be 1..8192, images 1..65535");
    FV_REQUIRE(ctx, ((uintptr_t)boxes & 15) == 0, "yolo_decode_nms: boxes must be 16-byte aligned");
    YoloDecodeArgs a{};
    a.y[0] = y13; a.y[1] = y26; a.y[2] = y52;
    a.grid0 = grid0; a.nclass = nclass; a.obj_thresh = obj_thresh;
    for (int i = 0; i < 18; ++i) a.anchors[i] = anchors18[i];
    for (int s = 0; s < 3; ++s) a.ystride[s] = (long long)(grid0 << s) * (grid0 << s) * 3 * (5 + nclass);
    a.net_h = net_h; a.net_w = net_w; a.image_h = image_h; a.image_w = image_w;
    // correct_yolo_boxes (yd.py:389-399), including the reference's `new_h = net_w` in the else branch
    double new_w, new_h;
    if ((double)net_w / image_w < (double)net_h / image_h) { new_w = net_w; new_h = ((double)image_h * net_w) / image_w; }
    else { new_h = net_w; new_w = ((double)image_w * net_h) / image_h; }
    a.x_off = (float)((net_w - new_w) / 2. / net_w); a.x_sc = (float)(new_w / net_w);
    a.y_off = (float)((net_h - new_h) / 2. / net_h); a.y_sc = (float)(new_h / net_h);
    a.capacity = capacity; a.boxes = boxes; a.objness = objness; a.classes = classes; a.count = count;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(nimg), dim3(1024), 0, ctx->stream, a);
    FV_LAUNCH_CHECK(ctx);
    int P = 1024;
    while (P < capacity) P <<= 1;
    const size_t lds = (size_t)P * 12;
    const dim3 grid(nclass, nimg);
    if (P == 1024) hipLaunchKernelGGL(yolo_nms_kernel<1024>, grid, dim3(1024), lds, ctx->stream, boxes, classes, count, nclass, nms_thresh, capacity);
    else if (P == 2048) hipLaunchKernelGGL(yolo_nms_kernel<2048>, grid, dim3(1024), lds, ctx->stream, boxes, classes, count, nclass, nms_thresh, capacity);
    else if (P == 4096) hipLaunchKernelGGL(yolo_nms_kernel<4096>, grid, dim3(1024), lds, ctx->stream, boxes, classes, count, nclass, nms_thresh, capacity);
    else {
        FV_HIP(ctx, hipFuncSetAttribute((const void*)yolo_nms_kernel<8192>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(yolo_nms_kernel<8192>, grid, dim3(1024), lds, ctx->stream, boxes, classes, count, nclass, nms_thresh, capacity);
    }
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

extern "C" int fv_yolo_decode_nms(fv_ctx* ctx, const float* y13, const float* y26, const float* y52, int grid0, int nclass,
                                  const float* anchors18, float obj_thresh, double nms_thresh, int net_h, int net_w, int image_h,
                                  int image_w, int capacity, int32_t* boxes, float* objness, float* classes, int32_t* count) {
    return fv_yolo_decode_nms_batch(ctx, y13, y26, y52, 1, grid0, nclass, anchors18, obj_thresh, nms_thresh, net_h, net_w, image_h, image_w,
                                    capacity, boxes, objness, classes, count);
}
