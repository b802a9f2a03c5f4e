/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, scalar, one core) of the reference's detect-path
 * post-processing.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object.
 *
 * Follows (paths relative to /root/reference/src/space):
 *   _sigmoid                      yolov3_detect.py:180-181
 *   _interval_overlap / bbox_iou  yolov3_detect.py:165-178, 183-194
 *   do_nms_v2                     yolov3_detect.py:446-458
 *   FaceDetector.detect           face_detection.py:899-949
 *
 * Pinned by tests/golden/{detect_cases,iou_cases}.npz, minted by running the
 * reference's own functions (tests/golden/make_golden.py).
 *
 * Two deliberate, documented definitions where the reference is
 * platform-dependent:
 *  (1) exp: the reference evaluates np.exp on a float32 array (SIMD polynomial,
 *      NumPy-build dependent).  Here e = (float)exp((double)-x), i.e. the
 *      correctly-rounded float32 exponential, then 1.0f/(1.0f+e) in IEEE float32.
 *      Agrees with NumPy to <= 2 ulp; boxes / index sets agree exactly on every
 *      golden case.
 *  (2) ties: np.argsort(kind='quicksort') is not stable; the order among exactly
 *      equal scores is undefined in the reference.  Here ties break toward the
 *      lower row-major cell index.  Golden cases are tie-free.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FVO_MAX_CELLS 4096

static float fvo_sigmoid(float x) {
    float e = (float)exp(-(double)x);
    return 1.0f / (1.0f + e);
}

float fvo_sigmoid_f32(float x) { return fvo_sigmoid(x); }

/* yolov3_detect.py:165-178 -- same branch structure, on exact integers */
static long long interval_overlap(long long x1, long long x2, long long x3, long long x4) {
    if (x3 < x1) {
        if (x4 < x1) return 0;
        return (x2 < x4 ? x2 : x4) - x1;
    } else {
        if (x2 < x3) return 0;
        return (x2 < x4 ? x2 : x4) - x3;
    }
}

/* yolov3_detect.py:183-194; boxes are xmin,ymin,xmax,ymax.  0/0 -> nan, x/0 -> inf
 * (NumPy int64 semantics, which is what FaceDetector.detect feeds it). */
double fvo_bbox_iou(const int* a, const int* b) {
    long long iw = interval_overlap(a[0], a[2], b[0], b[2]);
    long long ih = interval_overlap(a[1], a[3], b[1], b[3]);
    long long inter = iw * ih;
    long long w1 = (long long)a[2] - a[0], h1 = (long long)a[3] - a[1];
    long long w2 = (long long)b[2] - b[0], h2 = (long long)b[3] - b[1];
    long long uni = w1 * h1 + w2 * h2 - inter;
    return (double)inter / (double)uni;
}

void fvo_bbox_iou_batch(const int* a, const int* b, int n, double* out) {
    for (int k = 0; k < n; ++k) out[k] = fvo_bbox_iou(a + 4 * k, b + 4 * k);
}

/* face_detection.py:899-949 for ONE image.
 * head: [grid*grid*6] float32 raw head output (objness logit, bx, by, bw, bh, class logit)
 * outputs (capacity num_cands): boxes[4k..] = xmin,ymin,xmax,ymax; cell = row-major cell
 * index of the candidate; obj, score float32.  Returns number of boxes written, ordered
 * exactly as the reference returns them (ASCENDING score, lowest num_cands kept). */
int fvo_detect_postproc(const float* head, int grid, int image_size, double conf_th,
                        double iou_th, int num_cands, int* out_boxes, int* out_cell,
                        float* out_obj, float* out_score) {
    int ncell = grid * grid;
    if (ncell > FVO_MAX_CELLS) return -1;
    int cs = image_size / grid; /* face_detection.py:325 generalised: CELL_SIZE = grid */
    int S = image_size;
    static __thread int box[FVO_MAX_CELLS][4];
    static __thread int cell[FVO_MAX_CELLS];
    static __thread float obj[FVO_MAX_CELLS], val[FVO_MAX_CELLS];
    static __thread int order[FVO_MAX_CELLS];
    int n = 0;
    for (int i = 0; i < grid; ++i) {
        for (int j = 0; j < grid; ++j) {
            const float* h = head + (size_t)(i * grid + j) * 6;
            float o = fvo_sigmoid(h[0]);            /* fd.py:904 */
            float s = o * fvo_sigmoid(h[5]);        /* fd.py:905, float32 product */
            if (!(o > 0.0f && (double)s >= conf_th)) continue; /* fd.py:909 */
            double bx = (double)h[1] > 0.0 ? (double)h[1] : 0.0; /* fd.py:912-915 */
            double by = (double)h[2] > 0.0 ? (double)h[2] : 0.0;
            double bw = (double)h[3] > 0.0 ? (double)h[3] : 0.0;
            double bh = (double)h[4] > 0.0 ? (double)h[4] : 0.0;
            double fx = bx * cs, fy = by * cs;
            int ix = fx >= (double)cs ? cs - 1 : (int)fx;   /* min(int(bx*cs), cs-1) fd.py:919 */
            int iy = fy >= (double)cs ? cs - 1 : (int)fy;
            int px = ix + cs * j, py = iy + cs * i;
            double pw = bw * S < (double)S ? bw * S : (double)S; /* fd.py:921-922 */
            double ph = bh * S < (double)S ? bh * S : (double)S;
            int hw = (int)(pw / 2), hh = (int)(ph / 2);
            box[n][0] = px - hw > 0 ? px - hw : 0;               /* fd.py:925-928 */
            box[n][1] = py - hh > 0 ? py - hh : 0;
            box[n][2] = px + hw < S - 1 ? px + hw : S - 1;
            box[n][3] = py + hh < S - 1 ? py + hh : S - 1;
            cell[n] = i * grid + j; obj[n] = o; val[n] = s;
            ++n;
        }
    }
    if (n == 0) return 0; /* fd.py:935-936 */

    /* do_nms_v2 (yd.py:446-458): descending score, stable on candidate index */
    for (int k = 0; k < n; ++k) order[k] = k;
    for (int a = 1; a < n; ++a) { /* insertion sort: stable */
        int t = order[a]; int b = a - 1;
        while (b >= 0 && val[order[b]] < val[t]) { order[b + 1] = order[b]; --b; }
        order[b + 1] = t;
    }
    for (int a = 0; a < n; ++a) {
        int ia = order[a];
        if (val[ia] == 0.0f) continue;
        for (int b = a + 1; b < n; ++b) {
            int ib = order[b];
            if (fvo_bbox_iou(box[ia], box[ib]) >= iou_th) val[ib] = 0.0f;
        }
    }
    /* fd.py:942-947: keep score>0 (score=min(classes[0],1.0)), argsort ASCENDING, first num_cands */
    int m = 0;
    for (int k = 0; k < n; ++k) if (val[k] > 0.0f) order[m++] = k;
    for (int a = 1; a < m; ++a) {
        int t = order[a]; int b = a - 1;
        while (b >= 0 && val[order[b]] > val[t]) { order[b + 1] = order[b]; --b; }
        order[b + 1] = t;
    }
    int cnt = m < num_cands ? m : num_cands;
    if (cnt < 0) cnt = 0;
    for (int k = 0; k < cnt; ++k) {
        int s = order[k];
        memcpy(out_boxes + 4 * k, box[s], 4 * sizeof(int));
        out_cell[k] = cell[s]; out_obj[k] = obj[s];
        out_score[k] = val[s] < 1.0f ? val[s] : 1.0f;
    }
    return cnt;
}

/* Batch driver (used for the CPU-baseline timing and for parity over many frames). */
void fvo_detect_postproc_batch(const float* head, int nimg, int grid, int image_size,
                               double conf_th, double iou_th, int num_cands, int* out_boxes,
                               int* out_cell, float* out_obj, float* out_score, int* out_count) {
    size_t hs = (size_t)grid * grid * 6;
    for (int b = 0; b < nimg; ++b) {
        out_count[b] = fvo_detect_postproc(head + b * hs, grid, image_size, conf_th, iou_th, num_cands,
                                           out_boxes + (size_t)b * num_cands * 4,
                                           out_cell + (size_t)b * num_cands,
                                           out_obj + (size_t)b * num_cands,
                                           out_score + (size_t)b * num_cands);
    }
}
