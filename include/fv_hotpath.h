/*
 * fv_hotpath.h -- C ABI of the MI355X-native FaceDetector hot path.
 *
 * The reference (tonandr/face_vijnana_yolov3) is pure Python on Keras/TensorFlow and exposes
 * no C ABI of its own; the seam this library sits behind is the handful of Keras `Model`
 * methods and NumPy helpers that `FaceDetector` calls.  Every entry point below names the
 * reference interface it replaces (paths relative to /root/reference/src/space; fd.py =
 * face_detection.py, yd.py = yolov3_detect.py).  A ctypes binding is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; fv_last_error(ctx) returns a message
 *    owned by the context (fv_last_error(NULL): message of the last failed fv_create).
 *  - the CALLER owns all device memory (plain pointers + explicit sizes; in this project
 *    torch-ROCm tensors provide the storage).  The library allocates nothing persistent.
 *  - one context per GPU / rank / host thread; all work is enqueued on the context's HIP
 *    stream and is asynchronous with respect to the host unless stated otherwise.
 *  - activations are NHWC float32; conv kernels inside the flat parameter vector are stored
 *    OHWI ([cout][kh][kw][cin]); see fv_layer_desc for offsets.
 */
#ifndef FV_HOTPATH_H
#define FV_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fv_ctx fv_ctx;

#define FV_OK 0
#define FV_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define FV_ERR_HIP (-2)       /* a HIP runtime call failed */
#define FV_ERR_WORKSPACE (-3) /* caller workspace too small */

/* ------------------------------------------------------------------ context */
int fv_abi_version(void);
/* stream: a hipStream_t (may be NULL = default stream).  Replaces the implicit TF session
 * that Keras creates behind FaceDetector.__init__ (fd.py:312-382). */
int fv_create(int device, void* stream, fv_ctx** out);
void fv_destroy(fv_ctx* ctx);
const char* fv_last_error(const fv_ctx* ctx);
int fv_set_stream(fv_ctx* ctx, void* stream);

/* ------------------------------------------------------------------ detect post-processing
 * Replaces the NumPy/Python tail of FaceDetector.detect (fd.py:900-947): float32 sigmoid,
 * threshold, per-cell box decode, do_nms_v2 (yd.py:446-458, IoU yd.py:165-194), score>0
 * filter, ASCENDING argsort and first num_cands.  One workgroup per image.
 *   head   [nimg][grid][grid][6] float32 raw head outputs (device)
 *   boxes  [nimg][num_cands][4] int32 xmin,ymin,xmax,ymax   cell [nimg][num_cands] int32
 *   obj, score [nimg][num_cands] float32                     count [nimg] int32
 * Unused slots are filled with -1 / 0.  grid <= 22 (ncell <= 512), 1 <= num_cands <= 512.
 * Ties between exactly equal scores (undefined in the reference) break toward the lower
 * row-major cell index. */
int fv_decode_nms(fv_ctx* ctx, const float* head, int nimg, int grid, int image_size,
                  double conf_th, double iou_th, int num_cands, int32_t* boxes, int32_t* cell,
                  float* obj, float* score, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* FV_HOTPATH_H */
