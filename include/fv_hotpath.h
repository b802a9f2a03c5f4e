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
/* (The context owns a low-priority side stream for the backward overlap.  The HIP runtime multiplexes a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues, default 4: a context created AFTER an RCCL communicator has existed in the process can find
 * its side stream sharing a hardware queue with the compute stream, which serialises the overlap -- 30 - 40 % on a training step.
 * Create contexts before communicators, or run the process with GPU_MAX_HW_QUEUES=8; the Python host package sets that default.
 * DESIGN 6.) */
void fv_destroy(fv_ctx* ctx);
const char* fv_last_error(const fv_ctx* ctx);
int fv_set_stream(fv_ctx* ctx, void* stream);
/* Where fv_bucket_fn fires.  Default (0): after the context's stream has been made to wait for the range's weight-gradient --
 * work the callback enqueues on the context's stream sees the finished range.  on = 1 (with the overlap on): as soon as the
 * range's weight-gradient kernels are in the SIDE stream's queue -- the callback must enqueue its work (the all-reduce of the
 * bucket) on fv_side_stream(ctx), where it is ordered behind them and runs beside the data-gradient chain without an event
 * between the context's stream and the communication; the side stream is joined before fv_train_step returns. */
int fv_set_bucket_on_side(fv_ctx* ctx, int on);
void* fv_side_stream(fv_ctx* ctx);   /* the hipStream_t of the internal side stream */
/* Update rule of the BatchNormalization moving mean / variance in every training-mode BN launch that follows (reference
 * yd.py:212 `BatchNormalization(epsilon=0.001)`; the update itself is third-party: Keras 2.2.4 `K.moving_average_update` ->
 * TF 1.x `assign_moving_average(..., zero_debias=True)`).  step = 0 (default): plain EMA, moving <- m moving + (1 - m) batch.
 * step = t >= 1: the t-th zero-debiased update since the model was built, moving_t = b_t / (1 - m^t) with the zero-initialised
 * b_t = m b_{t-1} + (1 - m) batch_t -- the first update REPLACES the stored value (so loaded Darknet statistics are forgotten
 * at the first training step, as in the reference's stack).  The setting is consumed by the NEXT fv_train_step /
 * fv_yolov3_train_step (which resets it to 0 when it returns) or by the per-operator fv_bn_finalize / fv_bn_act_slots calls that
 * follow; the caller sets t before every training step.  Parity unpinned (neither Keras nor TF is importable here). */
int fv_set_bn_zero_debias_step(fv_ctx* ctx, long long step);
/* ------------------------------------------------------------------ tuning
 * Schedule / kernel-selection switches with no counterpart in the reference.  None of them changes WHAT is computed: every
 * value gives results that are bit-identical or differ only in fp32 summation order (stated per key); all default to 1 (the
 * fastest measured configuration) and exist for A/B measurements and for the tests that compare the specialised
 * kernels with the generic ones.  key (value != 0 = on):
 *   "overlap"           fv_train_step runs the weight-gradient kernels on an internal side stream, concurrently with the
 *                       data-gradient / BN-backward chain on the context's stream (independent given dz); all side-stream work is
 *                       joined back before a gradient range is reported (fv_bucket_fn) and before the call returns.  0 serialises
 *                       everything on the context's stream.
 *   "tail_split"        when the 128x128 output tiles of a conv launch do not fill whole rounds of the 512 resident workgroup
 *                       slots, the tiles of the last partial round are cut into K slices whose partial tiles a fix-up kernel sums
 *                       in fixed slice order.  Deterministic; changes the summation order of those tiles.
 *   "conv_waves8"       128- / 64-wide conv tiles as 512-thread workgroups (8 waves of 64x32 / 32x32: four waves per SIMD) instead
 *                       of 256 threads (4 waves of 64x64).  Bit-identical.
 *   "conv1x1_persist"   1x1 stride-1 launches with more than 512 tiles (forward and data-gradient of the 1x1 layers at batch >= ~16)
 *                       run as a persistent GEMM whose workgroups walk several tiles with the next tile's operands in flight
 *                       during the epilogue (conv1x1_mfma.hip) instead of one workgroup per tile.  Bit-identical.
 *   "conv_small"        small-M inference (batch 1): a layer whose launch fits one workgroup per CU (<= 256 tiles of 64 x 64, 64 x 32
 *                       or 32 x 32) runs with the K dimension split INSIDE each 512-thread workgroup (four wave pairs or eight waves
 *                       multiply 1/4 or 1/8 of the K steps each, the sums are formed in group order) instead of K slices of one
 *                       workgroup each + a finish launch.  Deterministic; another fp32 summation order.
 *   "conv_bm64"         small-M inference (fewer than 192 tiles of 128 x 128: batch 1): 64-row tiles where they leave fewer padded rows
 *                       than 128-row ones (13x13, 26x26, 52x52 pixels) -- less padding to multiply, fewer K slices to sum.  Changes the
 *                       K-split plan of those launches, i.e. their fp32 summation order.
 *   "conv_halo"         the 3x3 layers with 32 -> 64 channels (conv_1, conv_3): training forward and stride-2 data-gradient from
 *                       an LDS halo tile with resident weights (conv9_mfma.hip, dgrad9s2_mfma.hip).  z / dx bit-identical, the
 *                       statistics are the same sums in another (fp64) order.
 *   "conv0_direct"      first layer (3 -> 32 channels) as a direct vector-FMA kernel when W % 32 == 0 and H % 8 == 0 instead of
 *                       the matrix-core gather kernel.  Bit-identical.
 *   "wgrad_fused_taps"  weight-gradients of conv_0 / conv_1 / conv_2 / conv_3 from halo tiles / streaming units (wgrad0, wgrad1,
 *                       wgrad9) instead of the generic kernel.  Same products, other float-atomic summation order.
 * Unknown keys return FV_ERR_INVALID.  The environment variable FV_OPTIONS="key=0,key=1" sets initial values at fv_create. */
int fv_set_option(fv_ctx* ctx, const char* key, long long value);
int fv_get_option(fv_ctx* ctx, const char* key, long long* value);
/* The per-operator conv entry points (fv_conv2d_forward / fv_conv2d_dgrad) have no workspace
 * argument; a caller that wants the tail split there lends device scratch here (NULL, 0 = none;
 * 64 MiB covers every Darknet-53 shape at batch 40).  The buffer must stay valid until the calls that
 * use it have completed on the stream.  fv_train_step / fv_forward_infer use their own workspace. */
int fv_set_conv_scratch(fv_ctx* ctx, void* buf, size_t bytes);

/* ------------------------------------------------------------------ per-kernel timing
 * Measurement aid with no counterpart in the reference (it has no profiler hooks):
 * when enabled, every kernel launch of this library is bracketed by a HIP event pair on the
 * context's stream; fv_profile_collect synchronises and returns one aggregate per kernel with
 * the ALGORITHMIC flops / bytes of the launches (what bench.py's `roofline` is computed from).
 * on = 2: the records of the matrix kernels additionally carry the launch's problem shape in the name ("conv_kernel<128,2,4,false>
 * M108160 N256 K1152 r": rows, output channels, taps x input channels; s2 = stride-2 data-gradient, ks = K split, r = fused
 * BN-backward reduction), i.e. one aggregate per kernel AND shape (names are cut at 63 characters). */
typedef struct fv_profile_rec {
    char name[64];
    int64_t launches;
    double ms_total, flops_total, bytes_total;
} fv_profile_rec;
int fv_profile_enable(fv_ctx* ctx, int on);
int fv_profile_collect(fv_ctx* ctx, fv_profile_rec* out, int max_recs, int* n_out);

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


/* Batched bbox_iou (yd.py:183-194) for the accuracy metric cal_mAP_fd (evaluate.py:46-75):
 * boxes_a, boxes_b [npairs][4] float64 xmin,ymin,xmax,ymax (device) -> iou [npairs] float64, bit-identical to
 * the reference's Python-float arithmetic (nan / inf for a zero union, as NumPy division gives). */
int fv_bbox_iou_pairs(fv_ctx* ctx, const double* boxes_a, const double* boxes_b, int64_t npairs, double* iou);

/* ------------------------------------------------------------------ network description
 * The FaceDetector network: the first 52 conv+BN+LeakyReLU(0.1) layers / 23 residual adds of
 * make_yolov3_model (yd.py:221-267) as re-wired by FaceDetector.YOLOV3Base (fd.py:404-593), plus
 * the Conv2D(6, 3x3, 'same', linear, bias) head (fd.py:348-352).  Parameters live in ONE flat
 * float32 vector owned by the caller (so Adam and the gradient all-reduce are single ranges):
 *   per base layer: kernel OHWI [cout][k][k][cin] at w_off, gamma[cout] at gamma_off, beta[cout]
 *   at beta_off;  head: kernel at w_off, bias[6] at beta_off (gamma_off = -1).
 * BatchNorm moving statistics live in a second flat vector: mean at mean_off, var at var_off. */
typedef struct fv_layer_desc {
    int32_t darknet_index; /* conv_<i> / bnorm_<i> of yd.py; -1 for the head ('output') */
    int32_t ksize, stride, cin, cout;
    int32_t has_bn;        /* 1: BN(eps 1e-3)+LeakyReLU(0.1) follow; 0: linear + bias (head) */
    int32_t role;          /* 0 plain, 1 first conv of a residual block (its input is the skip),
                              2 second conv of a residual block (add(skip, x) follows), 3 head */
    int32_t in_div, out_div; /* spatial size = image_size / div */
    int64_t w_off, gamma_off, beta_off; /* offsets (floats) into the parameter vector */
    int64_t mean_off, var_off;          /* offsets into the BN-state vector (-1 for the head) */
} fv_layer_desc;

int fv_num_layers(void);                       /* 53 */
int fv_layer(int i, fv_layer_desc* out);
int64_t fv_param_count(void);                  /* 40 640 230 trainable floats */
int64_t fv_state_count(void);                  /* 35 712 BN moving mean/var floats */
/* bytes of caller-provided device workspace for a batch; training != 0 keeps every layer's
 * pre-BN and activated output for the backward pass. */
size_t fv_workspace_bytes(int batch, int image_size, int training);

/* ------------------------------------------------------------------ hot path: network level */
/* Replaces self.model.predict(image) (fd.py:899): BN in inference mode (moving statistics folded
 * into the conv epilogue), x [batch][S][S][3] float32 NHWC in [0,1], y [batch][S/32][S/32][6].
 * Every tensor is addressed through one 2 GiB buffer descriptor: batch * S * S * 32 (the first layer's output) must not exceed
 * 2^29 floats -- 96 images at 416, 45 at 608; beyond that the call returns FV_ERR_INVALID ("... exceeds 2^29 elements (2 GiB buffer
 * descriptor) ...") at the first convolution, before any of them is launched, and the caller splits the batch (inference is per image; the Python host does). The
 * same bound holds for fv_forward_base, fv_train_step and the fv_yolov3_* entry points. */
int fv_forward_infer(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch,
                     int image_size, void* workspace, size_t workspace_bytes, float* y);
/* The model `FaceDetector.YOLOV3Base` returns (fd.py:384-600): the Darknet-53 base alone, input -> the output of the last
 * residual add (add_23), feat [batch][S/32][S/32][1024] float32 -- the tensor the head conv reads (fd.py:344-352) and the
 * backbone output FaceIdentifier builds on (face_identification.py:323, 397-614).  Same kernels, same arithmetic and the same
 * workspace as fv_forward_infer; y (may be NULL) additionally receives the head output of the same pass. */
int fv_forward_base(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch,
                    int image_size, void* workspace, size_t workspace_bytes, float* feat, float* y);

/* Called (on the host, in enqueue order) when the gradient range [offset, offset+count) of the
 * flat gradient vector has been fully enqueued on the context's stream -- the hook a data-parallel
 * host uses to start the RCCL all-reduce of that bucket while the backward pass continues. */
typedef void (*fv_bucket_fn)(void* user, int64_t offset, int64_t count);

/* One optimisation step's forward + loss + backward; replaces the TF graph that
 * model.fit_generator runs per batch (fd.py:621-627) with loss='mse' (fd.py:381): training-mode BN
 * (batch statistics, moving statistics updated in bn_state), mean-squared error over every element
 * of [batch][G][G][6], gradients of all 40 640 230 parameters written to `grads` (overwritten).
 * loss: one float (device).  Follow with fv_adam_step.
 * loss_weight: 1 for a single-GPU step.  Data parallel (keras.utils.multi_gpu_model, fd.py:358-371: ONE loss over the
 * concatenated tower outputs): the share n_rank / n_total of the merged batch this call's slice holds -- it scales dL/dy in the
 * loss kernel, so every gradient of the step arrives pre-scaled and the SUM all-reduce over the ranks yields the gradient of
 * the merged-batch mean without a separate scaling pass over the 162 MB vector; `loss` stays this slice's own mean.
 * Reproducibility: kernel gradients are accumulated with float atomics and the BN statistics with fp64
 * atomics, so two runs agree to rounding (dW ~3e-6 relative, statistics in the last float bit at most),
 * not bit for bit; fv_forward_infer is bit-reproducible. */
int fv_train_step(fv_ctx* ctx, const float* params, float* bn_state, const float* x, const float* y_true,
                  int batch, int image_size, void* workspace, size_t workspace_bytes, float* grads,
                  float* loss, double loss_weight, fv_bucket_fn on_bucket, void* user);

/* Introspection of the training workspace after fv_train_step (test aid; Keras keeps these tensors
 * inside the TF graph): where layer `layer` (0..51) keeps which = 0 pre-BN output z, 1 activated output a
 * ([batch][S/div][S/div][cout] each), 2 batch mean, 3 1/sqrt(var+eps), 4 scale, 5 shift ([cout] each).
 * offset_bytes is relative to the workspace pointer given to fv_train_step with the same batch /
 * image_size.  The parity tests read z / scale / shift back to learn which LeakyReLU slope the GPU took
 * for every element, so that the float64 oracle can be evaluated on the same side of each kink. */
int fv_train_workspace_tensor(int batch, int image_size, int layer, int which, size_t* offset_bytes,
                              int64_t* count);

/* keras.optimizers.Adam(lr, beta_1, beta_2, decay) update (fd.py:376-379), Keras 2.2.4 formula:
 * t = iteration+1; lr_t = lr/(1+decay*iteration) * sqrt(1-b2^t)/(1-b1^t);
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps)  (eps = 1e-7). */
int fv_adam_step(fv_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n,
                 int64_t iteration, double lr, double beta_1, double beta_2, double eps, double decay);
/* v[i] *= alpha over n floats on the context's stream.  The data-parallel host uses it for the one small vector whose mean over
 * the ranks is not a gradient: the 35 712 BN moving statistics after their SUM all-reduce (the reference's towers race on
 * shared variables, fd.py:369 -- SURVEY 8e; the build keeps the ranks identical by averaging). */
int fv_scale(fv_ctx* ctx, float* v, int64_t n, double alpha);

/* ------------------------------------------------------------------ hot path: single operators
 * (what the network-level calls are built from; exported for unit parity tests) */
/* ZeroPadding2D(1)+Conv2D(k=3,'valid',strides=stride) or Conv2D(k=1) of yd.py:205-211.
 * x [B][H][W][cin], w OHWI [cout][k][k][cin] (cin % 32 == 0; for the cin=3 first layer pass the
 * [cout][32] form made by fv_pack_first_layer), out [B][H/stride][W/stride][cout].
 * out = acc*scale[c]+shift[c] (either may be NULL), then LeakyReLU(leaky) if leaky >= 0, then
 * + addend (may be NULL).  If psum != NULL the raw result is stored instead and psum/psq receive
 * per-tile column sums / sums of squares, [fv_conv2d_stat_rows(M)][cout] each. */
int fv_conv2d_forward(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout,
                      int ksize, int stride, const float* scale, const float* shift, float leaky,
                      const float* addend, float* out, float* psum, float* psq);
int fv_conv2d_stat_rows(int64_t out_pixels);
/* gradient w.r.t. the conv input.  dy [B][H/stride][W/stride][cout_pad], w_t [cin][k*k][cout_pad]
 * (fv_transpose_weights), dx [B][H][W][cin] = dgrad (+ addend if not NULL). */
int fv_conv2d_dgrad(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin,
                    int cout_pad, int ksize, int stride, const float* addend, float* dx);
/* gradient w.r.t. the kernel, ACCUMULATED into dw OHWI [cout][k][k][cin] (zero it first).
 * dy has dy_stride >= cout channels per pixel. */
int fv_conv2d_wgrad(fv_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int cin, int cout,
                    int dy_stride, int ksize, int stride, float* dw);
int fv_transpose_weights(fv_ctx* ctx, const float* w, int cout, int taps, int cin, int cout_pad, float* w_t);
int fv_pack_first_layer(fv_ctx* ctx, const float* w, int cout, int k_elems, float* w_packed /*[cout][32]*/);
/* training-mode BatchNormalization(eps) statistics from the conv partials: mean, 1/sqrt(var+eps),
 * scale = gamma*invstd, shift = beta-mean*scale; moving stats updated in place when not NULL
 * (Keras: moving = momentum*moving + (1-momentum)*batch, variance scaled by n/(n-(1+eps))). */
int fv_bn_finalize(fv_ctx* ctx, const float* psum, const float* psq, int stat_rows, int C, int64_t count,
                   const float* gamma, const float* beta, float eps, float momentum, float* mean,
                   float* invstd, float* scale, float* shift, float* moving_mean, float* moving_var);
/* out = LeakyReLU(z*scale+shift) (+ skip) over [rows][C] */
int fv_bn_act(fv_ctx* ctx, const float* z, const float* scale, const float* shift, const float* skip,
              float* out, int64_t rows, int C, float leaky);
/* backward of BN(train)+LeakyReLU: dz, dgamma[C], dbeta[C] from g = dL/d(activated output).
 * scratch: 2 * fv_bn_bwd_scratch_floats(rows, C) floats. */
int64_t fv_bn_bwd_scratch_floats(int64_t rows, int C);
int fv_bn_bwd(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift,
              const float* mean, const float* invstd, int64_t rows, int C, float leaky, float* scratch,
              float* dbeta, float* dgamma, float* dz);
/* ---- the fused forms fv_train_step actually runs (same Keras semantics, yd.py:212-215; exported so
 * that each is checked against float64 on its own).  "Slots": [nslot][2][C] float64 accumulators the
 * CALLER zeroes; producers ADD per-tile column sums with fp64 atomics (slot = tile % nslot), the
 * consumer sums the slots in fixed order.  nslot = fv_bn_stat_slots(C); C % 4 == 0, C <= 1024 and C
 * divides or is divided by 256. */
int fv_bn_stat_slots(int C);
/* conv forward storing the raw result z and ADDING the column sums of z and z^2 to `slots`. */
int fv_conv2d_forward_slots(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout,
                            int ksize, int stride, float* z, double* slots, int nslot);
/* training-mode BN + LeakyReLU (+ skip) fed by the slots: sums them, publishes mean / invstd / scale /
 * shift, updates the moving statistics (may be NULL), writes out = leaky(z*scale+shift) (+ skip). */
int fv_bn_act_slots(fv_ctx* ctx, const float* z, const double* slots, int nslot, int64_t rows, int C,
                    const float* gamma, const float* beta, float eps, float momentum, float* mean, float* invstd,
                    float* scale, float* shift, float* moving_mean, float* moving_var, const float* skip,
                    float* out, float leaky);
/* fv_conv2d_dgrad whose epilogue also reduces d-beta / d-gamma of the BN+LeakyReLU layer that PRODUCED
 * the conv input: with that layer's pre-BN tensor bn_z [B][H][W][cin] and its mean / invstd / scale /
 * shift, gy = dx * leaky'(bn_z*scale+shift) and the column sums of gy and gy*(bn_z-mean)*invstd are
 * ADDED to `slots`.  cin % 4 == 0.  With scratch lent (fv_set_conv_scratch) the launch may take the
 * tail split, whose fix-up kernel then does the same reduction. */
int fv_conv2d_dgrad_bnred(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin,
                          int cout_pad, int ksize, int stride, const float* addend, float* dx,
                          const float* bn_z, const float* scale, const float* shift, const float* mean,
                          const float* invstd, float leaky, double* slots, int nslot);
/* backward of BN(train)+LeakyReLU through the slots.  reduced = 0: this call first adds the column
 * sums to `slots` (its own reduction pass); reduced = 1: they are already there (fv_conv2d_dgrad_bnred).
 * Then: d-beta, d-gamma = slot sums (as float), dz = scale*(gy - dbeta/rows - xhat*dgamma/rows). */
int fv_bn_bwd_slots(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift,
                    const float* mean, const float* invstd, int64_t rows, int C, float leaky, double* slots,
                    int nslot, int reduced, float* dbeta, float* dgamma, float* dz);
/* loss = mean((yp-yt)^2) over [rows][C]; dy [rows][c_pad] = 2(yp-yt)/(rows*C) zero padded;
 * dbias[C] = column sums of dy (may be NULL). */
int fv_mse_loss_grad(fv_ctx* ctx, const float* yp, const float* yt, int rows, int C, int c_pad,
                     float* loss, float* dy, float* dbias);
/* Letterbox preprocessing (SURVEY 8f "next" row 1): replaces image/255 -> cv2.resize(INTER_CUBIC)
 * -> cv2.copyMakeBorder(zeros) of fd.py:112-147 / 656-694 / 798-835.  src: uint8 [h][w][3] (device),
 * dst: float32 [S][S][3]; geom (host, may be NULL) receives w_p, h_p, pad_t, pad_b, pad_l, pad_r.
 * Geometry is exact; pixels follow OpenCV's bicubic (a=-0.75) in fp32 (parity unpinned vs cv2). */
int fv_letterbox(fv_ctx* ctx, const uint8_t* src, int h, int w, int image_size, float* dst, int32_t* geom);
/* The same for a whole training batch in one launch (fd.py:98-147, the body of
 * TrainingSequence.__getitem__): the n decoded images lie back to back in `packed` (device; one
 * host-to-device copy per batch), image i at byte offsets[i] with hw[2i] rows and hw[2i+1] columns
 * (offsets, hw, geom: HOST arrays; geom [n][6] may be NULL).  dst [n][S][S][3].  Pixels identical to
 * n calls of fv_letterbox. */
int fv_letterbox_batch(fv_ctx* ctx, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                       int image_size, float* dst, int32_t* geom);

/* ------------------------------------------------------------------ JPEG decode, split host / device
 * (replaces `imread` of fd.py:112, 656, 798 for baseline / extended-sequential Huffman JPEGs with 1 or 3
 * components and 4:4:4 / 4:2:2 / 4:2:0 sampling; anything else -> FV_ERR_INVALID and the caller decodes that file elsewhere).
 * The host part (no context, thread-safe, pure CPU) parses the headers and Huffman-decodes the scan into quantised coefficients:
 * int16, natural (de-zigzagged) order, [64] per block, component after component, each component's block grid padded to whole
 * MCUs in raster order.  The device part dequantises, inverse-transforms (libjpeg's jpeg_idct_islow), upsamples the chroma
 * (libjpeg's "fancy" triangle filters) and converts YCbCr -> RGB for a whole batch in two launches: pixels bit-identical to
 * libjpeg-turbo's default decode (what Pillow / scikit-image return). */
typedef struct fv_jpeg_info {
    int32_t width, height, ncomp, hmax, vmax, restart_interval;
    int32_t h[3], v[3], blocks_w[3], blocks_h[3];
    int64_t coef_off[3];   /* first coefficient of component c, in int16 units from the image's coefficient array */
    int64_t total_coefs;
    uint16_t qt[3][64];    /* quantisation table of component c, natural order */
} fv_jpeg_info;
typedef struct fv_jpeg_desc {   /* one image of a batch; array in DEVICE memory */
    int32_t width, height, ncomp, hmax, vmax, reserved;
    int32_t blocks_w[3], blocks_h[3];
    int64_t coef_off[3];   /* absolute, int16 units from `coefs` */
    int64_t plane_off[3];  /* absolute, bytes from `planes` (component c needs blocks_w*blocks_h*64 bytes) */
    int64_t rgb_off;       /* bytes from `rgb` (height*width*3 bytes, rows packed) */
    uint16_t qt[3][64];
} fv_jpeg_desc;
int fv_jpeg_parse(const uint8_t* data, size_t nbytes, fv_jpeg_info* info);
int fv_jpeg_entropy_decode(const uint8_t* data, size_t nbytes, int16_t* coefs, int64_t ncoefs);   /* HOST buffers */
int64_t fv_jpeg_plane_bytes(const fv_jpeg_info* info);
/* coefs, descs, planes (scratch), rgb: device; max_blocks / max_pixels: the largest block / pixel count of one image (grid size) */
int fv_jpeg_reconstruct_batch(fv_ctx* ctx, const int16_t* coefs, const fv_jpeg_desc* descs, int n, uint8_t* planes, uint8_t* rgb,
                              int64_t max_blocks, int64_t max_pixels);

/* ------------------------------------------------------------------ secondary: three-scale YOLOv3
 * (SURVEY 8a-17/18).  The reference builds this graph in make_yolov3_model (yd.py:217-311) and
 * runs it only from yolov3_detect.py:_main_ (COCO demo: yd.py:596-598 decode, do_nms); FaceDetector
 * discards it.  Inference only.  Flat layout: the 52 base layers exactly as fv_layer(0..51), then
 * the 23 layers 75..105 (fv_yolov3_layer: role 4 = conv+BN+leaky, role 5 = detection conv with bias
 * at beta_off, linear).  out_channels = 3*(5+classes) (255 for COCO). */
int fv_yolov3_num_layers(void);                                   /* 75 */
int fv_yolov3_layer(int i, int out_channels, fv_layer_desc* out);
int64_t fv_yolov3_param_count(int out_channels);                  /* 61 949 149 at 255 */
int64_t fv_yolov3_state_count(int out_channels);                  /* 52 608 */
size_t fv_yolov3_workspace_bytes(int batch, int image_size, int out_channels);
/* replaces yolov3.predict (yd.py:590): y13/y26/y52 = [batch][S/32|S/16|S/8]^2[out_channels] */
int fv_yolov3_forward(fv_ctx* ctx, const float* params, const float* bn_state, const float* x, int batch,
                      int image_size, int out_channels, void* workspace, size_t workspace_bytes,
                      float* y13, float* y26, float* y52);
/* TRAINING the three-scale graph (SURVEY 8f row 4 -- "three-scale YOLO head + bbox/objectness loss" of the
 * north star; the reference builds this graph for inference only and defines no loss for it, so this is the
 * build's own definition, restated in oracle/net_oracle.py).  One step's forward (training-mode BN in all 72
 * BN layers, moving statistics updated) + loss + backward through the heads, both UpSampling2D+concatenate
 * routes and the base; gradients of every parameter of the fv_yolov3_layer layout into `grads` (overwritten).
 * yt13 / yt26 / yt52: targets shaped like the outputs, [batch][g][g][3][5+classes].  Loss = sum over the
 * three scales of the mean over (cell, anchor) of
 *     ( bce(t4, y4) + mean_{k<4} |t_k - y_k| + mean_c bce(t_{5+c}, y_{5+c}) ) / 3
 * -- the reference's fd_loss (fd.py:59-64) generalised to 3 anchors and `classes` classes, with the
 * cross-entropies on logits: bce(t, y) = max(t,0) - t*y + log1p(exp(-|t|)).  Follow with fv_adam_step.
 * loss_weight: as in fv_train_step (scales the gradient, not the reported loss).
 * on_bucket (may be NULL): as in fv_train_step -- called on the host as the gradient range of a layer completes, in reverse
 * execution order = descending offsets, contiguous, covering every parameter once (data-parallel overlap of the all-reduce). */
size_t fv_yolov3_train_workspace_bytes(int batch, int image_size, int out_channels);
int fv_yolov3_train_step(fv_ctx* ctx, const float* params, float* bn_state, const float* x, const float* yt13,
                         const float* yt26, const float* yt52, int batch, int image_size, int out_channels,
                         void* workspace, size_t workspace_bytes, float* grads, float* loss, double loss_weight,
                         fv_bucket_fn on_bucket, void* user);
/* as fv_train_workspace_tensor, for the workspace of fv_yolov3_train_step (BN layers of fv_yolov3_layer) */
int fv_yolov3_train_workspace_tensor(int batch, int image_size, int out_channels, int layer, int which,
                                     size_t* offset_bytes, int64_t* count);
/* replaces decode_netout x3 (yd.py:335-387, with its anchor skip list), correct_yolo_boxes
 * (yd.py:389-404) and do_nms (yd.py:426-444) for ONE image: outputs in the reference's list order;
 * boxes [capacity][4] int32 xmin,ymin,xmax,ymax in image pixels, objness [capacity],
 * classes [capacity][nclass] (suppressed entries zeroed), count (device int).  anchors18: host
 * floats, scale 0 first.  capacity <= 8192.  A zero-area pair (ZeroDivisionError in the reference)
 * does not suppress. */
int fv_yolo_decode_nms(fv_ctx* ctx, const float* y13, const float* y26, const float* y52, int grid0, int nclass,
                       const float* anchors18, float obj_thresh, double nms_thresh, int net_h, int net_w,
                       int image_h, int image_w, int capacity, int32_t* boxes, float* objness, float* classes,
                       int32_t* count);

/* The same for a BATCH of images of one size (the driver loop of yd.py:596-604 runs the chain image by image): outputs
 * [nimg][g][g][3*(5+nclass)] per scale, boxes [nimg][capacity][4], objness [nimg][capacity], classes [nimg][capacity][nclass],
 * count [nimg] -- one launch pair for the whole batch (workgroup = image in the decode, (class, image) in the NMS). */
int fv_yolo_decode_nms_batch(fv_ctx* ctx, const float* y13, const float* y26, const float* y52, int nimg, int grid0, int nclass,
                             const float* anchors18, float obj_thresh, double nms_thresh, int net_h, int net_w,
                             int image_h, int image_w, int capacity, int32_t* boxes, float* objness, float* classes,
                             int32_t* count);

/* fd_loss (fd.py:59-64) -- DEFINED BUT NEVER USED by the reference (every compile() passes
 * loss='mse', fd.py:335/366/370/381); provided as an operator only, not wired into fv_train_step.
 * yp, yt [cells][6]; per cell (BCE(y0,p0) + mean_{c=1..4} sqrt((y_c-p_c)^2) + BCE(y5,p5))/3 with
 * Keras' probability-space binary_crossentropy (p clipped to [1e-7, 1-1e-7]); loss = mean over
 * cells; dy [cells][c_pad] its gradient (0 outside the clip range and at y_c == p_c). */
int fv_fd_loss_grad(fv_ctx* ctx, const float* yp, const float* yt, int cells, int c_pad, float* loss,
                    float* dy);

#ifdef __cplusplus
}
#endif
#endif /* FV_HOTPATH_H */
