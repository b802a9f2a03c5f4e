// Internal launch interface of elementwise.hip.
#pragma once
#include "common.h"

int fv_ew_bn_finalize(fv_ctx* ctx, const float* psum, const float* psq, int mtiles, int C, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                      float* moving_mean, float* moving_var);
int fv_ew_bn_fold(fv_ctx* ctx, const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                  float* scale, float* shift);
int fv_ew_bn_act(fv_ctx* ctx, const float* z, const float* scale, const float* shift, const float* skip, float* out,
                 long long rows, int C, float leaky);
int fv_ew_bn_bwd_chunks(long long rows, int C);
int fv_ew_bn_bwd(fv_ctx* ctx, const float* g, const float* z, const float* scale, const float* shift, const float* mean,
                 const float* invstd, long long rows, int C, float leaky, float* pdb, float* pdg, float* dbeta, float* dgamma,
                 float* dz, double* slots = nullptr, int nslot = 0, bool reduced = false);
// part != NULL: multi-workgroup form with fv_ew_mse_scratch_floats() floats of scratch (8-byte aligned); NULL: one workgroup
int fv_ew_mse(fv_ctx* ctx, const float* yp, const float* yt, int rows, int C, int Cpad, float* loss, float* dy, float* dbias,
              double* part = nullptr, double grad_weight = 1.0);
int fv_ew_mse_scratch_floats();
int fv_ew_scale(fv_ctx* ctx, float* v, long long n, float alpha);
int fv_ew_adam(fv_ctx* ctx, float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2, float eps);
// training-mode BN without a finalize launch: the conv epilogue adds its column sums to
// [nslot][2][C] fp64 accumulator slots (zeroed by the caller); this pass sums them, normalises, and
// publishes mean/invstd/scale/shift (+ moving statistics) for the backward pass
int fv_ew_bn_stat_slots(int C);
int fv_ew_bn_act_stats(fv_ctx* ctx, const float* z, const double* slots, int nslot, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale, float* shift,
                       float* moving_mean, float* moving_var, const float* skip, float* out, long long rows, int C, float leaky);
int fv_ew_transpose_ntc(fv_ctx* ctx, const float* src, float* dst, int N, int T, int C, int Npad);
// the same for up to 64 layers in one launch; offsets in floats from the two base pointers
int fv_ew_transpose_all(fv_ctx* ctx, const float* src_base, float* dst_base, int nlayers, const long long* src_off,
                        const long long* dst_off, const int* N, const int* T, const int* C, const int* Npad);
int fv_ew_pad_rows(fv_ctx* ctx, const float* src, float* dst, int N, int K, int Kpad);
int fv_ew_slice_cols(fv_ctx* ctx, const float* src, float* dst, long long rows, int C, int Cpad);
int fv_ew_splitk_finish(fv_ctx* ctx, const float* slabs, int ksplit, long long stride, const float* scale, const float* shift,
                        const float* skip, float* out, long long n, int C, float leaky, int do_leaky);
int fv_ew_bn_fold_all(fv_ctx* ctx, const float* params, const float* state, int nlayers, const int* ch_begin, const long long* gamma_off,
                      const long long* beta_off, const long long* mean_off, const long long* var_off, float eps, int total,
                      float* scale, float* shift);
int fv_ew_fd_loss(fv_ctx* ctx, const float* yp, const float* yt, int cells, int Cpad, float* loss, float* dy);
int fv_ew_upsample_concat(fv_ctx* ctx, const float* src, const float* skip, float* out, int B, int Hs, int Ws, int C1, int C2);
// three-scale training (net_yolov3.hip): backward of upsample+concat, bias-gradient column sums, the per-scale detection loss
int fv_ew_upsample_concat_bwd(fv_ctx* ctx, const float* g, float* g_up, float* g_skip, int B, int Hs, int Ws, int C1, int C2);
int fv_ew_colsum_chunks(long long rows);
int fv_ew_colsum(fv_ctx* ctx, const float* dy, long long rows, int C, int Cpad, double* part /*[chunks][C]*/, float* out);
int fv_ew_yolo_loss_blocks(long long nbox);
int fv_ew_yolo_loss_part(fv_ctx* ctx, const float* t, const float* y, long long cells, int ncls, int A, int Cpad, float* dy, double* part,
                         double grad_weight = 1.0);
int fv_ew_yolo_loss_finish(fv_ctx* ctx, const double* part, const long long* cells3, int A, float* loss);
