// Internal operator-level launchers (ops.hip), used by net.hip.
#pragma once
#include "common.h"

int fv_op_conv_forward(fv_ctx* ctx, const float* x, const float* w, int B, int H, int W, int cin, int cout, int ksize,
                       int stride, int epi, const float* scale, const float* shift, float leaky, const float* addend,
                       float* out, float* psum, float* psq, int ksplit = 1,
                       double* stat_slots = nullptr, int stat_nslot = 0);
// optional fused BN-backward reduction of the layer whose output gradient a data-gradient produces (conv.h FV_EPI_BNRED)
struct FvBnRed { const float *z, *scale, *shift, *mean, *invstd; double* slots; int nslot; float leaky; };
int fv_op_conv_dgrad(fv_ctx* ctx, const float* dy, const float* w_t, int B, int H, int W, int cin, int cout_pad, int ksize,
                     int stride, const float* addend, float* dx, const FvBnRed* bn = nullptr);
int fv_op_conv_wgrad(fv_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int cin, int cout, int dy_stride,
                     int ksize, int stride, float* dw);
