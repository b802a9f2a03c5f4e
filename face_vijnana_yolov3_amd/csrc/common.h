// Shared internals of libfv_hotpath (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/fv_hotpath.h"

struct FvProfRec {
    const char* name;
    double flops, bytes;
    hipEvent_t e0, e1;
};

struct fv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // optional per-launch timing (fv_profile_enable): HIP event pairs on the launch stream
    bool prof_on = false;
    bool prof_shapes = false;          // fv_profile_enable(ctx, 2): the matrix kernels' records carry their problem shape in the name
    std::vector<std::string*> prof_names;   // interned shape-tagged names (stable addresses for FvProfRec::name)
    std::vector<FvProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    // backward-pass overlap: weight-gradient kernels run on a side stream next to the
    // data-gradient / BN-backward chain (option "overlap")
    bool overlap = true;
    hipStream_t side = nullptr;
    bool bucket_on_side = false;   // fv_set_bucket_on_side: fv_bucket_fn fires when the range's weight-gradient is in the side stream's queue
    hipEvent_t ev_dz[2] = {nullptr, nullptr}, ev_wg[2] = {nullptr, nullptr};
    // scratch for the conv tail split (conv.h); lent by the network-level entry points out of the
    // caller's workspace for the duration of one call, NULL otherwise
    float* tail_slab = nullptr;
    long long tail_slab_floats = 0;
    bool tail_split = true;
    bool conv_waves8 = true;     // 512-thread (8-wave) form of the 128x128 conv kernel (option "conv_waves8" = 0: the 4-wave form)
    long long bn_ema_step = 0;   // fv_set_bn_zero_debias_step: 0 plain EMA of the BN moving statistics, t >= 1 Keras 2.2.4's zero-debiased update t
    bool wgrad_fused_taps = true;   // option "wgrad_fused_taps": wgrad9_mfma.hip for the 32 -> 64 channel 3x3 layers
    bool conv_halo = true;      // option "conv_halo": conv9_mfma.hip (training forward) and dgrad9s2_mfma.hip (stride-2 data-gradient) for the 32 -> 64 channel 3x3 layers
    bool conv_small = true;      // option "conv_small": conv_small_kernel for small-M inference launches (fv_conv_small_plan)
    bool conv_bm64 = true;       // option "conv_bm64": 64-row tiles for small-M inference launches (fv_conv_bm64)
    bool conv1x1_persist = true; // option "conv1x1_persist": conv1x1_mfma.hip for 1x1 launches with more than 512 tiles
    bool conv0_direct = true;    // option "conv0_direct": vector-FMA first layer (conv0_direct.hip) instead of the gather kernel
    ~fv_ctx();
};

// RAII: brackets one kernel launch with HIP events when profiling is enabled.
struct FvProfScope {
    fv_ctx* ctx;
    hipEvent_t e1 = nullptr;
    FvProfScope(fv_ctx* c, const char* name, double flops, double bytes);
    // shape-tagged form: `tag` is appended to the name when fv_profile_enable(ctx, 2) is active
    FvProfScope(fv_ctx* c, const char* name, const std::string& tag, double flops, double bytes);
    void begin(fv_ctx* c, const char* name, double flops, double bytes);
    ~FvProfScope();
};

int fv_fail(fv_ctx* ctx, int code, const char* fmt, ...);

#define FV_HIP(ctx, expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fv_fail((ctx), FV_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                           hipGetErrorString(_e), __FILE__, __LINE__);                       \
    } while (0)

#define FV_LAUNCH_CHECK(ctx) FV_HIP(ctx, hipGetLastError())

#define FV_REQUIRE(ctx, cond, ...)                                                           \
    do {                                                                                     \
        if (!(cond)) return fv_fail((ctx), FV_ERR_INVALID, __VA_ARGS__);                     \
    } while (0)
