// Shared internals of libfv_hotpath (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include "../../include/fv_hotpath.h"

struct fv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
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
