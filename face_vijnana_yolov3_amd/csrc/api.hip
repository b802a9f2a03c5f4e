// Context management of the C ABI (include/fv_hotpath.h).
#include <cstdarg>
#include "common.h"

static thread_local std::string g_create_err;

int fv_fail(fv_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

extern "C" {

int fv_abi_version(void) { return 1; }

int fv_create(int device, void* stream, fv_ctx** out) {
    if (!out) return fv_fail(nullptr, FV_ERR_INVALID, "fv_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fv_fail(nullptr, FV_ERR_HIP, "fv_create: no HIP device visible (%s)",
                       e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fv_fail(nullptr, FV_ERR_INVALID, "fv_create: device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fv_fail(nullptr, FV_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fv_fail(nullptr, FV_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fv_fail(nullptr, FV_ERR_INVALID, "fv_create: device is %s; this library is built for gfx950 only",
                       prop.gcnArchName);
    fv_ctx* c = new fv_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    *out = c;
    return FV_OK;
}

void fv_destroy(fv_ctx* ctx) { delete ctx; }

const char* fv_last_error(const fv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int fv_set_stream(fv_ctx* ctx, void* stream) {
    if (!ctx) return FV_ERR_INVALID;
    ctx->stream = (hipStream_t)stream;
    return FV_OK;
}

}  // extern "C"
