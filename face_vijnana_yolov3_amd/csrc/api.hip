// Context management of the C ABI (include/fv_hotpath.h).
#include <cstdarg>
#include <cstdlib>
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

fv_ctx::~fv_ctx() {
    for (auto& r : prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    for (auto* n : prof_names) delete n;
    for (int i = 0; i < 2; ++i) {
        if (ev_dz[i]) (void)hipEventDestroy(ev_dz[i]);
        if (ev_wg[i]) (void)hipEventDestroy(ev_wg[i]);
    }
    if (side) (void)hipStreamDestroy(side);
}

static hipEvent_t take_event(fv_ctx* c) {
    hipEvent_t e = nullptr;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void FvProfScope::begin(fv_ctx* c, const char* name, double flops, double bytes) {
    if (!c || !c->prof_on) return;
    hipEvent_t e0 = take_event(c);
    e1 = take_event(c);
    if (!e0 || !e1) { e1 = nullptr; return; }
    (void)hipEventRecord(e0, c->stream);
    c->prof.push_back(FvProfRec{name, flops, bytes, e0, e1});
}
FvProfScope::FvProfScope(fv_ctx* c, const char* name, double flops, double bytes) : ctx(c) { begin(c, name, flops, bytes); }
FvProfScope::FvProfScope(fv_ctx* c, const char* name, const std::string& tag, double flops, double bytes) : ctx(c) {
    if (c && c->prof_on && c->prof_shapes) {
        const std::string full = std::string(name) + " " + tag;
        for (auto* s : c->prof_names) if (*s == full) { begin(c, s->c_str(), flops, bytes); return; }
        c->prof_names.push_back(new std::string(full));
        begin(c, c->prof_names.back()->c_str(), flops, bytes);
        return;
    }
    begin(c, name, flops, bytes);
}
FvProfScope::~FvProfScope() {
    if (e1) (void)hipEventRecord(e1, ctx->stream);
}

extern "C" {

int fv_abi_version(void) { return 3; }   // 3: fv_set_option / fv_get_option replace the per-switch setters; the opt-in batch-1 variants are gone

// Tuning options (include/fv_hotpath.h, "tuning"): every switch is a bool member of the context; all default to on.
namespace {
struct FvOption { const char* key; bool fv_ctx::*member; };
const FvOption kOptions[] = {
    {"overlap", &fv_ctx::overlap},
    {"tail_split", &fv_ctx::tail_split},
    {"conv_waves8", &fv_ctx::conv_waves8},
    {"conv1x1_persist", &fv_ctx::conv1x1_persist},
    {"conv_bm64", &fv_ctx::conv_bm64},
    {"conv_small", &fv_ctx::conv_small},
    {"conv_halo", &fv_ctx::conv_halo},
    {"conv0_direct", &fv_ctx::conv0_direct},
    {"wgrad_fused_taps", &fv_ctx::wgrad_fused_taps},
};
const FvOption* find_option(const char* key) {
    if (!key) return nullptr;
    for (const auto& o : kOptions)
        if (std::string(o.key) == key) return &o;
    return nullptr;
}
}  // namespace

int fv_set_option(fv_ctx* ctx, const char* key, long long value) {
    if (!ctx) return FV_ERR_INVALID;
    const FvOption* o = find_option(key);
    if (!o) return fv_fail(ctx, FV_ERR_INVALID, "fv_set_option: unknown option '%s'", key ? key : "(null)");
    ctx->*(o->member) = value != 0;
    return FV_OK;
}

int fv_get_option(fv_ctx* ctx, const char* key, long long* value) {
    if (!ctx || !value) return FV_ERR_INVALID;
    const FvOption* o = find_option(key);
    if (!o) return fv_fail(ctx, FV_ERR_INVALID, "fv_get_option: unknown option '%s'", key ? key : "(null)");
    *value = ctx->*(o->member) ? 1 : 0;
    return FV_OK;
}

int fv_set_bucket_on_side(fv_ctx* ctx, int on) {
    if (!ctx) return FV_ERR_INVALID;
    ctx->bucket_on_side = on != 0;
    return FV_OK;
}

void* fv_side_stream(fv_ctx* ctx) { return ctx ? (void*)ctx->side : nullptr; }

int fv_set_bn_zero_debias_step(fv_ctx* ctx, long long step) {
    if (!ctx || step < 0) return FV_ERR_INVALID;
    ctx->bn_ema_step = step;
    return FV_OK;
}

int fv_set_conv_scratch(fv_ctx* ctx, void* buf, size_t bytes) {
    if (!ctx) return FV_ERR_INVALID;
    ctx->tail_slab = buf ? (float*)buf : nullptr;
    ctx->tail_slab_floats = buf ? (long long)(bytes / sizeof(float)) : 0;
    return FV_OK;
}

int fv_profile_enable(fv_ctx* ctx, int on) {
    if (!ctx) return FV_ERR_INVALID;
    ctx->prof_on = on != 0;
    ctx->prof_shapes = on == 2;
    return FV_OK;
}

int fv_profile_collect(fv_ctx* ctx, fv_profile_rec* out, int max_recs, int* n_out) {
    if (!ctx || !n_out) return FV_ERR_INVALID;
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<fv_profile_rec> agg;
    for (auto& r : ctx->prof) {
        float ms = 0.f;
        FV_HIP(ctx, hipEventElapsedTime(&ms, r.e0, r.e1));
        size_t k = 0;
        for (; k < agg.size(); ++k) if (std::string(agg[k].name) == r.name) break;
        if (k == agg.size()) {
            fv_profile_rec a{};
            snprintf(a.name, sizeof a.name, "%s", r.name);
            agg.push_back(a);
        }
        agg[k].launches += 1; agg[k].ms_total += ms; agg[k].flops_total += r.flops; agg[k].bytes_total += r.bytes;
        ctx->ev_pool.push_back(r.e0); ctx->ev_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    *n_out = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < max_recs; ++i) out[i] = agg[i];
    return FV_OK;
}

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
    // FV_OPTIONS="key=0,key=1,...": initial values of the tuning options (A/B runs of unmodified programs)
    if (const char* e = getenv("FV_OPTIONS")) {
        std::string spec(e);
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t end = spec.find(',', pos);
            if (end == std::string::npos) end = spec.size();
            const std::string item = spec.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq != std::string::npos)
                if (const FvOption* o = find_option(item.substr(0, eq).c_str())) c->*(o->member) = item[eq + 1] != '0';
            pos = end + 1;
        }
    }
    c->stream = (hipStream_t)stream;
    // The side stream carries the weight-gradient kernels of the backward overlap at the LOWEST stream priority:
    // the dispatcher then serves the data-gradient / BN-backward chain (the critical path) first and the weight
    // gradients fill what is left -- measured 675 -> 683 img/s against a default-priority side stream (highest: no
    // change).  FV_SIDE_PRIORITY=default|high overrides (A/B knob).
    bool ok;
    {
        const char* pr = getenv("FV_SIDE_PRIORITY");
        int least = 0, greatest = 0;
        const bool range = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        if (range && !(pr && pr[0] == 'd'))
            ok = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, (pr && pr[0] == 'h') ? greatest : least) == hipSuccess;
        else
            ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
    }
    for (int i = 0; i < 2 && ok; ++i)
        ok = hipEventCreateWithFlags(&c->ev_dz[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_wg[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete c; return fv_fail(nullptr, FV_ERR_HIP, "fv_create: could not create the side stream / events"); }
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
