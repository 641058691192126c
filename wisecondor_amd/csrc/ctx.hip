// Context, error reporting and small host helpers of the C ABI.
#include "ctx.h"

#include <stdarg.h>

namespace wc {

static thread_local char g_error[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace wc

extern "C" {

const char *wc_last_error(void) { return wc::g_error; }

const char *wc_version(void) { return "wisecondor_hip 0.1 (gfx950)"; }

wc_ctx *wc_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        wc::set_error("no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        wc::set_error("device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        wc::set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return nullptr;
    }
    wc_ctx *ctx = new wc_ctx();
    ctx->device = device;
    return ctx;
}

void wc_destroy(wc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (wc::DevBuf *b : ctx->all_buffers()) b->release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side2) (void)hipStreamDestroy(ctx->side2);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    for (hipEvent_t e : ctx->ts.prof_ev) (void)hipEventDestroy(e);
    if (ctx->ts.lat_exec) (void)hipGraphExecDestroy(ctx->ts.lat_exec);
    if (ctx->lat_stream) (void)hipStreamDestroy(ctx->lat_stream);
    if (ctx->ev_lat_in) (void)hipEventDestroy(ctx->ev_lat_in);
    if (ctx->ev_lat_out) (void)hipEventDestroy(ctx->ev_lat_out);
    delete ctx;
}

void wc_get_part(int64_t partnum, int64_t outof, int64_t bincount, int64_t *start, int64_t *end) {
    // wisetools.py:358-361: int(bincount / float(outof) * partnum)
    double per = (double)bincount / (double)outof;
    *start = (int64_t)(per * (double)partnum);
    *end = (int64_t)(per * (double)(partnum + 1));
}

}  // extern "C"
