// stream_util.h — grow-only device buffers and absolutely indexed device FIFOs of the streaming handles (not installed)
#pragma once
#include "nae_internal.h"
#include <utility>

struct DevBuf {
    float* p = nullptr;
    size_t cap = 0; // floats
    size_t len = 0; // floats in use
};

static inline int devbuf_reserve(nae_ctx* ctx, DevBuf& b, size_t want)
{
    if (want <= b.cap) return NAE_OK;
    size_t cap = b.cap ? b.cap : 1 << 16;
    while (cap < want) cap *= 2;
    float* np = nullptr;
    (void)hipSetDevice(ctx->device);
    if (hipMalloc((void**)&np, cap * sizeof(float)) != hipSuccess) return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(stream buffer)");
    if (b.len) {
        hipError_t e = hipMemcpyAsync(np, b.p, b.len * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(np); return nae_check(ctx, e, "hipMemcpyAsync(grow)"); }
    }
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
    }
    b.p = np;
    b.cap = cap;
    return NAE_OK;
}

static inline void devbuf_free(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b = DevBuf{};
}

// A FIFO of device samples addressed by ABSOLUTE index: element i (i >= base) lives at cur.p[(i - base) * width + ...].
// Dropping the consumed head copies the tail into the alternate buffer (regions may overlap, so never in place).
struct DevFifo {
    DevBuf cur, alt;
    size_t base = 0;   // absolute index of cur.p[0]
};

static inline void fifo_free(DevFifo& f) { devbuf_free(f.cur); devbuf_free(f.alt); }

// keep elements [new_base, total) of an interleaved FIFO (width floats per element)
static inline int fifo_drop_interleaved(nae_ctx* ctx, DevFifo& f, size_t new_base, size_t total, size_t width)
{
    if (new_base <= f.base) return NAE_OK;
    const size_t keep = total > new_base ? (total - new_base) * width : 0;
    f.alt.len = 0;
    int rc = devbuf_reserve(ctx, f.alt, keep ? keep : 1);
    if (rc) return rc;
    if (keep) {
        hipError_t e = hipMemcpyAsync(f.alt.p, f.cur.p + (new_base - f.base) * width, keep * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(fifo)");
    }
    std::swap(f.cur, f.alt);
    f.cur.len = keep;
    f.base = new_base;
    return NAE_OK;
}

// make room for elements up to `want_total` (absolute), interleaved; [f.base, used_total) is live and survives a grow
static inline int fifo_reserve_interleaved(nae_ctx* ctx, DevFifo& f, size_t used_total, size_t want_total, size_t width)
{
    f.cur.len = used_total > f.base ? (used_total - f.base) * width : 0;
    return devbuf_reserve(ctx, f.cur, (want_total > f.base ? want_total - f.base : 1) * width);
}

