// nae_stream.hip — SoundTouch-shaped and spectrum streaming handles on top of the block kernels.
//
// nae_stretch mirrors the calls soundtouch_process_payload makes (/root/reference/src/processor/
// audio-velocity.cpp:369-428): putSamples / numSamples / receiveSamples / flush.
#include "nae_internal.h"
#include <new>
#include <string.h>

namespace {

struct DevBuf {
    float* p = nullptr;
    size_t cap = 0; // floats
    size_t len = 0; // floats in use
};

int devbuf_reserve(nae_ctx* ctx, DevBuf& b, size_t want)
{
    if (want <= b.cap) return NAE_OK;
    size_t cap = b.cap ? b.cap : 1 << 16;
    while (cap < want) cap *= 2;
    float* np = nullptr;
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

void devbuf_free(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b = DevBuf{};
}

} // namespace

struct nae_stretch {
    nae_ctx* ctx;
    int sample_rate, ch;
    double rate, pitch;
    DevBuf in, out;
    size_t out_read = 0; // sample-frames already handed out
    size_t out_frames = 0;
    bool flushed = false;
};

struct nae_spectrum {
    nae_ctx* ctx;
    int ch;
    DevBuf pending;      // interleaved samples not yet covered by a complete hop
    DevBuf out;          // [frames][ch][513]
    size_t out_read = 0; // frames handed out
    size_t out_frames = 0;
};

extern "C" {

int nae_stretch_create(nae_ctx* ctx, int sample_rate, int channels, float rate, float pitch, nae_stretch** h)
{
    if (!ctx || !h) return NAE_ERR_INVALID;
    *h = nullptr;
    // audio-velocity.cpp:371-379 rejects rates outside 8..48 kHz for SoundTouch; the vocoder has no such
    // limit, but the drop-in keeps the reference's envelope (lift it with sample_rate = 0).
    if (sample_rate != 0 && (sample_rate < 8000 || sample_rate > 48000)) return nae_fail(ctx, NAE_ERR_UNSUPPORTED, "Unsupported sample rate: requires 8000..48000 Hz");
    if (channels != 1 && channels != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    nae_stretch_plan pl;
    int rc = nae_stretch_plan_make(rate, pitch, 0, &pl);
    if (rc) return nae_fail(ctx, rc, "rate/pitch outside the supported range");
    nae_stretch* s = new (std::nothrow) nae_stretch();
    if (!s) return NAE_ERR_NOMEM;
    s->ctx = ctx;
    s->sample_rate = sample_rate;
    s->ch = channels;
    s->rate = rate;
    s->pitch = pitch;
    *h = s;
    return NAE_OK;
}

static int stretch_append(nae_stretch* h, const float* p, size_t S, bool host)
{
    if (!h || (S && !p)) return NAE_ERR_INVALID;
    if (h->flushed) return nae_fail(h->ctx, NAE_ERR_STATE, "put after flush");
    if (S == 0) return NAE_OK;
    const size_t n = S * h->ch;
    int rc = devbuf_reserve(h->ctx, h->in, h->in.len + n);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(h->in.p + h->in.len, p, n * sizeof(float), host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->ctx->stream);
    if (e != hipSuccess) return nae_check(h->ctx, e, "hipMemcpyAsync(put)");
    if (host) (void)hipStreamSynchronize(h->ctx->stream); // the caller may reuse its buffer
    h->in.len += n;
    return NAE_OK;
}

int nae_stretch_put(nae_stretch* h, const float* interleaved, size_t S) { return stretch_append(h, interleaved, S, false); }
int nae_stretch_put_host(nae_stretch* h, const float* interleaved, size_t S) { return stretch_append(h, interleaved, S, true); }

// v1: the whole stream is transformed when flush() arrives (identical samples to the block call; the
// incremental form that bounds memory is DESIGN.md §6 "next").
int nae_stretch_flush(nae_stretch* h)
{
    if (!h) return NAE_ERR_INVALID;
    if (h->flushed) return NAE_OK;
    h->flushed = true;
    const size_t L = h->in.len / h->ch;
    nae_stretch_plan pl;
    int rc = nae_stretch_plan_make(h->rate, h->pitch, L, &pl);
    if (rc) return rc;
    h->out_frames = pl.out_len;
    if (pl.out_len == 0) return NAE_OK;
    rc = devbuf_reserve(h->ctx, h->out, pl.out_len * h->ch);
    if (rc) return rc;
    h->out.len = pl.out_len * h->ch;
    nae_sig src{h->in.p, 0, 1, (size_t)h->ch}, dst{h->out.p, 0, 1, (size_t)h->ch};
    return nae_stretch_block_f32(h->ctx, h->rate, h->pitch, &src, L, h->ch, 1, &dst);
}

size_t nae_stretch_available(nae_stretch* h) { return h ? h->out_frames - h->out_read : 0; }

static int stretch_take(nae_stretch* h, float* dst, size_t max_frames, size_t* got, bool host)
{
    if (!h || !got || (max_frames && !dst)) return NAE_ERR_INVALID;
    size_t n = h->out_frames - h->out_read;
    if (n > max_frames) n = max_frames;
    *got = n;
    if (n == 0) return NAE_OK;
    hipError_t e = hipMemcpyAsync(dst, h->out.p + h->out_read * h->ch, n * h->ch * sizeof(float),
                                  host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->ctx->stream);
    if (e != hipSuccess) return nae_check(h->ctx, e, "hipMemcpyAsync(receive)");
    if (host) {
        e = hipStreamSynchronize(h->ctx->stream);
        if (e != hipSuccess) return nae_check(h->ctx, e, "hipStreamSynchronize");
    }
    h->out_read += n;
    return NAE_OK;
}

int nae_stretch_receive(nae_stretch* h, float* dst, size_t max_frames, size_t* got) { return stretch_take(h, dst, max_frames, got, false); }
int nae_stretch_receive_host(nae_stretch* h, float* dst, size_t max_frames, size_t* got) { return stretch_take(h, dst, max_frames, got, true); }

int nae_stretch_destroy(nae_stretch* h)
{
    if (!h) return NAE_OK;
    (void)hipStreamSynchronize(h->ctx->stream);
    devbuf_free(h->in);
    devbuf_free(h->out);
    delete h;
    return NAE_OK;
}

// ------------------------------------------------------------------------------------------------ spectrum
int nae_spectrum_create(nae_ctx* ctx, int n_fft, int hop, int channels, nae_spectrum** h)
{
    if (!ctx || !h) return NAE_ERR_INVALID;
    *h = nullptr;
    if (n_fft != NAE_FFT_N || hop != NAE_HOP) return nae_fail(ctx, NAE_ERR_UNSUPPORTED, "only N = 1024, hop = 256 is implemented");
    if (channels != 1 && channels != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    nae_spectrum* s = new (std::nothrow) nae_spectrum();
    if (!s) return NAE_ERR_NOMEM;
    s->ctx = ctx;
    s->ch = channels;
    *h = s;
    return NAE_OK;
}

int nae_spectrum_put(nae_spectrum* h, const float* interleaved, size_t S)
{
    if (!h || (S && !interleaved)) return NAE_ERR_INVALID;
    if (S == 0) return NAE_OK;
    nae_ctx* ctx = h->ctx;
    const size_t n = S * h->ch;
    int rc = devbuf_reserve(ctx, h->pending, h->pending.len + n);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(h->pending.p + h->pending.len, interleaved, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(put)");
    h->pending.len += n;
    const size_t T = h->pending.len / h->ch;
    const size_t F = nae_spectrum_frames(T);
    if (F == 0) return NAE_OK;
    const size_t rec = (size_t)h->ch * NAE_FFT_BINS;
    // compact what has been read, then append the new frames
    if (h->out_read) {
        const size_t keep = (h->out_frames - h->out_read) * rec;
        if (keep) {
            // overlapping forward move is not safe with memcpy: stage through a fresh buffer only when needed
            DevBuf nb;
            rc = devbuf_reserve(ctx, nb, keep + F * rec);
            if (rc) return rc;
            e = hipMemcpyAsync(nb.p, h->out.p + h->out_read * rec, keep * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) { devbuf_free(nb); return nae_check(ctx, e, "hipMemcpyAsync(compact)"); }
            (void)hipStreamSynchronize(ctx->stream);
            devbuf_free(h->out);
            h->out = nb;
            h->out.len = keep;
        } else
            h->out.len = 0;
        h->out_frames -= h->out_read;
        h->out_read = 0;
    }
    rc = devbuf_reserve(ctx, h->out, h->out.len + F * rec);
    if (rc) return rc;
    nae_sig src{h->pending.p, 0, 1, (size_t)h->ch};
    rc = nae_spectrum_block_f32(ctx, &src, T, h->ch, 1, h->out.p + h->out.len, 0);
    if (rc) return rc;
    h->out.len += F * rec;
    h->out_frames += F;
    // keep the samples the next frame still needs: everything from F*hop on
    const size_t drop = F * NAE_HOP * h->ch;
    const size_t tail = h->pending.len - drop;
    DevBuf nb;
    rc = devbuf_reserve(ctx, nb, tail ? tail : 1);
    if (rc) return rc;
    if (tail) {
        e = hipMemcpyAsync(nb.p, h->pending.p + drop, tail * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { devbuf_free(nb); return nae_check(ctx, e, "hipMemcpyAsync(tail)"); }
    }
    (void)hipStreamSynchronize(ctx->stream);
    devbuf_free(h->pending);
    h->pending = nb;
    h->pending.len = tail;
    return NAE_OK;
}

size_t nae_spectrum_available(nae_spectrum* h) { return h ? h->out_frames - h->out_read : 0; }

int nae_spectrum_receive(nae_spectrum* h, float* dst, size_t max_frames, size_t* got)
{
    if (!h || !got || (max_frames && !dst)) return NAE_ERR_INVALID;
    size_t n = h->out_frames - h->out_read;
    if (n > max_frames) n = max_frames;
    *got = n;
    if (n == 0) return NAE_OK;
    const size_t rec = (size_t)h->ch * NAE_FFT_BINS;
    hipError_t e = hipMemcpyAsync(dst, h->out.p + h->out_read * rec, n * rec * sizeof(float), hipMemcpyDeviceToDevice, h->ctx->stream);
    if (e != hipSuccess) return nae_check(h->ctx, e, "hipMemcpyAsync(receive)");
    h->out_read += n;
    return NAE_OK;
}

int nae_spectrum_destroy(nae_spectrum* h)
{
    if (!h) return NAE_OK;
    (void)hipStreamSynchronize(h->ctx->stream);
    devbuf_free(h->pending);
    devbuf_free(h->out);
    delete h;
    return NAE_OK;
}

} // extern "C"
