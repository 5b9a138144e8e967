// nae_stream.hip — SoundTouch-shaped and spectrum streaming handles on top of the block kernels.
//
// nae_stretch mirrors the calls soundtouch_process_payload makes (/root/reference/src/processor/
// audio-velocity.cpp:369-428): putSamples / numSamples / receiveSamples / flush.
#include "nae_internal.h"
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>

#include "stream_util.h"

struct nae_stretch {
    nae_ctx* ctx;
    int sample_rate, ch;
    double rate, pitch;
    nae_stretch_plan pl{};        // parameters (in_len = 0)
    DevFifo in;                   // interleaved input, sample-frames [in.base, in_total)
    size_t in_total = 0;
    // phase vocoder
    size_t blocks_done = 0;       // hop blocks produced == frames folded into the carried phase
    uint32_t* carry[2] = {nullptr, nullptr};
    int carry_cur = 0;
    DevFifo mid;                  // planar stretched signal, per-channel capacity mid_cap, samples [mid.base, mid_total)
    size_t mid_cap = 0, mid_total = 0;
    // output
    DevFifo out;                  // interleaved result, sample-frames [out.base, out_total)
    size_t out_total = 0, out_read = 0;
    bool flushed = false;
};

struct nae_spectrum {
    nae_ctx* ctx;
    int ch;
    // ping-pong pairs: the live data always moves into the OTHER buffer of its pair (an in-place forward move would
    // overlap), and nothing is allocated, freed or waited for once the buffers have grown to their working size
    DevBuf pending, pending_alt;   // interleaved samples not yet covered by a complete hop
    DevBuf out, out_alt;           // [frames][ch][513]
    size_t out_read = 0;           // frames handed out
    size_t out_frames = 0;
};

namespace {

inline long long frame_start_host(const nae_stretch_plan& pl, long long f)
{
    return (((f - 1) * pl.ha_q24 + (1ll << (NAE_HA_FRAC_BITS - 1))) >> NAE_HA_FRAC_BITS) - NAE_FFT_N / 2;
}

// number of leading frames whose 1024-sample window lies inside [.., in_total)
size_t frames_available(const nae_stretch_plan& pl, size_t in_total)
{
    if (in_total < NAE_FFT_N / 2) return 0;
    // estimate, then correct with the exact start formula
    long long f = (long long)(((double)in_total - 512.0) / ((double)pl.ha_q24 / (double)(1 << NAE_HA_FRAC_BITS))) + 2;
    if (f < 0) f = 0;
    while (f > 0 && frame_start_host(pl, f - 1) + NAE_FFT_N > (long long)in_total) f--;
    while (frame_start_host(pl, f) + NAE_FFT_N <= (long long)in_total) f++;
    return (size_t)f;
}

int stretch_process(nae_stretch* h)
{
    nae_ctx* ctx = h->ctx;
    const int ch = h->ch;
    const nae_stretch_plan& pl = h->pl;
    nae_stretch_plan fin{};
    if (h->flushed) {
        int rc = nae_stretch_plan_make(h->rate, h->pitch, h->in_total, &fin);
        if (rc) return rc;
    }
    // ---- neither stage: the node is a wire
    if (!pl.pv_on && !pl.rs_on) {
        const size_t n = h->in_total - h->out_total;
        if (n) {
            int rc = fifo_reserve_interleaved(ctx, h->out, h->out_total, h->in_total, ch);
            if (rc) return rc;
            hipError_t e = hipMemcpyAsync(h->out.cur.p + (h->out_total - h->out.base) * ch, h->in.cur.p + (h->out_total - h->in.base) * ch,
                                          n * ch * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(wire)");
            h->out_total = h->in_total;
            return fifo_drop_interleaved(ctx, h->in, h->in_total, h->in_total, ch);
        }
        return NAE_OK;
    }
    // ---- transposer first (rate_eff > 1): in -> [RS] -> mid (interleaved FIFO) -> [PV] -> out
    if (pl.rs_first) {
        size_t J_r;
        if (h->flushed) J_r = fin.mid_len;
        else if (h->in_total <= NAE_RS_TAPS / 2) J_r = 0;
        else {
            const unsigned __int128 lim = ((unsigned __int128)(h->in_total - NAE_RS_TAPS / 2) << 32) - 1;
            J_r = (size_t)(lim / pl.step_q32) + 1;
        }
        if (J_r > h->mid_total) {
            int rc = nae_ensure_rs_table(ctx, pl.rate_eff);
            if (rc) return rc;
            rc = fifo_reserve_interleaved(ctx, h->mid, h->mid_total, J_r, ch);
            if (rc) return rc;
            nae_sig src{h->in.cur.p - (ptrdiff_t)h->in.base * ch, 0, 1, (size_t)ch};
            nae_sig dst{h->mid.cur.p - (ptrdiff_t)h->mid.base * ch, 0, 1, (size_t)ch};
            rc = nae_launch_resample(ctx, &pl, &src, h->in_total, ch, 1, ctx->d_rs_tab, &dst, h->mid_total, J_r);
            if (rc) return rc;
            h->mid_total = J_r;
            const unsigned __int128 pos = (unsigned __int128)J_r * pl.step_q32;
            const long long need_from = ((long long)(pos >> 32) - (NAE_RS_TAPS / 2 - 1)) & ~3ll;
            const size_t nb = need_from > 0 ? (size_t)need_from : 0;
            rc = fifo_drop_interleaved(ctx, h->in, nb < h->in_total ? nb : h->in_total, h->in_total, ch);
            if (rc) return rc;
        }
        size_t F_r, B_r;
        long long out_limit;
        if (h->flushed) {
            F_r = fin.frames;
            B_r = (fin.out_len + NAE_HOP - 1) / NAE_HOP;
            out_limit = (long long)fin.out_len;
        } else {
            F_r = frames_available(pl, h->mid_total);
            B_r = F_r >= 3 ? F_r - 3 : 0;
            out_limit = (long long)1 << 60;
        }
        if (B_r > h->blocks_done) {
            const size_t count = B_r - h->blocks_done;
            // a short segment (what a node's batch of waiting frames gives) is ONE tile run frame-interleaved — four consecutive
            // frames per step — and the pipeline itself hands the phase on: one launch instead of pass 1 + scan + pass 3.
            // Long segments (a whole file in one put) are cut into 64-frame tiles that run side by side.
            const bool one_tile = ctx->pv_tile <= 0 && count <= 256;
            const int tile = one_tile ? (int)count : (ctx->pv_tile > 0 ? ctx->pv_tile : 64);
            const int fps = one_tile ? 4 : 1;
            int rc = nae_ws_reserve(ctx, &ctx->ws_phase, &ctx->ws_phase_bytes, nae_pv_phase_workspace_bytes(count, ch, 1, tile));
            if (rc) return rc;
            for (int i = 0; i < 2; i++)
                if (!h->carry[i] && hipMalloc((void**)&h->carry[i], (size_t)ch * kPhasePad * sizeof(uint32_t)) != hipSuccess)
                    return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(carry)");
            nae_pv_segment seg{(long long)h->blocks_done, (long long)count, (long long)F_r, out_limit,
                               h->blocks_done ? h->carry[h->carry_cur] : nullptr, h->carry[h->carry_cur ^ 1], one_tile};
            const size_t produced_total = h->flushed ? fin.out_len : B_r * NAE_HOP;
            rc = fifo_reserve_interleaved(ctx, h->out, h->out_total, produced_total, ch);
            if (rc) return rc;
            nae_sig src{h->mid.cur.p - (ptrdiff_t)h->mid.base * ch, 0, 1, (size_t)ch};
            nae_sig dst{h->out.cur.p - (ptrdiff_t)h->out.base * ch, 0, 1, (size_t)ch};
            rc = nae_launch_pv_phase(ctx, &pl, &src, h->mid_total, ch, 1, tile, tile, static_cast<uint32_t*>(ctx->ws_phase), &seg);
            if (rc) return rc;
            rc = nae_launch_pv_synth(ctx, &pl, &src, h->mid_total, ch, 1, tile, tile, static_cast<const uint32_t*>(ctx->ws_phase), &dst, &seg, fps);
            if (rc) return rc;
            h->carry_cur ^= 1;
            h->blocks_done = B_r;
            h->out_total = produced_total;
            const long long s_keep = frame_start_host(pl, (long long)B_r - 1);
            rc = fifo_drop_interleaved(ctx, h->mid, s_keep > 0 ? (size_t)s_keep : 0, h->mid_total, ch);
            if (rc) return rc;
        }
        return NAE_OK;
    }
    // ---- vocoder first: in -> [PV] -> mid (planar FIFO) -> [RS] -> out
    // ---- stage 1: phase vocoder over the hop blocks that became computable
    if (pl.pv_on) {
        size_t F_r, B_r;
        long long mid_limit;
        if (h->flushed) {
            F_r = fin.frames;
            B_r = (fin.mid_len + NAE_HOP - 1) / NAE_HOP;
            mid_limit = (long long)fin.mid_len;
        } else {
            F_r = frames_available(pl, h->in_total);
            B_r = F_r >= 3 ? F_r - 3 : 0;
            mid_limit = (long long)1 << 60;
        }
        if (B_r > h->blocks_done) {
            const size_t count = B_r - h->blocks_done;
            // a short segment (what a node's batch of waiting frames gives) is ONE tile run frame-interleaved — four consecutive
            // frames per step — and the pipeline itself hands the phase on: one launch instead of pass 1 + scan + pass 3.
            // Long segments (a whole file in one put) are cut into 64-frame tiles that run side by side.
            const bool one_tile = ctx->pv_tile <= 0 && count <= 256;
            const int tile = one_tile ? (int)count : (ctx->pv_tile > 0 ? ctx->pv_tile : 64);
            const int fps = one_tile ? 4 : 1;
            int rc = nae_ws_reserve(ctx, &ctx->ws_phase, &ctx->ws_phase_bytes, nae_pv_phase_workspace_bytes(count, ch, 1, tile));
            if (rc) return rc;
            for (int i = 0; i < 2; i++)
                if (!h->carry[i] && hipMalloc((void**)&h->carry[i], (size_t)ch * kPhasePad * sizeof(uint32_t)) != hipSuccess)
                    return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(carry)");
            nae_pv_segment seg{(long long)h->blocks_done, (long long)count, (long long)F_r, mid_limit,
                               h->blocks_done ? h->carry[h->carry_cur] : nullptr, h->carry[h->carry_cur ^ 1], one_tile};
            nae_sig src{h->in.cur.p - (ptrdiff_t)h->in.base * ch, 0, 1, (size_t)ch};   // absolute indexing
            nae_sig dst;
            size_t produced_total = h->flushed ? fin.mid_len : B_r * NAE_HOP;
            if (pl.rs_on) {
                // planar mid FIFO: grow (re-pack planes) when the per-channel capacity is too small
                const size_t need = produced_total - h->mid.base;
                if (need > h->mid_cap) {
                    size_t cap = h->mid_cap ? h->mid_cap : 1 << 15;
                    while (cap < need) cap *= 2;
                    h->mid.alt.len = 0;
                    rc = devbuf_reserve(ctx, h->mid.alt, cap * ch);
                    if (rc) return rc;
                    const size_t keep = h->mid_total - h->mid.base;
                    for (int c = 0; c < ch && keep; c++) {
                        hipError_t e = hipMemcpyAsync(h->mid.alt.p + (size_t)c * cap, h->mid.cur.p + (size_t)c * h->mid_cap, keep * sizeof(float),
                                                      hipMemcpyDeviceToDevice, ctx->stream);
                        if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(mid grow)");
                    }
                    std::swap(h->mid.cur, h->mid.alt);
                    h->mid_cap = cap;
                }
                dst = nae_sig{h->mid.cur.p - (ptrdiff_t)h->mid.base, 0, h->mid_cap, 1};
            } else {
                if (h->flushed && produced_total > fin.out_len) produced_total = fin.out_len;
                rc = fifo_reserve_interleaved(ctx, h->out, h->out_total, produced_total, ch);
                if (rc) return rc;
                dst = nae_sig{h->out.cur.p - (ptrdiff_t)h->out.base * ch, 0, 1, (size_t)ch};
                if (h->flushed) seg.mid_limit = (long long)fin.out_len;
            }
            rc = nae_launch_pv_phase(ctx, &pl, &src, h->in_total, ch, 1, tile, tile, static_cast<uint32_t*>(ctx->ws_phase), &seg);
            if (rc) return rc;
            rc = nae_launch_pv_synth(ctx, &pl, &src, h->in_total, ch, 1, tile, tile, static_cast<const uint32_t*>(ctx->ws_phase), &dst, &seg, fps);
            if (rc) return rc;
            h->carry_cur ^= 1;
            h->blocks_done = B_r;
            if (pl.rs_on) h->mid_total = produced_total;
            else h->out_total = produced_total;
            // input still needed: from the start of frame B_r - 1 (it primes the next call's phase difference)
            const long long s_keep = frame_start_host(pl, (long long)B_r - 1);
            rc = fifo_drop_interleaved(ctx, h->in, s_keep > 0 ? (size_t)s_keep : 0, h->in_total, ch);
            if (rc) return rc;
        }
    }
    // ---- stage 2: rate transposer over the outputs whose 16 taps are known
    if (pl.rs_on) {
        const size_t src_avail = pl.pv_on ? h->mid_total : h->in_total;
        size_t J_r;
        if (h->flushed) J_r = fin.out_len;
        else if (src_avail <= NAE_RS_TAPS / 2) J_r = 0;
        else {
            const unsigned __int128 lim = ((unsigned __int128)(src_avail - NAE_RS_TAPS / 2) << 32) - 1;
            J_r = (size_t)(lim / pl.step_q32) + 1;
        }
        if (J_r > h->out_total) {
            int rc = nae_ensure_rs_table(ctx, pl.rate_eff);
            if (rc) return rc;
            rc = fifo_reserve_interleaved(ctx, h->out, h->out_total, J_r, ch);
            if (rc) return rc;
            nae_sig src = pl.pv_on ? nae_sig{h->mid.cur.p - (ptrdiff_t)h->mid.base, 0, h->mid_cap, 1}
                                   : nae_sig{h->in.cur.p - (ptrdiff_t)h->in.base * ch, 0, 1, (size_t)ch};
            nae_sig dst{h->out.cur.p - (ptrdiff_t)h->out.base * ch, 0, 1, (size_t)ch};
            const size_t src_len = h->flushed ? (pl.pv_on ? fin.mid_len : h->in_total) : src_avail;
            rc = nae_launch_resample(ctx, &pl, &src, src_len, ch, 1, ctx->d_rs_tab, &dst, h->out_total, J_r);
            if (rc) return rc;
            h->out_total = J_r;
            // source still needed: from idx(J_r) - 7
            const unsigned __int128 pos = (unsigned __int128)J_r * pl.step_q32;
            // the tiled kernel stages from (idx - 7) rounded DOWN to a multiple of 4 samples: keep that much
            const long long need_from = ((long long)(pos >> 32) - (NAE_RS_TAPS / 2 - 1)) & ~3ll;
            const size_t nb = need_from > 0 ? (size_t)need_from : 0;
            if (pl.pv_on) {
                if (nb > h->mid.base) {   // planar: shift every plane
                    const size_t nbase = nb < h->mid_total ? nb : h->mid_total;
                    const size_t keep = h->mid_total - nbase;
                    h->mid.alt.len = 0;
                    rc = devbuf_reserve(ctx, h->mid.alt, h->mid_cap * ch);
                    if (rc) return rc;
                    for (int c = 0; c < ch && keep; c++) {
                        hipError_t e = hipMemcpyAsync(h->mid.alt.p + (size_t)c * h->mid_cap, h->mid.cur.p + (size_t)c * h->mid_cap + (nbase - h->mid.base),
                                                      keep * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
                        if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(mid shift)");
                    }
                    std::swap(h->mid.cur, h->mid.alt);
                    h->mid.base = nbase;
                }
            } else {
                rc = fifo_drop_interleaved(ctx, h->in, nb < h->in_total ? nb : h->in_total, h->in_total, ch);
                if (rc) return rc;
            }
        }
    }
    return NAE_OK;
}

} // namespace

extern "C" {

int nae_stretch_create(nae_ctx* ctx, int sample_rate, int channels, float rate, float pitch, nae_stretch** h)
{
    if (!ctx || !h) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
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
    s->pl = pl;
    *h = s;
    return NAE_OK;
}

static int stretch_append(nae_stretch* h, const float* p, size_t S, bool host)
{
    if (!h || (S && !p)) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    if (h->flushed) return nae_fail(h->ctx, NAE_ERR_STATE, "put after flush");
    if (S == 0) return NAE_OK;
    const size_t n = S * h->ch;
    int rc = fifo_reserve_interleaved(h->ctx, h->in, h->in_total, h->in_total + S, h->ch);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(h->in.cur.p + (h->in_total - h->in.base) * h->ch, p, n * sizeof(float),
                                  host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->ctx->stream);
    if (e != hipSuccess) return nae_check(h->ctx, e, "hipMemcpyAsync(put)");
    if (host) (void)hipStreamSynchronize(h->ctx->stream); // the caller may reuse its buffer
    h->in_total += S;
    h->in.cur.len = (h->in_total - h->in.base) * h->ch;
    return stretch_process(h);
}

int nae_stretch_put(nae_stretch* h, const float* interleaved, size_t S) { return stretch_append(h, interleaved, S, false); }
int nae_stretch_put_host(nae_stretch* h, const float* interleaved, size_t S) { return stretch_append(h, interleaved, S, true); }

// everything still buffered is transformed as if the input ended here (zero padding behind the last sample);
// the samples delivered over the handle's life equal nae_stretch_block_f32 on the whole input, bit for bit
int nae_stretch_flush(nae_stretch* h)
{
    if (!h) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    if (h->flushed) return NAE_OK;
    h->flushed = true;
    return stretch_process(h);
}

size_t nae_stretch_available(nae_stretch* h) { return h ? h->out_total - h->out_read : 0; }

static int stretch_take(nae_stretch* h, float* dst, size_t max_frames, size_t* got, bool host)
{
    if (!h || !got || (max_frames && !dst)) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    size_t n = h->out_total - h->out_read;
    if (n > max_frames) n = max_frames;
    *got = n;
    if (n == 0) return NAE_OK;
    hipError_t e = hipMemcpyAsync(dst, h->out.cur.p + (h->out_read - h->out.base) * h->ch, n * h->ch * sizeof(float),
                                  host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->ctx->stream);
    if (e != hipSuccess) return nae_check(h->ctx, e, "hipMemcpyAsync(receive)");
    if (host) {
        e = hipStreamSynchronize(h->ctx->stream);
        if (e != hipSuccess) return nae_check(h->ctx, e, "hipStreamSynchronize");
    }
    h->out_read += n;
    // drop what has been handed out once it dominates the buffer
    if (h->out_read - h->out.base > (1u << 16)) return fifo_drop_interleaved(h->ctx, h->out, h->out_read, h->out_total, h->ch);
    return NAE_OK;
}

int nae_stretch_receive(nae_stretch* h, float* dst, size_t max_frames, size_t* got) { return stretch_take(h, dst, max_frames, got, false); }
int nae_stretch_receive_host(nae_stretch* h, float* dst, size_t max_frames, size_t* got) { return stretch_take(h, dst, max_frames, got, true); }

int nae_stretch_destroy(nae_stretch* h)
{
    if (!h) return NAE_OK;
    (void)nae_use_device(h->ctx);
    (void)hipStreamSynchronize(h->ctx->stream);
    fifo_free(h->in);
    fifo_free(h->mid);
    fifo_free(h->out);
    for (int i = 0; i < 2; i++)
        if (h->carry[i]) (void)hipFree(h->carry[i]);
    delete h;
    return NAE_OK;
}

// ------------------------------------------------------------------------------------------------ spectrum
int nae_spectrum_create(nae_ctx* ctx, int n_fft, int hop, int channels, nae_spectrum** h)
{
    if (!ctx || !h) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
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
    (void)nae_use_device(h->ctx);
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
    // compact what has been read (into the other buffer of the pair), then append the new frames
    if (h->out_read) {
        const size_t keep = (h->out_frames - h->out_read) * rec;
        if (keep) {
            h->out_alt.len = 0;
            rc = devbuf_reserve(ctx, h->out_alt, keep + F * rec);
            if (rc) return rc;
            e = hipMemcpyAsync(h->out_alt.p, h->out.p + h->out_read * rec, keep * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(compact)");
            std::swap(h->out, h->out_alt);
        }
        h->out.len = keep;
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
    h->pending_alt.len = 0;
    rc = devbuf_reserve(ctx, h->pending_alt, tail ? tail : 1);
    if (rc) return rc;
    if (tail) {
        e = hipMemcpyAsync(h->pending_alt.p, h->pending.p + drop, tail * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(tail)");
    }
    std::swap(h->pending, h->pending_alt);
    h->pending.len = tail;
    return NAE_OK;
}

size_t nae_spectrum_available(nae_spectrum* h) { return h ? h->out_frames - h->out_read : 0; }

int nae_spectrum_receive(nae_spectrum* h, float* dst, size_t max_frames, size_t* got)
{
    if (!h || !got || (max_frames && !dst)) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
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
    (void)nae_use_device(h->ctx);
    (void)hipStreamSynchronize(h->ctx->stream);
    devbuf_free(h->pending);
    devbuf_free(h->pending_alt);
    devbuf_free(h->out);
    devbuf_free(h->out_alt);
    delete h;
    return NAE_OK;
}

} // extern "C"
