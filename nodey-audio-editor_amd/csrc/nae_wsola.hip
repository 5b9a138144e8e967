// nae_wsola.hip — host side of K7 option A (SoundTouch-shaped WSOLA chain): parameters, the scalar bookkeeping of
// st_chain.h, the block entry point and the streaming handle.  Device work is in kernels_wsola.hip.
//
// Replaces soundtouch::SoundTouch as used by /root/reference/src/processor/audio-velocity.cpp:369-428.
#include "st_chain.h"
#include <math.h>
#include <new>
#include <string.h>
#include <vector>

namespace nae {

int st_cfg_make(StCfg& c, int sample_rate, int channels, double rate, double pitch)
{
    if (channels != 1 && channels != 2) return NAE_ERR_INVALID;
    if (sample_rate < 8000 || sample_rate > 48000) return NAE_ERR_UNSUPPORTED;   // audio-velocity.cpp:371-379
    if (!(rate > 0.0) || !(pitch > 0.0)) return NAE_ERR_INVALID;
    c.sr = sample_rate;
    c.ch = channels;
    c.tempo = 1.0 / pitch;
    c.rate = pitch * rate;
    if (!(c.tempo >= NAE_TEMPO_MIN && c.tempo <= NAE_TEMPO_MAX) || !(c.rate >= NAE_RATE_MIN && c.rate <= NAE_RATE_MAX))
        return NAE_ERR_UNSUPPORTED;
    c.order = (c.rate > 1.0) ? 0 : (c.rate < 1.0 ? 2 : 1);
    // sequence 90..40 ms and seek window 20..15 ms, linear in tempo over [0.5, 2], whole milliseconds; overlap 8 ms
    const double seq_k = (40.0 - 90.0) / (2.0 - 0.5), seq_c = 90.0 - seq_k * 0.5;
    const double seek_k = (15.0 - 20.0) / (2.0 - 0.5), seek_c = 20.0 - seek_k * 0.5;
    double seq = seq_c + seq_k * c.tempo;
    seq = seq < 40.0 ? 40.0 : (seq > 90.0 ? 90.0 : seq);
    double seek = seek_c + seek_k * c.tempo;
    seek = seek < 15.0 ? 15.0 : (seek > 20.0 ? 20.0 : seek);
    const int seq_ms = (int)(seq + 0.5), seek_ms = (int)(seek + 0.5);
    int ovl = (sample_rate * 8) / 1000;
    if (ovl < 16) ovl = 16;
    ovl -= ovl % 8;
    c.ovl = ovl;
    c.swl = (sample_rate * seq_ms) / 1000;
    if (c.swl < 2 * ovl) c.swl = 2 * ovl;
    c.seekl = (sample_rate * seek_ms) / 1000;
    c.body = c.swl - 2 * ovl;
    c.nominal_skip = c.tempo * (double)(c.swl - ovl);
    const int intskip = (int)(c.nominal_skip + 0.5);
    const int a = intskip + ovl, b = c.swl;
    c.sample_req = (a > b ? a : b) + c.seekl;
    c.first_skip = (int)(c.tempo * (double)ovl + 0.5 * (double)c.seekl + 0.5);
    // anti-alias filter: Hamming-windowed sinc, cutoff 0.5/rate above unity and 0.5*rate below; scaled to a 2^14
    // grid with the library's rounding half left in (float build)
    const double cutoff = c.rate > 1.0 ? 0.5 / c.rate : 0.5 * c.rate;
    const double pi = 3.14159265358979323846;
    double work[kAaLen], sum = 0.0;
    const double wc = 2.0 * pi * cutoff, tc = (2.0 * pi) / (double)kAaLen;
    for (int i = 0; i < kAaLen; i++) {
        const double t = (double)i - (double)(kAaLen / 2);
        const double arg = t * wc;
        const double h = (arg != 0.0) ? sin(arg) / arg : 1.0;
        work[i] = (0.54 + 0.46 * cos(tc * t)) * h;
        sum += work[i];
    }
    const double scale = 16384.0 / sum;
    for (int i = 0; i < kAaLen; i++) {
        double v = work[i] * scale;
        v += (v >= 0.0) ? 0.5 : -0.5;
        c.aa[i] = (float)v / 16384.0f;
    }
    return NAE_OK;
}

static void sim_td(const StCfg& c, StState& s)
{
    while (s.td_in - s.td_ip >= c.sample_req) {
        if (!s.td_begin) {
            s.td_out += c.ovl;
        } else {
            s.td_begin = false;
            s.td_skip -= (double)c.first_skip;
            if (s.td_skip <= -c.nominal_skip) s.td_skip = -c.nominal_skip;
        }
        s.td_out += c.body;
        s.td_nseq++;
        s.td_skip += c.nominal_skip;
        const int adv = (int)s.td_skip;
        s.td_skip -= (double)adv;
        s.td_ip += adv;
    }
}

static void sim_aa(const StCfg& c, StState& s)
{
    const long long n = s.aa_in - s.aa_out;
    if (n < kAaLen) return;
    long long count = n - kAaLen;
    if (c.ch == 2) {
        count &= ~1ll;
        if (count < 2) return;
    }
    s.aa_out += count;
}

static void sim_cu(const StCfg& c, StState& s, CuTable* tab)
{
    const long long n = s.cu_in - s.cu_pos, end = n - 4;
    long long used = 0;
    while (used < end) {
        if (tab) {
            tab->pos.push_back(s.cu_pos + used);
            tab->fract.push_back((float)s.cu_fract);
        }
        s.cu_out++;
        s.cu_fract += c.rate;
        const int whole = (int)s.cu_fract;
        s.cu_fract -= (double)whole;
        used += whole;
    }
    s.cu_pos += used;
}

void st_sim_put(const StCfg& c, StState& s, long long n, CuTable* tab)
{
    if (n <= 0) return;
    s.expected += (double)n / (c.rate * c.tempo);
    switch (c.order) {
    case 0:
        s.td_in += n;
        sim_td(c, s);
        if (s.td_out > s.aa_in) {
            s.aa_in = s.td_out;
            sim_aa(c, s);
            s.cu_in = s.aa_out;
            sim_cu(c, s, tab);
        }
        break;
    case 1:
        s.aa_in += n;
        sim_aa(c, s);
        s.cu_in = s.aa_out;
        sim_cu(c, s, tab);
        s.td_in = s.cu_out;
        sim_td(c, s);
        break;
    default:
        s.cu_in += n;
        sim_cu(c, s, tab);
        s.aa_in = s.cu_out;
        sim_aa(c, s);
        s.td_in = s.aa_out;
        sim_td(c, s);
        break;
    }
}

long long st_final_out(const StCfg& c, const StState& s) { return c.order == 0 ? s.cu_out : s.td_out; }

// the zero puts of SoundTouch::flush: 128 frames at a time until `still` frames are available, at most 200 times
static long long sim_flush(const StCfg& c, StState& s, long long received, CuTable* tab, long long* zeros_fed)
{
    long long still = (long long)(s.expected + 0.5) - received;
    if (still < 0) still = 0;
    long long fed = 0;
    for (int i = 0; still > st_final_out(c, s) - received && i < 200; i++) {
        st_sim_put(c, s, 128, tab);
        fed += 128;
    }
    if (zeros_fed) *zeros_fed = fed;
    const long long avail = st_final_out(c, s) - received;
    return avail < still ? avail : still;
}

static int upload_cu(nae_ctx* ctx, const CuTable& tab, size_t first, size_t count, long long* d_pos, float* d_fract)
{
    if (!count) return NAE_OK;
    hipError_t e = hipMemcpyAsync(d_pos + first, tab.pos.data() + first, count * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_fract + first, tab.fract.data() + first, count * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the host vectors may move after this call
    return nae_check(ctx, e, "hipMemcpyAsync(cubic table)");
}

} // namespace nae

using namespace nae;

// plan cache of the block entry point (one entry: the bench and batch jobs repeat one parameter set)
struct nae_wsola_cache {
    bool valid = false;
    int sr = 0, ch = 0;
    double rate = 0, pitch = 0;
    size_t in_len = 0;
    StCfg cfg;
    StState fin;
    long long out_len = 0;
    long long* d_pos = nullptr;
    float* d_fract = nullptr;
    size_t tab_cap = 0;
    int* d_tile_n = nullptr;        // first cubic output of every fused filter+cubic tile (orders 0 and 1)
    size_t tile_cap = 0, n_tiles = 0;
    void* ws_a = nullptr; size_t ws_a_bytes = 0;
    void* ws_b = nullptr; size_t ws_b_bytes = 0;
};

static nae_wsola_cache* cache_of(nae_ctx* ctx)
{
    if (!ctx->wsola_cache) ctx->wsola_cache = new (std::nothrow) nae_wsola_cache();
    return ctx->wsola_cache;
}

void nae_wsola_cache_free(nae_ctx* ctx)
{
    nae_wsola_cache* c = ctx->wsola_cache;
    if (!c) return;
    if (c->d_pos) (void)hipFree(c->d_pos);
    if (c->d_fract) (void)hipFree(c->d_fract);
    if (c->d_tile_n) (void)hipFree(c->d_tile_n);
    if (c->ws_a) (void)hipFree(c->ws_a);
    if (c->ws_b) (void)hipFree(c->ws_b);
    delete c;
    ctx->wsola_cache = nullptr;
}

static int plan_fill(const StCfg& cfg, const StState& fin, long long out_len, size_t in_len, long long zeros, nae_wsola_plan* pl)
{
    memset(pl, 0, sizeof *pl);
    pl->sample_rate = cfg.sr;
    pl->channels = cfg.ch;
    pl->rate_eff = cfg.rate;
    pl->tempo_eff = cfg.tempo;
    pl->order = cfg.order;
    pl->overlap_len = cfg.ovl;
    pl->seq_len = cfg.swl;
    pl->seek_len = cfg.seekl;
    pl->sample_req = cfg.sample_req;
    pl->nominal_skip = cfg.nominal_skip;
    pl->in_len = in_len;
    pl->flush_zeros = (size_t)zeros;
    pl->n_seq = (size_t)fin.td_nseq;
    pl->td_out_len = (size_t)fin.td_out;
    pl->aa_out_len = (size_t)fin.aa_out;
    pl->cu_out_len = (size_t)fin.cu_out;
    pl->out_len = (size_t)out_len;
    return NAE_OK;
}

extern "C" {

int nae_wsola_plan_make(int sample_rate, int channels, double rate, double pitch, size_t in_len, nae_wsola_plan* plan)
{
    if (!plan) return NAE_ERR_INVALID;
    StCfg cfg;
    int rc = st_cfg_make(cfg, sample_rate, channels, rate, pitch);
    if (rc) return rc;
    StState s;
    st_sim_put(cfg, s, (long long)in_len, nullptr);
    long long zeros = 0;
    const long long out_len = sim_flush(cfg, s, 0, nullptr, &zeros);
    return plan_fill(cfg, s, out_len, in_len, zeros, plan);
}

int nae_wsola_block_f32(nae_ctx* ctx, int sample_rate, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, const nae_sig* dst, int32_t* offsets_dbg)
{
    if (!ctx) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    if (!src || !dst || !src->base || !dst->base) return nae_fail(ctx, NAE_ERR_INVALID, "nae_wsola_block_f32: null signal");
    nae_wsola_cache* wc = cache_of(ctx);
    if (!wc) return nae_fail(ctx, NAE_ERR_NOMEM, "plan cache");
    int rc;
    if (!(wc->valid && wc->sr == sample_rate && wc->ch == ch && wc->rate == rate && wc->pitch == pitch && wc->in_len == in_len)) {
        wc->valid = false;
        rc = st_cfg_make(wc->cfg, sample_rate, ch, rate, pitch);
        if (rc) return nae_fail(ctx, rc, "nae_wsola_block_f32: parameters");
        CuTable tab;
        StState s;
        st_sim_put(wc->cfg, s, (long long)in_len, &tab);
        wc->out_len = sim_flush(wc->cfg, s, 0, &tab, nullptr);
        wc->fin = s;
        if (tab.pos.size() > wc->tab_cap) {
            (void)hipStreamSynchronize(ctx->stream);
            if (wc->d_pos) (void)hipFree(wc->d_pos);
            if (wc->d_fract) (void)hipFree(wc->d_fract);
            wc->d_pos = nullptr; wc->d_fract = nullptr; wc->tab_cap = 0;
            const size_t cap = tab.pos.size() + tab.pos.size() / 8 + 1024;
            if (hipMalloc((void**)&wc->d_pos, cap * sizeof(long long)) != hipSuccess ||
                hipMalloc((void**)&wc->d_fract, cap * sizeof(float)) != hipSuccess)
                return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(cubic table)");
            wc->tab_cap = cap;
        }
        rc = upload_cu(ctx, tab, 0, tab.pos.size(), wc->d_pos, wc->d_fract);
        if (rc) return rc;
        wc->n_tiles = 0;
        if (wc->cfg.order != 2) {
            // filter and cubic stage are adjacent: they run as one launch, tile by tile
            std::vector<int> tile_n;
            st_tile_starts(tab, wc->cfg.order == 0 ? wc->out_len : s.cu_out, tile_n);
            if (tile_n.size() > wc->tile_cap) {
                (void)hipStreamSynchronize(ctx->stream);
                if (wc->d_tile_n) (void)hipFree(wc->d_tile_n);
                wc->d_tile_n = nullptr; wc->tile_cap = 0;
                const size_t cap = tile_n.size() + tile_n.size() / 8 + 64;
                if (hipMalloc((void**)&wc->d_tile_n, cap * sizeof(int)) != hipSuccess) return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(tile table)");
                wc->tile_cap = cap;
            }
            if (!tile_n.empty()) {
                hipError_t e = hipMemcpyAsync(wc->d_tile_n, tile_n.data(), tile_n.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(tile table)");
                wc->n_tiles = tile_n.size() - 1;
            }
        }
        wc->sr = sample_rate; wc->ch = ch; wc->rate = rate; wc->pitch = pitch; wc->in_len = in_len;
        wc->valid = true;
    }
    if (n_streams == 0 || wc->out_len == 0) return NAE_OK;
    const StCfg& cfg = wc->cfg;
    const StState& fin = wc->fin;
    // intermediate signals: interleaved [n][ch] per stream
    const size_t w = (size_t)ch;
    size_t len1, len2;     // outputs of the first and the second stage
    switch (cfg.order) {
    case 0: len1 = (size_t)fin.td_out; len2 = (size_t)fin.aa_out; break;
    case 1: len1 = (size_t)fin.aa_out; len2 = (size_t)fin.cu_out; break;
    default: len1 = (size_t)fin.cu_out; len2 = (size_t)fin.aa_out; break;
    }
    const size_t stride1 = (len1 + 8) * w, stride2 = (len2 + 8) * w;
    rc = nae_ws_reserve(ctx, &wc->ws_a, &wc->ws_a_bytes, n_streams * stride1 * sizeof(float));
    if (rc) return rc;
    rc = nae_ws_reserve(ctx, &wc->ws_b, &wc->ws_b_bytes, n_streams * stride2 * sizeof(float));
    if (rc) return rc;
    const StView v_src{(const float*)src->base, (long long)src->stream_stride, (long long)src->chan_stride,
                       (long long)src->frame_stride, 0, (long long)in_len};
    const StOut o_a{(float*)wc->ws_a, (long long)stride1, 1, (long long)w, 0};
    const StOut o_b{(float*)wc->ws_b, (long long)stride2, 1, (long long)w, 0};
    const StView v_a{(const float*)wc->ws_a, (long long)stride1, 1, (long long)w, 0, (long long)len1};
    const StView v_b{(const float*)wc->ws_b, (long long)stride2, 1, (long long)w, 0, (long long)len2};
    const StOut o_dst{(float*)dst->base, (long long)dst->stream_stride, (long long)dst->chan_stride, (long long)dst->frame_stride, 0};
    const TdRange all{0, 0, fin.td_nseq, 0.0, 1, cfg.order == 0 ? (long long)len1 : wc->out_len};
    const long long n_offs = fin.td_nseq > 0 ? fin.td_nseq - 1 : 0;
    switch (cfg.order) {
    case 0:
        rc = st_launch_td(ctx, cfg, v_src, all, o_a, n_streams, nullptr, offsets_dbg, n_offs);
        if (!rc && !ctx->dbg_st_unfused)
            rc = st_launch_aa_cu(ctx, cfg, v_a, wc->d_pos, wc->d_fract, wc->d_tile_n, wc->n_tiles, wc->out_len, o_dst, n_streams);
        else {
            if (!rc) rc = st_launch_aa(ctx, cfg, v_a, 0, fin.aa_out, o_b, n_streams);
            if (!rc) rc = st_launch_cu(ctx, cfg, v_b, wc->d_pos, wc->d_fract, 0, 0, wc->out_len, o_dst, n_streams);
        }
        break;
    case 1:
        if (!ctx->dbg_st_unfused)
            rc = st_launch_aa_cu(ctx, cfg, v_src, wc->d_pos, wc->d_fract, wc->d_tile_n, wc->n_tiles, fin.cu_out, o_b, n_streams);
        else {
            rc = st_launch_aa(ctx, cfg, v_src, 0, fin.aa_out, o_a, n_streams);
            if (!rc) rc = st_launch_cu(ctx, cfg, v_a, wc->d_pos, wc->d_fract, 0, 0, fin.cu_out, o_b, n_streams);
        }
        if (!rc) rc = st_launch_td(ctx, cfg, v_b, all, o_dst, n_streams, nullptr, offsets_dbg, n_offs);
        break;
    default:
        if (!ctx->dbg_st_unfused)
            rc = st_launch_cu_aa(ctx, cfg, v_src, wc->d_pos, wc->d_fract, fin.cu_out, fin.aa_out, o_b, n_streams);
        else {
            rc = st_launch_cu(ctx, cfg, v_src, wc->d_pos, wc->d_fract, 0, 0, fin.cu_out, o_a, n_streams);
            if (!rc) rc = st_launch_aa(ctx, cfg, v_a, 0, fin.aa_out, o_b, n_streams);
        }
        if (!rc) rc = st_launch_td(ctx, cfg, v_b, all, o_dst, n_streams, nullptr, offsets_dbg, n_offs);
        break;
    }
    return rc;
}


// ------------------------------------------------------------------ streaming handle
// SoundTouch-shaped calls (audio-velocity.cpp:403 putSamples, :399 numSamples, :298 receiveSamples, :427 flush) on
// one stream.  Every put advances the scalar bookkeeping, then each stage is launched once for the index range
// that became computable; the FIFOs between the stages are addressed by absolute frame index.
} // extern "C"

namespace {

struct AbsFifo {                 // frames [base, total) live at p[(i - base) * w]
    float* p = nullptr;
    size_t cap = 0;              // frames
    long long base = 0, total = 0;
    int w = 1;
};

int absfifo_reserve(nae_ctx* ctx, AbsFifo& f, long long want_total)
{
    if (want_total - f.base <= (long long)f.cap) return NAE_OK;
    size_t cap = f.cap ? f.cap : 8192;
    while ((long long)cap < want_total - f.base) cap *= 2;
    float* np = nullptr;
    if (hipMalloc((void**)&np, cap * (size_t)f.w * sizeof(float)) != hipSuccess) return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(stream FIFO)");
    const long long live = f.total - f.base;
    if (live > 0) {
        hipError_t e = hipMemcpyAsync(np, f.p, (size_t)live * (size_t)f.w * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(np); return nae_check(ctx, e, "hipMemcpyAsync(FIFO grow)"); }
    }
    if (f.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(f.p);
    }
    f.p = np;
    f.cap = cap;
    return NAE_OK;
}

// forget frames below new_base; the live part moves to the front once the dead prefix is at least as long (the two
// regions of the copy then do not overlap)
int absfifo_drop(nae_ctx* ctx, AbsFifo& f, long long new_base)
{
    if (new_base > f.total) new_base = f.total;
    if (new_base <= f.base) return NAE_OK;
    const long long dead = new_base - f.base, live = f.total - new_base;
    if (live == 0) {
        f.base = new_base;
        return NAE_OK;
    }
    if (dead < live) return NAE_OK;      // not yet worth it (and not safe in place)
    hipError_t e = hipMemcpyAsync(f.p, f.p + (size_t)dead * (size_t)f.w, (size_t)live * (size_t)f.w * sizeof(float),
                                  hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(FIFO compact)");
    f.base = new_base;
    return NAE_OK;
}

void absfifo_free(AbsFifo& f)
{
    if (f.p) (void)hipFree(f.p);
    f = AbsFifo{};
}

StView view_of(const AbsFifo& f, long long valid_end) { return StView{f.p, 0, 1, (long long)f.w, f.base, valid_end}; }
StOut out_of(const AbsFifo& f) { return StOut{f.p, 0, 1, (long long)f.w, f.base}; }

} // namespace

struct nae_wsola {
    nae_ctx* ctx = nullptr;
    StCfg cfg;
    StState st;                  // everything up to this state has been computed on the device
    long long* d_pos = nullptr;
    float* d_fract = nullptr;
    size_t tab_cap = 0;
    AbsFifo in, a, b, out;       // input, stage 1 -> 2, stage 2 -> 3, result
    float* d_mid = nullptr;      // stretcher tail carried between calls
    long long in_real = 0;       // frames really put (flush zeros are virtual)
    long long received = 0;
    long long out_limit = -1;    // total frames that may ever be handed out once flushed
    bool flushed = false;
};

namespace {

int wsola_run(nae_wsola* h, const StState& before)
{
    nae_ctx* ctx = h->ctx;
    const StCfg& c = h->cfg;
    const StState& now = h->st;
    int rc;
    // table entries for the new cubic outputs: generated on the device from the state in front of them (no host copy,
    // so a put never blocks)
    const long long cu_new = now.cu_out - before.cu_out;
    if (cu_new > 0) {
        if ((size_t)cu_new > h->tab_cap) {
            (void)hipStreamSynchronize(ctx->stream);     // growth only: the old table may still be in use
            if (h->d_pos) (void)hipFree(h->d_pos);
            if (h->d_fract) (void)hipFree(h->d_fract);
            h->d_pos = nullptr; h->d_fract = nullptr; h->tab_cap = 0;
            const size_t cap = (size_t)cu_new * 2 + 4096;
            if (hipMalloc((void**)&h->d_pos, cap * sizeof(long long)) != hipSuccess ||
                hipMalloc((void**)&h->d_fract, cap * sizeof(float)) != hipSuccess)
                return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(cubic table)");
            h->tab_cap = cap;
        }
        rc = st_launch_cu_table(ctx, before.cu_pos, before.cu_fract, c.rate, cu_new, h->d_pos, h->d_fract);
        if (rc) return rc;
    }
    AbsFifo* chain[4] = {&h->in, &h->a, &h->b, &h->out};
    const int kinds[3][3] = {{0, 1, 2}, {1, 2, 0}, {2, 1, 0}};   // 0 TD, 1 AA, 2 CU per stage slot
    long long in_valid[3];
    in_valid[0] = h->in_real;
    for (int slot = 0; slot < 3; slot++) {
        const int kind = kinds[c.order][slot];
        AbsFifo& src = *chain[slot];
        AbsFifo& dst = *chain[slot + 1];
        const long long out_before = kind == 0 ? before.td_out : (kind == 1 ? before.aa_out : before.cu_out);
        const long long out_now = kind == 0 ? now.td_out : (kind == 1 ? now.aa_out : now.cu_out);
        if (out_now > out_before) {
            rc = absfifo_reserve(ctx, dst, out_now);
            if (rc) return rc;
            const StView vin = view_of(src, in_valid[slot]);
            const StOut vout = out_of(dst);
            if (kind == 0) {
                const TdRange r{before.td_ip, before.td_out, now.td_nseq - before.td_nseq, before.td_skip, before.td_begin ? 1 : 0,
                                out_now};
                rc = st_launch_td(ctx, c, vin, r, vout, 1, h->d_mid, nullptr, 0);
            } else if (kind == 1) {
                rc = st_launch_aa(ctx, c, vin, out_before, out_now, vout, 1);
            } else {
                rc = st_launch_cu(ctx, c, vin, h->d_pos, h->d_fract, before.cu_out, out_before, out_now, vout, 1);
            }
            if (rc) return rc;
            dst.total = out_now;
        }
        if (slot < 2) in_valid[slot + 1] = dst.total;
        // what this stage will never read again
        const long long keep_from = kind == 0 ? now.td_ip : (kind == 1 ? now.aa_out : now.cu_pos);
        rc = absfifo_drop(ctx, src, keep_from);
        if (rc) return rc;
    }
    return NAE_OK;
}

} // namespace

extern "C" {

int nae_wsola_create(nae_ctx* ctx, int sample_rate, int channels, double rate, double pitch, nae_wsola** out)
{
    if (!ctx) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    if (!out) return nae_fail(ctx, NAE_ERR_INVALID, "nae_wsola_create: null handle pointer");
    nae_wsola* h = new (std::nothrow) nae_wsola();
    if (!h) return nae_fail(ctx, NAE_ERR_NOMEM, "nae_wsola_create");
    h->ctx = ctx;
    const int rc = st_cfg_make(h->cfg, sample_rate, channels, rate, pitch);
    if (rc) {
        delete h;
        return nae_fail(ctx, rc, rc == NAE_ERR_UNSUPPORTED ? "nae_wsola_create: sample rate outside 8000..48000 Hz or ratio out of range"
                                                           : "nae_wsola_create: bad parameters");
    }
    h->in.w = h->a.w = h->b.w = h->out.w = channels;
    if (hipMalloc((void**)&h->d_mid, (size_t)h->cfg.ovl * (size_t)channels * sizeof(float)) != hipSuccess) {
        delete h;
        return nae_fail(ctx, NAE_ERR_NOMEM, "hipMalloc(stretcher tail)");
    }
    *out = h;
    return NAE_OK;
}

static int wsola_append(nae_wsola* h, const float* p, size_t S, bool host)
{
    if (!h) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    nae_ctx* ctx = h->ctx;
    if (h->flushed) return nae_fail(ctx, NAE_ERR_STATE, "nae_wsola_put after flush");
    if (S == 0) return NAE_OK;
    if (!p) return nae_fail(ctx, NAE_ERR_INVALID, "nae_wsola_put: null samples");
    int rc = absfifo_reserve(ctx, h->in, h->in.total + (long long)S);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(h->in.p + (size_t)(h->in.total - h->in.base) * (size_t)h->in.w, p, S * (size_t)h->in.w * sizeof(float),
                                  host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(put)");
    if (host) {
        // the caller may reuse its buffer as soon as the call returns (with pinned memory the copy is truly asynchronous)
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipStreamSynchronize(put)");
    }
    h->in.total += (long long)S;
    h->in_real = h->in.total;
    const StState before = h->st;
    st_sim_put(h->cfg, h->st, (long long)S, nullptr);
    return wsola_run(h, before);
}

int nae_wsola_put(nae_wsola* h, const float* interleaved, size_t S) { return wsola_append(h, interleaved, S, false); }
int nae_wsola_put_host(nae_wsola* h, const float* interleaved, size_t S) { return wsola_append(h, interleaved, S, true); }

int nae_wsola_flush(nae_wsola* h)
{
    if (!h) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    if (h->flushed) return NAE_OK;
    const StState before = h->st;
    const long long avail = sim_flush(h->cfg, h->st, h->received, nullptr, nullptr);
    const int rc = wsola_run(h, before);
    if (rc) return rc;
    h->out_limit = h->received + avail;
    h->flushed = true;
    return NAE_OK;
}

size_t nae_wsola_available(const nae_wsola* h)
{
    if (!h) return 0;
    long long total = h->out.total;
    if (h->out_limit >= 0 && total > h->out_limit) total = h->out_limit;
    return total > h->received ? (size_t)(total - h->received) : 0;
}

static int wsola_take(nae_wsola* h, float* dst, size_t max_frames, size_t* got, bool host)
{
    if (!h) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    nae_ctx* ctx = h->ctx;
    size_t n = nae_wsola_available(h);
    if (n > max_frames) n = max_frames;
    if (got) *got = n;
    if (n == 0) return NAE_OK;
    if (!dst) return nae_fail(ctx, NAE_ERR_INVALID, "nae_wsola_receive: null destination");
    hipError_t e = hipMemcpyAsync(dst, h->out.p + (size_t)(h->received - h->out.base) * (size_t)h->out.w, n * (size_t)h->out.w * sizeof(float),
                                  host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(receive)");
    if (host) {
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipStreamSynchronize");
    }
    h->received += (long long)n;
    return absfifo_drop(ctx, h->out, h->received);
}

int nae_wsola_receive(nae_wsola* h, float* dst, size_t max_frames, size_t* got) { return wsola_take(h, dst, max_frames, got, false); }
int nae_wsola_receive_host(nae_wsola* h, float* dst, size_t max_frames, size_t* got) { return wsola_take(h, dst, max_frames, got, true); }

int nae_wsola_destroy(nae_wsola* h)
{
    if (!h) return NAE_OK;
    if (h->ctx) (void)nae_use_device(h->ctx);
    if (h->ctx && h->ctx->stream) (void)hipStreamSynchronize(h->ctx->stream);
    absfifo_free(h->in);
    absfifo_free(h->a);
    absfifo_free(h->b);
    absfifo_free(h->out);
    if (h->d_mid) (void)hipFree(h->d_mid);
    if (h->d_pos) (void)hipFree(h->d_pos);
    if (h->d_fract) (void)hipFree(h->d_fract);
    delete h;
    return NAE_OK;
}

} // extern "C"
