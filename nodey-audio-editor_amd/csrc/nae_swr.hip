// nae_swr.hip — N2: the input conversion every mixer input of the reference runs through libswresample for
// (/root/reference/src/processor/audio-amix.cpp:212-240,263-282; audio-bimix.cpp:198-240,259-294;
// include/utility/sw-resample.hpp:55-70): any supported sample format / mono|stereo / rate -> out_rate stereo f32.
//
// Rate conversion follows libswresample's DEFAULT resampler (the reference sets no resampler option): polyphase
// Kaiser-windowed sinc, filter_size 32, phase_shift 10 (at most 1024 phases; exact_rational: 160 for 44.1 -> 48 kHz), nearest phase, kaiser_beta 9, cutoff 0.97, the
// signal reflected about its first sample and behind its last one — specification in include/nae_dsp_spec.h and
// oracle/orc_swr.c.  UNPINNED versus FFmpeg (the library is absent; x86 builds sum the taps in SIMD order); GPU output is
// bit-identical to the oracle, the oracle within 1e-7 of the float64 golden.
// Same rate: a wire (the only case the reference's own tests could pin: FLT -> FLTP is a bit copy).
//
// Kernel: 256 outputs per workgroup; the input span they touch is staged once in LDS as (L, R) pairs with the reflection
// already applied, every thread then walks its 32..512 taps — LDS for the samples, the 128 KiB filter bank through L1/L2
// (a tile touches few of the 1024 phases: 44.1 -> 48 kHz cycles through 160).  Taps are summed as the library's C
// template does: even and odd taps in two float accumulators, in increasing order, no fused multiply-add.
#include "stream_util.h"
#include <math.h>
#include <new>
#include <string.h>

namespace {

struct SwrPlan {
    int in_rate, out_rate;
    int L, alloc, P;
    int src_incr, div, mod;
    long long index0;
    double factor;
};

long long gcd_ll(long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; }

double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

int swr_plan_make(int in_rate, int out_rate, SwrPlan* p)
{
    memset(p, 0, sizeof *p);
    p->in_rate = in_rate;
    p->out_rate = out_rate;
    p->P = 1 << NAE_SWR_PHASE_SHIFT;
    {
        // exact_rational (the library's default): a ratio whose reduced numerator fits the phase table uses exactly that many phases
        const long long exact = (long long)out_rate / gcd_ll(out_rate, in_rate);
        if (exact <= p->P) p->P = (int)exact;
    }
    p->factor = (double)out_rate * NAE_SWR_CUTOFF / (double)in_rate;
    if (p->factor > 1.0) p->factor = 1.0;
    p->L = (int)ceil(NAE_SWR_FILTER_SIZE / p->factor);
    if (p->L > 1) p->L = (p->L + 1) & ~1;          // the library rounds a multi-tap filter up to an even length
    if (p->L < 1) p->L = 1;
    if (p->L > NAE_SWR_MAX_TAPS) return NAE_ERR_UNSUPPORTED;
    p->alloc = (p->L + 7) & ~7;
    const long long num = out_rate, den = (long long)in_rate * p->P, g = gcd_ll(num, den);
    if (num / g > 0x3fffffff || den / g > 0x3fffffff) return NAE_ERR_UNSUPPORTED;
    p->src_incr = (int)(num / g);
    const long long dst_incr = den / g;
    p->div = (int)(dst_incr / p->src_incr);
    p->mod = (int)(dst_incr % p->src_incr);
    p->index0 = -(long long)p->P * ((p->L - 1) / 2);
    return NAE_OK;
}

void swr_build_filter(const SwrPlan& p, std::vector<float>& bank)
{
    const int L = p.L, center = (L - 1) / 2;
    const double pi = 3.14159265358979323846;
    std::vector<double> tab(L);
    bank.assign((size_t)p.P * p.alloc, 0.0f);
    for (int ph = 0; ph < p.P; ph++) {
        double norm = 0.0;
        for (int i = 0; i < L; i++) {
            const double x = pi * ((double)(i - center) - (double)ph / p.P) * p.factor;
            double y = (x == 0.0) ? 1.0 : sin(x) / x;
            const double w = 2.0 * x / (p.factor * L * pi);
            const double a = 1.0 - w * w;
            y *= bessel_i0(NAE_SWR_KAISER_BETA * sqrt(a > 0.0 ? a : 0.0));
            tab[i] = y;
            norm += y;
        }
        for (int i = 0; i < L; i++) bank[(size_t)ph * p.alloc + i] = (float)(tab[i] / norm);
    }
}

// position of output n: first input sample of its window and filter phase; `frac` = (n * mod) mod src_incr
void swr_position(const SwrPlan& p, uint64_t n, long long* s, int* ph, long long* pos_out = nullptr, int* frac = nullptr)
{
    const unsigned __int128 nm = (unsigned __int128)n * (unsigned)p.mod;
    const long long pos = p.index0 + (long long)n * p.div + (long long)(nm / (unsigned)p.src_incr);
    long long q = pos / p.P;
    if (pos % p.P < 0) q--;
    *s = q;
    *ph = (int)(pos - q * p.P);
    if (pos_out) *pos_out = pos;
    if (frac) *frac = (int)(nm % (unsigned)p.src_incr);
}

size_t swr_reflection(const SwrPlan& p, size_t n_in) { return ((n_in < (size_t)p.L ? n_in : (size_t)p.L) + 1) / 2; }

size_t swr_outputs_upto(const SwrPlan& p, size_t n_avail)
{
    if (n_avail == 0) return 0;
    uint64_t lo = 0, hi = (uint64_t)((double)n_avail * p.out_rate / p.in_rate) + 4 * (uint64_t)p.L + 16;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        long long s; int ph;
        swr_position(p, mid, &s, &ph);
        if (s + p.L > (long long)n_avail) hi = mid; else lo = mid + 1;
    }
    return (size_t)lo;
}

struct SwrKernelArgs {
    const float* in;          // interleaved stereo frames [in_base, n_in)
    long long in_base, n_in;
    int refl;                 // frames the flush reflects behind the end (0 before the flush)
    int L, alloc, P, src_incr, div, mod;
    long long pos0;           // position (1/P samples) and fractional state of output n0
    int frac0;
    const float* bank;
    int count;                // outputs of this launch: n0 .. n0 + count - 1
    float* out;               // interleaved stereo
    int span_cap;             // frames of LDS staging
};

constexpr int kSwrTile = 256;

__device__ __forceinline__ void swr_pos(const SwrKernelArgs& a, int j, long long* s, int* ph)
{
    const long long pos = a.pos0 + (long long)j * a.div + ((long long)a.frac0 + (long long)j * a.mod) / a.src_incr;
    long long q = pos / a.P;
    if (pos % a.P < 0) q--;
    *s = q;
    *ph = (int)(pos - q * a.P);
}

__global__ __launch_bounds__(kSwrTile) void swr_resample_kernel(SwrKernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char swr_smem[];
    float2* stage = reinterpret_cast<float2*>(swr_smem);
    const int j0 = blockIdx.x * kSwrTile;
    const int j1 = (j0 + kSwrTile < a.count) ? j0 + kSwrTile : a.count;
    long long s_first, s_last;
    int ph;
    swr_pos(a, j0, &s_first, &ph);
    swr_pos(a, j1 - 1, &s_last, &ph);
    const int span = (int)(s_last + a.L - s_first);
    const float2* in2 = reinterpret_cast<const float2*>(a.in);
    for (int k = threadIdx.x; k < span; k += kSwrTile) {
        long long idx = s_first + k;
        float2 v{0.0f, 0.0f};
        if (idx < 0) idx = -idx;                                   // reflection about the first sample
        else if (idx >= a.n_in) {
            const long long back = idx - a.n_in;                   // what the flush appends: x[N + j] = x[N - 1 - j]
            idx = (back < a.refl) ? a.n_in - 1 - back : -1;
        }
        if (idx >= a.in_base && idx < a.n_in) v = in2[idx - a.in_base];
        stage[k] = v;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= j1) return;
    long long s;
    swr_pos(a, j, &s, &ph);
    const float* f = a.bank + (size_t)ph * a.alloc;
    const float2* x = stage + (int)(s - s_first);
    float vl = 0.0f, vl2 = 0.0f, vr = 0.0f, vr2 = 0.0f;
    int i = 0;
    for (; i + 1 < a.L; i += 2) {
        const float2 x0 = x[i], x1 = x[i + 1];
        const float f0 = f[i], f1 = f[i + 1];
        vl += x0.x * f0; vr += x0.y * f0;
        vl2 += x1.x * f1; vr2 += x1.y * f1;
    }
    if (i < a.L) { const float2 x0 = x[i]; vl += x0.x * f[i]; vr += x0.y * f[i]; }
    reinterpret_cast<float2*>(a.out)[j] = float2{vl + vl2, vr + vr2};
}

} // namespace

struct nae_swr {
    nae_ctx* ctx;
    int in_fmt, in_rate, in_ch, out_rate;
    bool identity;
    SwrPlan pl{};
    float* d_bank = nullptr;
    DevFifo in;                // interleaved stereo f32 frames [in.base, n_in)
    size_t n_in = 0;
    DevFifo out;               // interleaved stereo f32 frames [out.base, n_out)
    size_t n_out = 0, n_read = 0;
    DevBuf raw, f32, planes;   // staging of one convert call
    bool flushed = false;
};

static int swr_run(nae_swr* h)
{
    nae_ctx* ctx = h->ctx;
    const SwrPlan& p = h->pl;
    const size_t refl = h->flushed ? swr_reflection(p, h->n_in) : 0;
    const size_t n_avail = h->n_in ? swr_outputs_upto(p, h->n_in + refl) : 0;
    if (n_avail <= h->n_out) return NAE_OK;
    const size_t count = n_avail - h->n_out;
    if (count > 0x3fffffff) return nae_fail(ctx, NAE_ERR_INVALID, "too many output frames in one call");
    int rc = fifo_reserve_interleaved(ctx, h->out, h->n_out, n_avail, 2);
    if (rc) return rc;
    SwrKernelArgs a;
    a.in = h->in.cur.p; a.in_base = (long long)h->in.base; a.n_in = (long long)h->n_in; a.refl = (int)refl;
    a.L = p.L; a.alloc = p.alloc; a.P = p.P; a.src_incr = p.src_incr; a.div = p.div; a.mod = p.mod;
    long long s; int ph;
    swr_position(p, h->n_out, &s, &ph, &a.pos0, &a.frac0);
    a.bank = h->d_bank;
    a.count = (int)count;
    a.out = h->out.cur.p + (h->n_out - h->out.base) * 2;
    a.span_cap = (int)((double)kSwrTile * p.in_rate / p.out_rate) + p.L + 4;
    const unsigned grid = (unsigned)((count + kSwrTile - 1) / kSwrTile);
    NAE_KLAUNCH(ctx, "swr_resample_kernel", swr_resample_kernel, dim3(grid), dim3(kSwrTile), (size_t)a.span_cap * sizeof(float2), ctx->stream, a);
    rc = nae_check(ctx, hipGetLastError(), "swr_resample_kernel");
    if (rc) return rc;
    h->n_out = n_avail;
    h->out.cur.len = (h->n_out - h->out.base) * 2;
    // input in front of the next window is no longer needed — except the first L frames while a window can still start
    // in front of sample 0 (reflection), and never the tail the flush reflects
    if (!h->flushed) {
        swr_position(p, h->n_out, &s, &ph);
        if (s > (long long)h->in.base + (1 << 16)) return fifo_drop_interleaved(ctx, h->in, (size_t)s, h->n_in, 2);
    }
    return NAE_OK;
}

extern "C" {

int nae_swr_create(nae_ctx* ctx, int in_fmt, int in_rate, int in_channels, int out_rate, nae_swr** h)
{
    if (!ctx || !h) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    *h = nullptr;
    switch (in_fmt) {
    case NAE_FMT_S16: case NAE_FMT_S32: case NAE_FMT_FLT: case NAE_FMT_S16P: case NAE_FMT_S32P: case NAE_FMT_FLTP: break;
    default: return nae_fail(ctx, NAE_ERR_UNSUPPORTED, "Unsupported sample format");
    }
    if (in_channels != 1 && in_channels != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    if (in_rate <= 0 || out_rate <= 0) return nae_fail(ctx, NAE_ERR_INVALID, "sample rates must be positive");
    nae_swr* s = new (std::nothrow) nae_swr();
    if (!s) return NAE_ERR_NOMEM;
    s->ctx = ctx; s->in_fmt = in_fmt; s->in_rate = in_rate; s->in_ch = in_channels; s->out_rate = out_rate;
    s->identity = in_rate == out_rate;
    if (!s->identity) {
        int rc = swr_plan_make(in_rate, out_rate, &s->pl);
        if (rc) { delete s; return nae_fail(ctx, rc, "sample-rate ratio outside the supported range (at most ~15x down, reducible to 30-bit increments)"); }
        std::vector<float> bank;
        swr_build_filter(s->pl, bank);
        hipError_t e = hipMalloc((void**)&s->d_bank, bank.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(s->d_bank, bank.data(), bank.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (s->d_bank) (void)hipFree(s->d_bank); delete s; return nae_check(ctx, e, "filter bank upload"); }
    }
    *h = s;
    return NAE_OK;
}

size_t nae_swr_buffered(nae_swr* h) { return h ? h->n_out - h->n_read : 0; }

/* upload + format conversion of one call's input, appended to the input FIFO (or, equal rates, straight to the output
 * FIFO: a wire); queued on the stream, nothing waited for */
static int swr_feed(nae_swr* h, const void* const* planes, size_t n_in)
{
    nae_ctx* ctx = h->ctx;
    int rc;
    if (h->flushed) return nae_fail(ctx, NAE_ERR_STATE, "input after drain");
    const bool planar = (h->in_fmt == NAE_FMT_FLTP || h->in_fmt == NAE_FMT_S16P || h->in_fmt == NAE_FMT_S32P);
    const int bps = (h->in_fmt == NAE_FMT_S16 || h->in_fmt == NAE_FMT_S16P) ? 2 : 4;
    const int n_planes = planar ? h->in_ch : 1;
    const size_t plane_bytes = n_in * bps * (planar ? 1 : h->in_ch);
    const size_t stride = (plane_bytes + 255) / 256 * 256;
    if ((rc = devbuf_reserve(ctx, h->raw, stride * n_planes / sizeof(float) + 64))) return rc;
    unsigned char* raw = reinterpret_cast<unsigned char*>(h->raw.p);
    const void* dp[2] = {raw, raw + stride};
    for (int p = 0; p < n_planes; p++) {
        if (!planes[p]) return nae_fail(ctx, NAE_ERR_INVALID, "null plane pointer");
        hipError_t e = hipMemcpyAsync(raw + p * stride, planes[p], plane_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpyAsync(swr in)");
    }
    // format -> f32 (K6 scaling), mono -> stereo (L = R = m / sqrt(2), swr's default float rematrix)
    DevFifo& dstf = h->identity ? h->out : h->in;
    size_t& total = h->identity ? h->n_out : h->n_in;
    if ((rc = fifo_reserve_interleaved(ctx, dstf, total, total + n_in, 2))) return rc;
    float* tail = dstf.cur.p + (total - dstf.base) * 2;
    if (h->in_ch == 2) {
        if ((rc = nae_to_f32_interleaved(ctx, h->in_fmt, dp, n_in, 2, tail))) return rc;
    } else {
        if ((rc = devbuf_reserve(ctx, h->f32, n_in))) return rc;
        if ((rc = nae_to_f32_interleaved(ctx, h->in_fmt, dp, n_in, 1, h->f32.p))) return rc;
        if ((rc = nae_mono_to_stereo_f32(ctx, h->f32.p, tail, n_in, 0.70710678118654752440f))) return rc;
    }
    total += n_in;
    dstf.cur.len = (total - dstf.base) * 2;
    return NAE_OK;
}

/* the next min(buffered, max_out) output frames as two device planes; queued on the stream */
static int swr_emit(nae_swr* h, float* dL, float* dR, size_t max_out, size_t* n_out)
{
    nae_ctx* ctx = h->ctx;
    size_t n = h->n_out - h->n_read;
    if (n > max_out) n = max_out;
    *n_out = n;
    if (n == 0) return NAE_OK;
    float* planes_out[2] = {dL, dR};
    int rc = nae_deinterleave_f32(ctx, h->out.cur.p + (h->n_read - h->out.base) * 2, planes_out, n, 2);
    if (rc) return rc;
    h->n_read += n;
    if (h->n_read - h->out.base > (1u << 16)) return fifo_drop_interleaved(ctx, h->out, h->n_read, h->n_out, 2);
    return NAE_OK;
}

static int swr_convert_common(nae_swr* h, const void* const* planes, size_t n_in)
{
    int rc;
    if (planes && n_in) {
        if ((rc = swr_feed(h, planes, n_in))) return rc;
    } else if (!planes) {
        h->flushed = true;
    }
    return h->identity ? NAE_OK : swr_run(h);
}

/* swr_convert(ctx, out, max_out, in, n_in): consumes all n_in frames, delivers at most max_out, keeps the rest; planes == NULL
 * drains (audio-amix.cpp:281-282) */
int nae_swr_convert_host(nae_swr* h, const void* const* planes, size_t n_in, float* outL, float* outR, size_t max_out, size_t* n_out)
{
    if (!h || !n_out || (max_out && (!outL || !outR))) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    nae_ctx* ctx = h->ctx;
    *n_out = 0;
    int rc;
    if ((rc = swr_convert_common(h, planes, n_in))) return rc;
    size_t n = h->n_out - h->n_read;
    if (n > max_out) n = max_out;
    if (n == 0) {
        // the caller may reuse its planes
        return nae_check(ctx, hipStreamSynchronize(ctx->stream), "swr in");
    }
    if ((rc = devbuf_reserve(ctx, h->planes, 2 * n))) return rc;
    float* dL = h->planes.p;
    float* dR = h->planes.p + n;
    if ((rc = swr_emit(h, dL, dR, n, &n))) return rc;
    hipError_t e = hipMemcpyAsync(outL, dL, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(outR, dR, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "swr out");
    *n_out = n;
    return NAE_OK;
}

int nae_swr_convert(nae_swr* h, const void* const* planes, size_t n_in, float* outL, float* outR, size_t max_out, size_t* n_out)
{
    if (!h || !n_out || (max_out && (!outL || !outR))) return NAE_ERR_INVALID;
    (void)nae_use_device(h->ctx);
    *n_out = 0;
    int rc;
    if ((rc = swr_convert_common(h, planes, n_in))) return rc;
    return swr_emit(h, outL, outR, max_out, n_out);
}

int nae_swr_destroy(nae_swr* h)
{
    if (!h) return NAE_OK;
    (void)nae_use_device(h->ctx);
    (void)hipStreamSynchronize(h->ctx->stream);
    fifo_free(h->in);
    fifo_free(h->out);
    devbuf_free(h->raw); devbuf_free(h->f32); devbuf_free(h->planes);
    if (h->d_bank) (void)hipFree(h->d_bank);
    delete h;
    return NAE_OK;
}

} // extern "C"
