/* orc_swr.c — CPU oracle of the N2 input resampler: TEST INFRASTRUCTURE ONLY.
 *
 * What it restates: the rate conversion libswresample performs for every mixer input of the reference
 * (/root/reference/src/processor/audio-amix.cpp:212-240,263-282, audio-bimix.cpp:198-240,259-294,
 * include/utility/sw-resample.hpp:55-70) with the library's DEFAULT resampler options — the reference sets none:
 * polyphase Kaiser-windowed sinc, filter_size 32, phase_shift 10, kaiser_beta 9, cutoff 0.97, exact_rational on,
 * nearest phase.
 * libswresample (FFmpeg 7.1, xmake.lua:12) is a third-party dependency that is neither under /root/reference nor
 * installed here; the algorithm below follows the published structure of its resample.c / resample_template.c
 * (build_filter, swri_resample's index / frac stepping, the reflected head and tail of invert_initial_buffer and
 * resample_flush) from public knowledge.  PARITY UNPINNED: no fixture of the reference covers it, and x86 builds of
 * the library sum the taps in SIMD order.  Pinned instead: tests/golden/swr_golden.npz (float64 numpy restatement of
 * the same specification, tests/golden/swr_numpy.py).
 *
 * Specification (one channel; channels are independent):
 *   factor = min(out_rate * 0.97 / in_rate, 1);  L = max(ceil(32 / factor), 1);
 *   P = out_rate / gcd(out_rate, in_rate) if that is <= 1024 (the library's exact_rational option, on by default), else 1024
 *   bank[ph][i] = f32( y / sum_i y ),  y = sinc(x) * I0(9 sqrt(max(1 - w^2, 0))),
 *                 x = pi ((i - (L-1)/2) - ph/P) factor,  w = 2 x / (factor L pi)           (double arithmetic)
 *   out_rate / (in_rate P) reduced by their gcd gives src_incr / dst_incr; position of output n in 1/P samples:
 *   pos(n) = -P ((L-1)/2) + n (dst_incr div src_incr) + floor(n (dst_incr mod src_incr) / src_incr)
 *   s = floor(pos / P), ph = pos - s P;  out[n] = (sum over even i of x[s+i] bank[ph][i]) + (sum over odd i ...), f32,
 *   taps in increasing order (the C template's two accumulators)
 *   x[k] for k < 0 is x[-k] (reflection about the first sample); behind the end x[N + j] = x[N - 1 - j] for
 *   j < R = (min(N, L) + 1) / 2 (what a flush appends); outputs exist while s + L <= N + R. */
#include "nae_oracle.h"
#include "../include/nae_dsp_spec.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static long long gcd_ll(long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; }

static double bessel_i0(double x)
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

int orc_swr_plan_make(int in_rate, int out_rate, orc_swr_plan* p)
{
    if (!p || in_rate <= 0 || out_rate <= 0) return -1;
    memset(p, 0, sizeof *p);
    p->in_rate = in_rate;
    p->out_rate = out_rate;
    p->phase_count = 1 << NAE_SWR_PHASE_SHIFT;
    {
        /* exact_rational (on by default): when out_rate / in_rate in lowest terms has a numerator <= 2^phase_shift, that
         * numerator is the phase count — every output then falls exactly on a phase (44.1 -> 48 kHz: 160 phases, 147 / 160 of a
         * sample per output; 96 -> 48 kHz: one phase).  Other ratios keep 2^phase_shift phases and the nearest-phase rule. */
        const long long exact = (long long)out_rate / gcd_ll(out_rate, in_rate);
        if (exact <= p->phase_count) p->phase_count = (int)exact;
    }
    p->factor = (double)out_rate * NAE_SWR_CUTOFF / (double)in_rate;
    if (p->factor > 1.0) p->factor = 1.0;
    p->filter_length = (int)ceil(NAE_SWR_FILTER_SIZE / p->factor);
    if (p->filter_length < 1) p->filter_length = 1;
    /* resample_init rounds a filter longer than one tap up to an EVEN length (FFALIGN(filter_length, 2)) before it builds the
     * bank: 88.2 -> 48 kHz gives 62 taps, not 61 (restated from public knowledge of FFmpeg 7.1's resample.c; UNPINNED) */
    if (p->filter_length > 1) p->filter_length = (p->filter_length + 1) & ~1;
    if (p->filter_length > NAE_SWR_MAX_TAPS) return -2;
    p->filter_alloc = (p->filter_length + 7) & ~7;
    const long long num = out_rate, den = (long long)in_rate * p->phase_count;
    const long long g = gcd_ll(num, den);
    if (num / g > 0x3fffffff || den / g > 0x3fffffff) return -2;      /* av_reduce's bound INT32_MAX / 2 */
    p->src_incr = (int)(num / g);
    const long long dst_incr = den / g;
    p->dst_incr_div = (int)(dst_incr / p->src_incr);
    p->dst_incr_mod = (int)(dst_incr % p->src_incr);
    p->index0 = -(long long)p->phase_count * ((p->filter_length - 1) / 2);
    return 0;
}

void orc_swr_build_filter(const orc_swr_plan* p, float* bank)
{
    const int L = p->filter_length, center = (L - 1) / 2;
    const double pi = 3.14159265358979323846;
    double* tab = (double*)malloc(sizeof(double) * (size_t)L);
    memset(bank, 0, sizeof(float) * (size_t)p->phase_count * (size_t)p->filter_alloc);
    for (int ph = 0; ph < p->phase_count; ph++) {
        double norm = 0.0;
        for (int i = 0; i < L; i++) {
            const double x = pi * ((double)(i - center) - (double)ph / p->phase_count) * p->factor;
            double y = (x == 0.0) ? 1.0 : sin(x) / x;
            const double w = 2.0 * x / (p->factor * L * pi);
            const double a = 1.0 - w * w;
            y *= bessel_i0(NAE_SWR_KAISER_BETA * sqrt(a > 0.0 ? a : 0.0));
            tab[i] = y;
            norm += y;
        }
        for (int i = 0; i < L; i++) bank[(size_t)ph * p->filter_alloc + i] = (float)(tab[i] / norm);
    }
    free(tab);
}

void orc_swr_position(const orc_swr_plan* p, uint64_t n, long long* s, int* ph)
{
    const unsigned __int128 carry = (unsigned __int128)n * (unsigned)p->dst_incr_mod / (unsigned)p->src_incr;
    const long long pos = p->index0 + (long long)n * p->dst_incr_div + (long long)carry;
    long long q = pos / p->phase_count;
    if (pos % p->phase_count < 0) q--;
    *s = q;
    *ph = (int)(pos - q * p->phase_count);
}

size_t orc_swr_reflection(const orc_swr_plan* p, size_t n_in)
{
    const size_t m = n_in < (size_t)p->filter_length ? n_in : (size_t)p->filter_length;
    return (m + 1) / 2;
}

/* outputs whose window [s, s + L) ends inside n_avail input samples (n_avail = N + R after the flush) */
size_t orc_swr_outputs_upto(const orc_swr_plan* p, size_t n_avail)
{
    if (n_avail == 0) return 0;
    /* pos(n) is non-decreasing: binary search for the first n with s(n) + L > n_avail */
    uint64_t lo = 0, hi = (uint64_t)((double)n_avail * p->out_rate / p->in_rate) + 4 * (uint64_t)p->filter_length + 16;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        long long s; int ph;
        orc_swr_position(p, mid, &s, &ph);
        if (s + p->filter_length > (long long)n_avail) hi = mid; else lo = mid + 1;
    }
    return (size_t)lo;
}

size_t orc_swr_out_len(const orc_swr_plan* p, size_t n_in)
{
    return n_in ? orc_swr_outputs_upto(p, n_in + orc_swr_reflection(p, n_in)) : 0;
}

static float tap(const float* x, size_t n, size_t stride, size_t refl, long long k)
{
    if (k < 0) k = -k;                                         /* reflection about sample 0 */
    else if (k >= (long long)n) {
        const long long j = k - (long long)n;
        if (j >= (long long)refl) return 0.0f;
        k = (long long)n - 1 - j;                              /* what the flush appended */
    }
    if (k < 0 || k >= (long long)n) return 0.0f;               /* (streams shorter than the filter) */
    return x[(size_t)k * stride];
}

/* one channel, whole signal (everything a drain delivers); returns the number of outputs written */
size_t orc_swr_resample_f32(const orc_swr_plan* p, const float* bank, const float* x, size_t n_in, size_t stride, float* out,
                            size_t out_stride)
{
    const size_t n_out = orc_swr_out_len(p, n_in), refl = orc_swr_reflection(p, n_in);
    const int L = p->filter_length;
    for (size_t n = 0; n < n_out; n++) {
        long long s; int ph;
        orc_swr_position(p, n, &s, &ph);
        const float* f = bank + (size_t)ph * p->filter_alloc;
        float val = 0.0f, val2 = 0.0f;
        int i = 0;
        for (; i + 1 < L; i += 2) {
            val += tap(x, n_in, stride, refl, s + i) * f[i];
            val2 += tap(x, n_in, stride, refl, s + i + 1) * f[i + 1];
        }
        if (i < L) val += tap(x, n_in, stride, refl, s + i) * f[i];
        out[n * out_stride] = val + val2;
    }
    return n_out;
}
