/* orc_stft.c — CPU ORACLE (test infrastructure only; see nae_oracle.h).
 *
 * Scalar, sequential implementation of the builder-specified nodes:
 *   K8  FFT spectrum  (no reference code: FFTW is declared at /root/reference/xmake.lua:15,33 and never
 *       called) — pinned against scipy.fft.rfft in float64 (tests/golden).
 *   K7  tempo/pitch   (reference = SoundTouch 2.3.2 behind /root/reference/src/processor/
 *       audio-velocity.cpp:369-428, library absent)  — PARITY UNPINNED vs the reference; this file is the
 *       CPU statement of the phase-vocoder + rate-transposer specified in DESIGN.md §3.
 *
 * The phase path (window -> FFT -> atan2 -> Q0.32 phase) is specified down to the order of every f32
 * operation so that the HIP kernels reproduce the integer phases bit for bit; everything behind the integer
 * phase accumulator is compared under the 1e-4 RMS tolerance.
 *
 * Compile with -ffp-contract=off (fused multiply-adds appear only where fmaf() is written).
 */
#include "nae_oracle.h"
#include "../include/nae_dsp_spec.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } cf;

static cf W512[512];       /* W512[k]  = exp(-2 pi i k / 512)        */
static cf T1024[513];      /* T1024[k] = exp(-2 pi i k / 1024)       */
static float HANN[1024];   /* periodic Hann, 0.5 - 0.5 cos(2 pi n/N) */
static int tables_ready = 0;

static void make_tables(void)
{
    if (tables_ready) return;
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < 512; k++) {
        W512[k].x = (float)cos(two_pi * k / 512.0);
        W512[k].y = (float)(-sin(two_pi * k / 512.0));
    }
    for (int k = 0; k <= 512; k++) {
        T1024[k].x = (float)cos(two_pi * k / 1024.0);
        T1024[k].y = (float)(-sin(two_pi * k / 1024.0));
    }
    for (int n = 0; n < 1024; n++) HANN[n] = (float)(0.5 - 0.5 * cos(two_pi * n / 1024.0));
    tables_ready = 1;
}

const float* orc_hann1024(void)
{
    make_tables();
    return HANN;
}

/* canonical twiddle multiply: one rounded product + one fused multiply-add per component */
static inline cf cmul_tw(cf v, cf w)
{
    cf r;
    r.x = fmaf(-v.y, w.y, v.x * w.x);
    r.y = fmaf(v.y, w.x, v.x * w.y);
    return r;
}
static inline cf cadd(cf a, cf b) { cf r = {a.x + b.x, a.y + b.y}; return r; }
static inline cf csub(cf a, cf b) { cf r = {a.x - b.x, a.y - b.y}; return r; }
static inline cf mul_mi(cf a) { cf r = {a.y, -a.x}; return r; } /* a * (-i) */

/* canonical forward 8-point DFT, natural-order output: three radix-2 DIF layers */
static void dft8_fwd(const cf a[8], cf b[8])
{
    const float c = NAE_SQRT1_2;
    cf s0 = cadd(a[0], a[4]), d0 = csub(a[0], a[4]);
    cf s1 = cadd(a[1], a[5]), e1 = csub(a[1], a[5]);
    cf s2 = cadd(a[2], a[6]), e2 = csub(a[2], a[6]);
    cf s3 = cadd(a[3], a[7]), e3 = csub(a[3], a[7]);
    cf d1 = {(e1.x + e1.y) * c, (e1.y - e1.x) * c};   /* * W8^1 = (1-i)/sqrt2  */
    cf d2 = mul_mi(e2);                               /* * W8^2 = -i           */
    cf d3 = {(e3.y - e3.x) * c, -((e3.x + e3.y) * c)};/* * W8^3 = (-1-i)/sqrt2 */
    cf t0 = cadd(s0, s2), t1 = csub(s0, s2), t2 = cadd(s1, s3), t3 = mul_mi(csub(s1, s3));
    b[0] = cadd(t0, t2); b[4] = csub(t0, t2); b[2] = cadd(t1, t3); b[6] = csub(t1, t3);
    cf u0 = cadd(d0, d2), u1 = csub(d0, d2), u2 = cadd(d1, d3), u3 = mul_mi(csub(d1, d3));
    b[1] = cadd(u0, u2); b[5] = csub(u0, u2); b[3] = cadd(u1, u3); b[7] = csub(u1, u3);
}

/* canonical forward 512-point FFT: DIF, 512 = 8 x 8 x 8 */
static void fft512_fwd(const cf* z, cf* Z)
{
    cf u1[8][64], u2[8][8][8];
    cf a[8], b[8];
    for (int l = 0; l < 64; l++) {                 /* pass A: stride-64 butterflies, twiddle W512^(l q) */
        for (int j = 0; j < 8; j++) a[j] = z[l + 64 * j];
        dft8_fwd(a, b);
        u1[0][l] = b[0];
        for (int q = 1; q < 8; q++) u1[q][l] = cmul_tw(b[q], W512[l * q]);
    }
    for (int q = 0; q < 8; q++)                    /* pass B: stride-8 butterflies, twiddle W64^(m p)   */
        for (int m = 0; m < 8; m++) {
            for (int j = 0; j < 8; j++) a[j] = u1[q][m + 8 * j];
            dft8_fwd(a, b);
            u2[q][0][m] = b[0];
            for (int p = 1; p < 8; p++) u2[q][p][m] = cmul_tw(b[p], W512[8 * m * p]);
        }
    for (int q = 0; q < 8; q++)                    /* pass C: final 8-point DFTs                        */
        for (int p = 0; p < 8; p++) {
            for (int j = 0; j < 8; j++) a[j] = u2[q][p][j];
            dft8_fwd(a, b);
            for (int r = 0; r < 8; r++) Z[q + 8 * p + 64 * r] = b[r];
        }
}

void orc_fft512_fwd(const float* zin, float* zout)
{
    make_tables();
    fft512_fwd((const cf*)zin, (cf*)zout);
}

/* canonical r2c: pack pairs, 512-point FFT, split */
static void rfft1024(const float* xw, cf* X)
{
    cf Z[512];
    fft512_fwd((const cf*)xw, Z);
    for (int k = 0; k <= 512; k++) {
        const cf A = Z[k & 511], B = Z[(512 - k) & 511];
        const cf E = {0.5f * (A.x + B.x), 0.5f * (A.y - B.y)};
        const cf O = {0.5f * (A.x - B.x), 0.5f * (A.y + B.y)};
        const cf P = cmul_tw(O, T1024[k]);
        X[k].x = E.x + P.y;
        X[k].y = E.y - P.x;
    }
}

void orc_rfft1024(const float* xw, float* X)
{
    make_tables();
    rfft1024(xw, (cf*)X);
}

/* c2r, 1/N normalised; tolerance-compared, so any correct evaluation order will do */
static void irfft1024(const cf* X, float* y)
{
    cf Z[512], Zc[512], z[512];
    for (int k = 0; k < 512; k++) {
        cf Xk = X[k], Xm = X[512 - k];
        if (k == 0) { Xk.y = 0.0f; Xm.y = 0.0f; }
        const cf E = {0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y)};
        const cf D = {0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y)};
        const cf T = T1024[k];
        const cf Q = {T.x * D.x + T.y * D.y, T.x * D.y - T.y * D.x}; /* conj(T) * D */
        Z[k].x = E.x - Q.y;
        Z[k].y = E.y + Q.x;
    }
    for (int k = 0; k < 512; k++) { Zc[k].x = Z[k].x; Zc[k].y = -Z[k].y; }
    fft512_fwd(Zc, z);
    for (int m = 0; m < 512; m++) {
        y[2 * m] = z[m].x * (1.0f / 512.0f);
        y[2 * m + 1] = -z[m].y * (1.0f / 512.0f);
    }
}

void orc_irfft1024(const float* X, float* y)
{
    make_tables();
    irfft1024((const cf*)X, y);
}

/* canonical atan2 in turns, quantised to Q0.32 (wrapping): include/nae_dsp_spec.h, revision 2 — reciprocal by an integer
 * seed and three Newton steps (fma only), octants by integer reflections that follow the sign bits */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int32_t orc_atan2_q32(float im, float re)
{
    const float ax = fabsf(re), ay = fabsf(im);
    if (!((ax + ay) < NAE_ATAN_HUGE)) return 0;           /* a bin of 2^100 or more — NaN, Inf, an overflowing sum included — has phase 0 */
    float mx = ax > ay ? ax : ay;
    if (!(mx > NAE_ATAN_TINY)) mx = NAE_ATAN_TINY;
    const float mn = ax > ay ? ay : ax;
    float r = u2f(NAE_RCP_MAGIC - f2u(mx));
    for (int it = 0; it < 3; it++) {
        const float e = fmaf(-mx, r, 1.0f);
        r = fmaf(r, e, r);
    }
    const float t = mn * r;
    const float s = t * t;
    float q = NAE_ATAN_C6 * NAE_ATAN_SCALE;
    q = fmaf(q, s, NAE_ATAN_C5 * NAE_ATAN_SCALE);
    q = fmaf(q, s, NAE_ATAN_C4 * NAE_ATAN_SCALE);
    q = fmaf(q, s, NAE_ATAN_C3 * NAE_ATAN_SCALE);
    q = fmaf(q, s, NAE_ATAN_C2 * NAE_ATAN_SCALE);
    q = fmaf(q, s, NAE_ATAN_C1 * NAE_ATAN_SCALE);
    q = fmaf(q, s, NAE_ATAN_C0 * NAE_ATAN_SCALE);
    uint32_t i = (uint32_t)(int32_t)rintf(q * t);
    if (ay > ax) i = 0x40000000u - i;
    const uint32_t m_re = (f2u(re) >> 31) ? 0xffffffffu : 0u, m_im = (f2u(im) >> 31) ? 0xffffffffu : 0u;
    i = (i ^ m_re) + (m_re & 0x80000001u);
    i = (i ^ m_im) - m_im;
    return (int32_t)i;
}

/* ------------------------------------------------------------------------------------------ K8 */
size_t orc_spectrum_frames(size_t T) { return T < NAE_FFT_N ? 0 : (T - NAE_FFT_N) / NAE_HOP + 1; }

void orc_spectrum_f32(const float* src, size_t T, int ch, float* dst)
{
    make_tables();
    const size_t F = orc_spectrum_frames(T);
    float xw[NAE_FFT_N];
    cf X[NAE_FFT_BINS];
    for (size_t f = 0; f < F; f++)
        for (int c = 0; c < ch; c++) {
            const float* s = src + (f * NAE_HOP) * (size_t)ch + c;
            for (int n = 0; n < NAE_FFT_N; n++) xw[n] = s[(size_t)n * ch] * HANN[n];
            rfft1024(xw, X);
            float* o = dst + (f * (size_t)ch + c) * NAE_FFT_BINS;
            for (int k = 0; k < NAE_FFT_BINS; k++) o[k] = sqrtf(X[k].x * X[k].x + X[k].y * X[k].y);
        }
}

/* ------------------------------------------------------------------------------------------ K7 */
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

static float rs_table[(NAE_RS_PHASES + 1) * NAE_RS_TAPS];

const float* orc_rs_table(double rate_eff)
{
    const double pi = 3.14159265358979323846;
    const double c = NAE_RS_CUTOFF * (rate_eff > 1.0 ? 1.0 / rate_eff : 1.0);
    const double half = NAE_RS_TAPS / 2.0;
    const double i0b = bessel_i0(NAE_RS_KAISER_BETA);
    for (int p = 0; p <= NAE_RS_PHASES; p++) {
        double row[NAE_RS_TAPS], sum = 0.0;
        for (int i = 0; i < NAE_RS_TAPS; i++) {
            const double x = (double)(i - (NAE_RS_TAPS / 2 - 1)) - (double)p / NAE_RS_PHASES;
            const double a = x / half;
            double w = 0.0;
            if (a > -1.0 && a < 1.0) w = bessel_i0(NAE_RS_KAISER_BETA * sqrt(1.0 - a * a)) / i0b;
            else if (a == 1.0 || a == -1.0) w = 1.0 / i0b;
            const double arg = pi * c * x;
            const double sinc = (fabs(arg) < 1e-12) ? 1.0 : sin(arg) / arg;
            row[i] = c * sinc * w;
            sum += row[i];
        }
        for (int i = 0; i < NAE_RS_TAPS; i++) rs_table[p * NAE_RS_TAPS + i] = (float)(row[i] / sum);
    }
    return rs_table;
}

int orc_stretch_plan_make(double rate, double pitch, size_t in_len, orc_stretch_plan* pl)
{
    memset(pl, 0, sizeof *pl);
    if (!(rate > 0.0) || !(pitch > 0.0)) return -1;
    double tempo = 1.0 / pitch, rho = rate * pitch;
    if (fabs(tempo - 1.0) < 1e-6) tempo = 1.0;
    if (fabs(rho - 1.0) < 1e-6) rho = 1.0;
    pl->pv_on = tempo != 1.0;
    pl->rs_on = rho != 1.0;
    if (pl->pv_on && (tempo < NAE_TEMPO_MIN || tempo > NAE_TEMPO_MAX)) return -2;
    if (pl->rs_on && (rho < NAE_RATE_MIN || rho > NAE_RATE_MAX)) return -2;
    pl->tempo_eff = tempo;
    pl->rate_eff = rho;
    pl->ha_q24 = (int64_t)llround((double)NAE_HOP * tempo * (double)(1 << NAE_HA_FRAC_BITS));
    pl->d0 = (int32_t)(pl->ha_q24 >> NAE_HA_FRAC_BITS);
    for (int i = 0; i < 2; i++) {
        const uint64_t d = (uint64_t)(pl->d0 + i);
        pl->r_q24[i] = (uint32_t)((((uint64_t)NAE_HOP << NAE_R_FRAC_BITS) + d / 2) / d);
    }
    pl->step_q32 = (uint64_t)llround(rho * 4294967296.0);
    pl->out_len = (size_t)floor((double)in_len / (tempo * rho) + 0.5);
    pl->rs_first = pl->pv_on && pl->rs_on && rho > 1.0;
    size_t pv_out;                                     /* samples the vocoder stage has to deliver */
    if (pl->rs_first) {
        pl->mid_len = (size_t)floor((double)in_len / rho + 0.5);   /* transposer output = vocoder input */
        pv_out = pl->out_len;
    } else if (pl->rs_on) {
        if (pl->out_len == 0) pl->mid_len = 0;
        else {
            const unsigned __int128 pos = (unsigned __int128)(pl->out_len - 1) * pl->step_q32;
            pl->mid_len = (size_t)(pos >> 32) + NAE_RS_TAPS / 2 + 1;
        }
        pv_out = pl->mid_len;
    } else {
        pl->mid_len = pl->out_len;
        pv_out = pl->out_len;
    }
    pl->frames = pl->pv_on ? (pv_out + NAE_FFT_N / 2 + NAE_HOP - 1) / NAE_HOP + 1 : 0;
    return 0;
}

static inline int64_t frame_start(const orc_stretch_plan* pl, int64_t f)
{
    /* analysis frame f is centred on input time (f-1)*Ha; arithmetic shift = floor */
    return (((f - 1) * pl->ha_q24 + ((int64_t)1 << (NAE_HA_FRAC_BITS - 1))) >> NAE_HA_FRAC_BITS) - NAE_FFT_N / 2;
}

/* one channel of the phase-vocoder stage; v[0..mid_len) is overwritten.  qs_tap (optional) receives the
 * synthesis phases, stride `tap_stride` int32 per frame. */
static void pv_channel(const float* src, size_t L, int ch, int c, const orc_stretch_plan* pl, size_t M, float* v,
                       int32_t* qs_tap, size_t tap_stride)
{
    float xw[NAE_FFT_N], y[NAE_FFT_N];
    cf X[NAE_FFT_BINS], Y[NAE_FFT_BINS];
    uint32_t qa[NAE_FFT_BINS], qa_prev[NAE_FFT_BINS], qs[NAE_FFT_BINS];
    const double two_pi = 6.283185307179586476925286766559;
    if (v) memset(v, 0, M * sizeof(float));
    int64_t s_prev = 0;
    for (size_t f = 0; f < pl->frames; f++) {
        const int64_t s = frame_start(pl, (int64_t)f);
        for (int n = 0; n < NAE_FFT_N; n++) {
            const int64_t i = s + n;
            const float x = (i >= 0 && (uint64_t)i < L) ? src[(size_t)i * ch + c] : 0.0f;
            xw[n] = x * HANN[n];
        }
        rfft1024(xw, X);
        for (int k = 0; k < NAE_FFT_BINS - 1; k++) qa[k] = (uint32_t)orc_atan2_q32(X[k].y, X[k].x);
        /* bin N/2 of a real signal is real: its phase is 0 or half a turn, read off the sign of the real part (the
         * imaginary part the split leaves behind is rounding residue of the order of 1e-16 relative) */
        qa[NAE_FFT_BINS - 1] = (X[NAE_FFT_BINS - 1].x < 0.0f) ? 0x80000000u : 0u;
        if (f == 0)
            memcpy(qs, qa, sizeof qs);
        else {
            const int64_t d = s - s_prev;
            const uint32_t R = pl->r_q24[d - pl->d0];
            for (int k = 0; k < NAE_FFT_BINS; k++) {
                const uint32_t e = (uint32_t)(((uint64_t)k * (uint64_t)d) & (NAE_FFT_N - 1)) << 22;
                const int32_t dw = (int32_t)(qa[k] - qa_prev[k] - e);
                const uint32_t adv = (uint32_t)((k * NAE_HOP) & (NAE_FFT_N - 1)) << 22;
                const int64_t scaled = ((int64_t)dw * (int64_t)R + ((int64_t)1 << (NAE_R_FRAC_BITS - 1))) >> NAE_R_FRAC_BITS;
                qs[k] += adv + (uint32_t)scaled;
            }
        }
        memcpy(qa_prev, qa, sizeof qa);
        s_prev = s;
        if (qs_tap) memcpy(qs_tap + f * tap_stride, qs, sizeof qs);
        if (!v) continue;
        for (int k = 0; k < NAE_FFT_BINS; k++) {
            const float mag = sqrtf(X[k].x * X[k].x + X[k].y * X[k].y);
            const double ph = two_pi * ((double)(int32_t)qs[k] * (1.0 / 4294967296.0));
            Y[k].x = mag * (float)cos(ph);
            Y[k].y = mag * (float)sin(ph);
        }
        irfft1024(Y, y);
        const int64_t o = ((int64_t)f - 1) * NAE_HOP - NAE_FFT_N / 2;
        for (int n = 0; n < NAE_FFT_N; n++) {
            const int64_t m = o + n;
            if (m >= 0 && (uint64_t)m < M) v[m] += HANN[n] * y[n];
        }
    }
    if (v)
        for (size_t m = 0; m < M; m++) v[m] *= NAE_OLA_GAIN;
}

static void rs_channel(const float* v, size_t M, size_t vstride, const orc_stretch_plan* pl, size_t n_out, const float* tab,
                       float* dst, int ch, int c)
{
    for (size_t j = 0; j < n_out; j++) {
        const unsigned __int128 pos = (unsigned __int128)j * pl->step_q32;
        const int64_t idx = (int64_t)(pos >> 32);
        const uint32_t frac = (uint32_t)pos;
        const uint32_t ph = frac >> 25;
        const float alpha = (float)(frac & 0x1FFFFFFu) * (1.0f / 33554432.0f);
        const float* t0 = tab + ph * NAE_RS_TAPS;
        const float* t1 = t0 + NAE_RS_TAPS;
        float acc = 0.0f;
        for (int i = 0; i < NAE_RS_TAPS; i++) {
            const int64_t m = idx - (NAE_RS_TAPS / 2 - 1) + i;
            const float x = (m >= 0 && (uint64_t)m < M) ? v[(size_t)m * vstride] : 0.0f;
            const float coef = t0[i] + alpha * (t1[i] - t0[i]);
            acc += coef * x;
        }
        dst[j * (size_t)ch + c] = acc;
    }
}

int orc_stretch_f32(const float* src, size_t L, int ch, double rate, double pitch, float* dst)
{
    make_tables();
    orc_stretch_plan pl;
    const int rc = orc_stretch_plan_make(rate, pitch, L, &pl);
    if (rc) return rc;
    if (!pl.pv_on && !pl.rs_on) {
        memmove(dst, src, L * (size_t)ch * sizeof(float));
        return 0;
    }
    const float* tab = pl.rs_on ? orc_rs_table(pl.rate_eff) : NULL;
    const size_t vlen = pl.mid_len > pl.out_len ? pl.mid_len : pl.out_len;
    float* v = pl.pv_on ? (float*)malloc((vlen + 1) * sizeof(float)) : NULL;
    float* w = pl.rs_first ? (float*)malloc((pl.out_len + 1) * sizeof(float)) : NULL;
    for (int c = 0; c < ch; c++) {
        if (pl.rs_first) {
            /* transposer first: src -> v[0..mid_len), then the vocoder on v -> w[0..out_len) */
            rs_channel(src + c, L, (size_t)ch, &pl, pl.mid_len, tab, v, 1, 0);
            pv_channel(v, pl.mid_len, 1, 0, &pl, pl.out_len, w, NULL, 0);
            for (size_t m = 0; m < pl.out_len; m++) dst[m * (size_t)ch + c] = w[m];
        } else if (pl.pv_on) {
            pv_channel(src, L, ch, c, &pl, pl.mid_len, v, NULL, 0);
            if (pl.rs_on) rs_channel(v, pl.mid_len, 1, &pl, pl.out_len, tab, dst, ch, c);
            else
                for (size_t m = 0; m < pl.out_len; m++) dst[m * (size_t)ch + c] = v[m];
        } else
            rs_channel(src + c, L, (size_t)ch, &pl, pl.out_len, tab, dst, ch, c);
    }
    free(v);
    free(w);
    return 0;
}

/* synthesis phases of the vocoder stage; with rs_first its input is the transposed signal */
int orc_pv_synth_phase(const float* src, size_t L, int ch, const orc_stretch_plan* pl, int32_t* qs)
{
    make_tables();
    if (!pl->pv_on) return -1;
    if (pl->rs_first) {
        const float* tab = orc_rs_table(pl->rate_eff);
        float* v = (float*)malloc((pl->mid_len + 1) * sizeof(float));
        for (int c = 0; c < ch; c++) {
            rs_channel(src + c, L, (size_t)ch, pl, pl->mid_len, tab, v, 1, 0);
            pv_channel(v, pl->mid_len, 1, 0, pl, pl->out_len, NULL, qs + (size_t)c * NAE_FFT_BINS, (size_t)ch * NAE_FFT_BINS);
        }
        free(v);
        return 0;
    }
    for (int c = 0; c < ch; c++)
        pv_channel(src, L, ch, c, pl, pl->mid_len, NULL, qs + (size_t)c * NAE_FFT_BINS, (size_t)ch * NAE_FFT_BINS);
    return 0;
}
