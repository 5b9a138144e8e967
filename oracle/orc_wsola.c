/* orc_wsola.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (see nae_oracle.h).
 *
 * K7 option A: the time-domain chain the reference actually runs for velocity_modifier /
 * pitch_modifier.  The arithmetic lives in a third-party dependency that is NOT under /root/reference:
 * SoundTouch 2.3.2 (xmake.lua:16), float-sample build, x86-64 with its SSE code paths, driven exactly as
 * audio-velocity.cpp does: construct, setSampleRate / setChannels / setRate(velocity) / setPitch(pitch)
 * (:369-385), putSamples per frame (:403), receiveSamples (:298), flush at end of input (:427).
 *
 * This file restates SoundTouch's published algorithm (its source is not available in this image and nothing was
 * fetched) as a streaming pipeline of three stages:
 *   TD   WSOLA time stretcher: 8 ms overlap, automatic sequence / seek-window lengths, exhaustive search of the
 *        overlap position by normalised cross-correlation with a parabolic preference for the middle of the
 *        window, linear cross-fade, fractional skip bookkeeping
 *   AA   64-tap Hamming-windowed-sinc anti-alias FIR
 *   CU   4-point cubic (Catmull-Rom form) rate transposer with a double-precision position accumulator
 * ordered TD -> AA -> CU for rate > 1, AA -> CU -> TD for rate == 1 and CU -> AA -> TD for rate < 1.
 * Floating-point evaluation orders follow the x86 SSE build (4-lane partial sums in the correlation, even/odd
 * tap sums in the stereo FIR), because the data-dependent arg-max makes the order observable.
 *
 * PARITY UNPINNED: the reference ships no SoundTouch fixtures and the library cannot be built or imported here,
 * so nothing in this file has been checked against SoundTouch output.  What the tests pin is (i) this file against
 * tests/golden/wsola_golden.npz, authored by a second, independently written restatement in numpy float32 block form
 * (tests/golden/st_numpy.py), (ii) the GPU path against this file, bit for bit, and (iii) signal-level properties
 * (length, pitch, tempo, chunk invariance).
 * Two points of this restatement are ASSUMPTIONS about the library, not pins: (a) the mono anti-alias FIR accumulates its float
 * products in ONE float accumulator in tap order (SoundTouch >= 2.1 typedefs LONG_SAMPLETYPE as float in float builds; the generic
 * evaluateFilterMono is a plain in-order loop) — but SoundTouch is usually built with fast-math and auto-vectorised, so the library's
 * actual mono summation order depends on its compiler; (b) the golden file above was regenerated from the numpy restatement when
 * (a) changed (round 4), so that gate compares two restatements by the same author with each other, not with SoundTouch.
 */
#include "nae_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ sample FIFO ---------------------------- */
typedef struct {
    float* d;
    size_t cap, beg, cnt; /* in sample-frames */
    int ch;
} fifo;

static void fifo_init(fifo* f, int ch)
{
    memset(f, 0, sizeof *f);
    f->ch = ch;
}
static void fifo_free(fifo* f)
{
    free(f->d);
    f->d = NULL;
}
static float* fifo_begin(fifo* f) { return f->d + f->beg * (size_t)f->ch; }
/* room for `need` more frames behind the stored ones; returns where they go */
static float* fifo_end(fifo* f, size_t need)
{
    if (f->beg + f->cnt + need > f->cap) {
        if (f->cnt + need > f->cap) {
            size_t ncap = (f->cnt + need) * 2 + 64;
            float* nd = (float*)malloc(ncap * (size_t)f->ch * sizeof(float));
            if (f->cnt) memcpy(nd, fifo_begin(f), f->cnt * (size_t)f->ch * sizeof(float));
            free(f->d);
            f->d = nd;
            f->cap = ncap;
        } else if (f->cnt)
            memmove(f->d, fifo_begin(f), f->cnt * (size_t)f->ch * sizeof(float));
        f->beg = 0;
    }
    return f->d + (f->beg + f->cnt) * (size_t)f->ch;
}
static void fifo_commit(fifo* f, size_t n) { f->cnt += n; }
static void fifo_put(fifo* f, const float* src, size_t n)
{
    if (n == 0) return;
    float* e = fifo_end(f, n);
    memcpy(e, src, n * (size_t)f->ch * sizeof(float));
    f->cnt += n;
}
static void fifo_pop(fifo* f, size_t n)
{
    if (n >= f->cnt) {
        f->cnt = 0;
        f->beg = 0;
    } else {
        f->beg += n;
        f->cnt -= n;
    }
}

/* ------------------------------------------------------------------ TD: WSOLA ------------------------------ */
typedef struct {
    int ch, sr;
    double tempo;
    int ovl, swl, seekl, sample_req;
    double nominal_skip, skip_fract;
    int beginning;
    float* mid;
    fifo in;
    /* debugging tap: the overlap offsets chosen so far */
    int32_t* offs;
    size_t n_offs, cap_offs;
} td_t;

static void td_params(td_t* t)
{
    /* sequence 90..40 ms and seek window 20..15 ms, linear in tempo over [0.5, 2], rounded to whole ms */
    const double seq_k = (40.0 - 90.0) / (2.0 - 0.5), seq_c = 90.0 - seq_k * 0.5;
    const double seek_k = (15.0 - 20.0) / (2.0 - 0.5), seek_c = 20.0 - seek_k * 0.5;
    double seq = seq_c + seq_k * t->tempo;
    seq = seq < 40.0 ? 40.0 : (seq > 90.0 ? 90.0 : seq);
    double seek = seek_c + seek_k * t->tempo;
    seek = seek < 15.0 ? 15.0 : (seek > 20.0 ? 20.0 : seek);
    const int seq_ms = (int)(seq + 0.5), seek_ms = (int)(seek + 0.5);
    int ovl = (t->sr * 8) / 1000;
    if (ovl < 16) ovl = 16;
    ovl -= ovl % 8;
    t->ovl = ovl;
    t->swl = (t->sr * seq_ms) / 1000;
    if (t->swl < 2 * ovl) t->swl = 2 * ovl;
    t->seekl = (t->sr * seek_ms) / 1000;
    t->nominal_skip = t->tempo * (double)(t->swl - ovl);
    const int intskip = (int)(t->nominal_skip + 0.5);
    const int a = intskip + ovl, b = t->swl;
    t->sample_req = (a > b ? a : b) + t->seekl;
}

static void td_init(td_t* t, int sr, int ch, double tempo)
{
    memset(t, 0, sizeof *t);
    t->ch = ch;
    t->sr = sr;
    t->tempo = tempo;
    td_params(t);
    t->mid = (float*)calloc((size_t)t->ovl * (size_t)ch, sizeof(float));
    t->beginning = 1;
    fifo_init(&t->in, ch);
}
static void td_free(td_t* t)
{
    free(t->mid);
    free(t->offs);
    fifo_free(&t->in);
}

/* normalised cross-correlation of the candidate with the stored tail; four partial sums over (index mod 4),
 * each accumulated in index order, combined left to right (the SSE build's order) */
static double td_corr(const td_t* t, const float* x)
{
    const int n = t->ch * t->ovl;
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 4)
        for (int l = 0; l < 4; l++) {
            const float v = x[i + l];
            s[l] = s[l] + v * t->mid[i + l];
            q[l] = q[l] + v * v;
        }
    const float norm = ((q[0] + q[1]) + q[2]) + q[3];
    const float sum = ((s[0] + s[1]) + s[2]) + s[3];
    return (double)sum / sqrt(norm < 1e-9 ? 1.0 : (double)norm);
}

static int td_seek(const td_t* t, const float* ref)
{
    double best = (td_corr(t, ref) + 0.1) * 0.75;
    int best_i = 0;
    for (int i = 1; i < t->seekl; i++) {
        double c = td_corr(t, ref + (size_t)t->ch * (size_t)i);
        const double u = (double)(2 * i - t->seekl) / (double)t->seekl;
        c = (c + 0.1) * (1.0 - 0.25 * u * u);
        if (c > best) {
            best = c;
            best_i = i;
        }
    }
    return best_i;
}

static void td_overlap(const td_t* t, float* out, const float* in)
{
    if (t->ch == 2) {
        const float step = 1.0f / (float)t->ovl;
        float f1 = 0.0f, f2 = 1.0f;
        for (int i = 0; i < 2 * t->ovl; i += 2) {
            out[i] = in[i] * f1 + t->mid[i] * f2;
            out[i + 1] = in[i + 1] * f1 + t->mid[i + 1] * f2;
            f1 += step;
            f2 -= step;
        }
    } else {
        float m1 = 0.0f, m2 = (float)t->ovl;
        for (int i = 0; i < t->ovl; i++) {
            out[i] = (in[i] * m1 + t->mid[i] * m2) / (float)t->ovl;
            m1 += 1.0f;
            m2 -= 1.0f;
        }
    }
}

static void td_note_offset(td_t* t, int off)
{
    if (t->n_offs == t->cap_offs) {
        t->cap_offs = t->cap_offs ? 2 * t->cap_offs : 256;
        t->offs = (int32_t*)realloc(t->offs, t->cap_offs * sizeof(int32_t));
    }
    t->offs[t->n_offs++] = off;
}

static void td_process(td_t* t, fifo* out)
{
    const size_t chs = (size_t)t->ch;
    while ((long long)t->in.cnt >= t->sample_req) {
        int offset = 0;
        const float* base = fifo_begin(&t->in);
        if (!t->beginning) {
            offset = td_seek(t, base);
            td_note_offset(t, offset);
            float* o = fifo_end(out, (size_t)t->ovl);
            td_overlap(t, o, base + chs * (size_t)offset);
            fifo_commit(out, (size_t)t->ovl);
            offset += t->ovl;
        } else {
            /* first sequence: no cross-fade; the missing overlap is charged to the skip accumulator */
            t->beginning = 0;
            const int skip = (int)(t->tempo * (double)t->ovl + 0.5 * (double)t->seekl + 0.5);
            t->skip_fract -= (double)skip;
            if (t->skip_fract <= -t->nominal_skip) t->skip_fract = -t->nominal_skip;
        }
        const int body = t->swl - 2 * t->ovl;
        fifo_put(out, base + chs * (size_t)offset, (size_t)body);
        base = fifo_begin(&t->in); /* (out is a different FIFO; base is unchanged, re-read for clarity) */
        memcpy(t->mid, base + chs * (size_t)(offset + body), chs * (size_t)t->ovl * sizeof(float));
        t->skip_fract += t->nominal_skip;
        const int ovl_skip = (int)t->skip_fract;
        t->skip_fract -= (double)ovl_skip;
        fifo_pop(&t->in, (size_t)ovl_skip);
    }
}

/* ------------------------------------------------------------------ AA: anti-alias FIR --------------------- */
#define AA_LEN 64
typedef struct {
    int ch;
    float coef[AA_LEN];
    fifo in;
} aa_t;

static void aa_design(float* coef, double cutoff)
{
    const double pi = 3.14159265358979323846;
    double work[AA_LEN], sum = 0.0;
    const double wc = 2.0 * pi * cutoff, tc = (2.0 * pi) / (double)AA_LEN;
    for (int i = 0; i < AA_LEN; i++) {
        const double c = (double)i - (double)(AA_LEN / 2);
        const double a = c * wc;
        const double h = (a != 0.0) ? sin(a) / a : 1.0;
        const double w = 0.54 + 0.46 * cos(tc * c);
        work[i] = w * h;
        sum += work[i];
    }
    /* the library scales to a 2^14 integer grid and adds a rounding half even in the float build, where nothing
     * is truncated afterwards: every tap carries that +-0.5/16384 */
    const double scale = 16384.0 / sum;
    for (int i = 0; i < AA_LEN; i++) {
        double v = work[i] * scale;
        v += (v >= 0.0) ? 0.5 : -0.5;
        coef[i] = (float)v / 16384.0f;
    }
}

static void aa_init(aa_t* a, int ch, double rate)
{
    a->ch = ch;
    aa_design(a->coef, rate > 1.0 ? 0.5 / rate : 0.5 * rate);
    fifo_init(&a->in, ch);
}

static void aa_process(aa_t* a, fifo* out)
{
    const size_t n = a->in.cnt;
    if (n < AA_LEN) return;
    const float* src = fifo_begin(&a->in);
    size_t count;
    if (a->ch == 2) {
        /* SSE stereo kernel: an even number of outputs per call; per channel, the even taps and the odd taps
         * are summed separately in tap order and then added */
        count = (n - AA_LEN) & ~(size_t)1;
        if (count < 2) return;
        float* dst = fifo_end(out, count);
        for (size_t j = 0; j < count; j++)
            for (int c = 0; c < 2; c++) {
                float ev = 0.0f, od = 0.0f;
                for (int k = 0; k < AA_LEN; k += 2) {
                    ev = ev + src[2 * (j + (size_t)k) + (size_t)c] * a->coef[k];
                    od = od + src[2 * (j + (size_t)k + 1) + (size_t)c] * a->coef[k + 1];
                }
                dst[2 * j + (size_t)c] = od + ev;
            }
    } else {
        /* generic mono kernel: products accumulated in tap order in LONG_SAMPLETYPE, which SoundTouch >= 2.1 typedefs as FLOAT in
         * float builds ("to enable efficient autovectorization", STTypes.h) — rounds 1-3 restated the older double accumulator */
        count = n - AA_LEN;
        if (count == 0) return;
        float* dst = fifo_end(out, count);
        for (size_t j = 0; j < count; j++) {
            float s = 0.0f;
            for (int k = 0; k < AA_LEN; k++) s = s + src[j + (size_t)k] * a->coef[k];
            dst[j] = s;
        }
    }
    fifo_commit(out, count);
    fifo_pop(&a->in, count);
}

/* ------------------------------------------------------------------ CU: cubic transposer ------------------- */
typedef struct {
    int ch;
    double rate, fract;
    fifo in;
} cu_t;

static void cu_init(cu_t* c, int ch, double rate)
{
    c->ch = ch;
    c->rate = rate;
    c->fract = 0.0;
    fifo_init(&c->in, ch);
}

void orc_st_cubic_weights(float x, float* y)
{
    static const float k[16] = {-0.5f, 1.0f, -0.5f, 0.0f, 1.5f, -2.5f, 0.0f, 1.0f,
                                -1.5f, 2.0f, 0.5f,  0.0f, 0.5f, -0.5f, 0.0f, 0.0f};
    const float x2 = x, x1 = x2 * x2, x0 = x1 * x2, x3 = 1.0f;
    for (int m = 0; m < 4; m++)
        y[m] = ((k[4 * m] * x0 + k[4 * m + 1] * x1) + k[4 * m + 2] * x2) + k[4 * m + 3] * x3;
}

static void cu_process(cu_t* c, fifo* out)
{
    const long long n = (long long)c->in.cnt;
    const long long end = n - 4;
    const size_t chs = (size_t)c->ch;
    const size_t demand = (size_t)((double)n / c->rate) + 8;
    const float* p = fifo_begin(&c->in);
    float* dst = fifo_end(out, demand);
    long long used = 0;
    size_t made = 0;
    while (used < end) {
        float y[4];
        orc_st_cubic_weights((float)c->fract, y);
        for (size_t k = 0; k < chs; k++)
            dst[made * chs + k] = ((y[0] * p[k] + y[1] * p[chs + k]) + y[2] * p[2 * chs + k]) + y[3] * p[3 * chs + k];
        made++;
        c->fract += c->rate;
        const int whole = (int)c->fract;
        c->fract -= (double)whole;
        p += chs * (size_t)whole;
        used += whole;
    }
    fifo_commit(out, made);
    fifo_pop(&c->in, (size_t)(used > 0 ? used : 0));
}

/* ------------------------------------------------------------------ the chain ------------------------------ */
struct orc_st {
    int ch, sr;
    double rate, tempo;
    td_t td;
    aa_t aa;
    cu_t cu;
    fifo mid; /* between AA and CU */
    fifo out;
    double expected;
    long long received;
};

int orc_st_create(int sample_rate, int ch, double rate, double pitch, orc_st** h)
{
    if (!h || (ch != 1 && ch != 2) || sample_rate < 8000 || sample_rate > 48000) return -1;
    if (!(rate > 0.0) || !(pitch > 0.0)) return -1;
    orc_st* s = (orc_st*)calloc(1, sizeof *s);
    s->ch = ch;
    s->sr = sample_rate;
    s->tempo = 1.0 / pitch; /* virtual tempo 1 */
    s->rate = pitch * rate;
    td_init(&s->td, sample_rate, ch, s->tempo);
    aa_init(&s->aa, ch, s->rate);
    cu_init(&s->cu, ch, s->rate);
    fifo_init(&s->mid, ch);
    fifo_init(&s->out, ch);
    *h = s;
    return 0;
}

void orc_st_destroy(orc_st* s)
{
    if (!s) return;
    td_free(&s->td);
    fifo_free(&s->aa.in);
    fifo_free(&s->cu.in);
    fifo_free(&s->mid);
    fifo_free(&s->out);
    free(s);
}

/* transposer stage = AA + CU in the order the rate asks for; input in `src` FIFO semantics via put */
static void rt_put(orc_st* s, const float* x, size_t n, fifo* out)
{
    if (n == 0) return;
    if (s->rate < 1.0) {
        fifo_put(&s->cu.in, x, n);
        cu_process(&s->cu, &s->aa.in);
        aa_process(&s->aa, out);
    } else {
        fifo_put(&s->aa.in, x, n);
        aa_process(&s->aa, &s->cu.in);
        cu_process(&s->cu, out);
    }
}

void orc_st_put(orc_st* s, const float* x, size_t n)
{
    s->expected += (double)n / (s->rate * s->tempo);
    if (s->rate <= 1.0) {
        /* transposer first; its output feeds the stretcher */
        rt_put(s, x, n, &s->td.in);
        td_process(&s->td, &s->out);
    } else {
        fifo_put(&s->td.in, x, n);
        td_process(&s->td, &s->mid);
        /* hand everything the stretcher produced to the transposer */
        const size_t m = s->mid.cnt;
        if (m) {
            rt_put(s, fifo_begin(&s->mid), m, &s->out);
            fifo_pop(&s->mid, m);
        }
    }
}

size_t orc_st_available(const orc_st* s) { return s->out.cnt; }

size_t orc_st_receive(orc_st* s, float* dst, size_t max)
{
    const size_t n = max < s->out.cnt ? max : s->out.cnt;
    if (n) memcpy(dst, fifo_begin(&s->out), n * (size_t)s->ch * sizeof(float));
    fifo_pop(&s->out, n);
    s->received += (long long)n;
    return n;
}

void orc_st_flush(orc_st* s)
{
    float zeros[128 * 2];
    memset(zeros, 0, sizeof zeros);
    long long still = (long long)(s->expected + 0.5) - s->received;
    if (still < 0) still = 0;
    for (int i = 0; still > (long long)s->out.cnt && i < 200; i++) orc_st_put(s, zeros, 128);
    if ((long long)s->out.cnt > still) s->out.cnt = (size_t)still;
    /* the stretcher's input side starts over (what is left in the transposer FIFOs stays, as in the library) */
    s->td.in.cnt = 0;
    s->td.in.beg = 0;
    memset(s->td.mid, 0, (size_t)s->td.ovl * (size_t)s->ch * sizeof(float));
    s->td.beginning = 1;
    s->td.skip_fract = 0.0;
}

size_t orc_st_offsets(const orc_st* s, int32_t* dst, size_t max)
{
    const size_t n = max < s->td.n_offs ? max : s->td.n_offs;
    if (n && dst) memcpy(dst, s->td.offs, n * sizeof(int32_t));
    return s->td.n_offs;
}

void orc_st_params(const orc_st* s, int* v /* ovl, swl, seekl, sample_req */)
{
    v[0] = s->td.ovl;
    v[1] = s->td.swl;
    v[2] = s->td.seekl;
    v[3] = s->td.sample_req;
}

const float* orc_st_aa_coef(const orc_st* s) { return s->aa.coef; }

/* whole buffer: one put, flush, receive everything.  dst holds at least orc_st_out_bound(L) frames */
size_t orc_st_out_bound(size_t L, double rate, double pitch)
{
    (void)pitch;
    return (size_t)((double)L / rate + 0.5) + 16; /* rate*tempo = rate_in */
}

int orc_st_process_f32(const float* src, size_t L, int ch, int sample_rate, double rate, double pitch, float* dst,
                       size_t* out_len)
{
    orc_st* s;
    const int rc = orc_st_create(sample_rate, ch, rate, pitch, &s);
    if (rc) return rc;
    orc_st_put(s, src, L);
    orc_st_flush(s);
    *out_len = orc_st_receive(s, dst, s->out.cnt);
    orc_st_destroy(s);
    return 0;
}
