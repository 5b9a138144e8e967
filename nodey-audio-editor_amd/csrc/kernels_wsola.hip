// kernels_wsola.hip — K7 option A: the SoundTouch-shaped time-domain chain on gfx950.
//
// Replaces what soundtouch::SoundTouch does between putSamples and receiveSamples for
// /root/reference/src/processor/audio-velocity.cpp:369-428 (library absent: algorithm restated).
// Three kernels, all bit-exact against oracle/orc_wsola.c (this file is built with -ffp-contract=off):
//
//   st_td_kernel   WSOLA stretcher.  One 256-thread workgroup walks one stream's sequences in order (the chosen
//                  offset of sequence k decides the tail that sequence k+1 is matched against, so sequences of a
//                  stream are serial; streams are parallel).  Per sequence the seek window is staged in LDS and
//                  every thread scores 4 candidate offsets at once: candidates 2 frames (16 B) apart see the same
//                  4-float groups one step later, so one LDS read of the window feeds 4 candidates and the stored
//                  tail is read once per step through a 4-deep register window.  The window is stored transposed
//                  (row = unit mod R) so that a half-wave reads consecutive LDS words: conflict-free.
//                  VALU-bound: 13 flop per candidate per 4 floats (4 mul + 4 add correlation, 1 mul + 4 add norm).
//   st_aa_kernel   64-tap FIR, 4 consecutive outputs per thread from a 67-frame register window, even and odd taps
//                  summed separately (the SSE stereo order); mono: one float accumulator, taps in order.
//   st_cu_kernel   4-point cubic read at host-tabulated positions (the position accumulator is a sequential
//                  double recurrence that does not depend on the audio: st_chain.h).
#include "st_chain.h"
#include <type_traits>

namespace nae {

// device-side views: the host views plus "interleaved stereo, 8-byte aligned" (one 8-byte access per frame)
struct DView { StView v; int vec2; };
struct DOut { StOut o; int vec2; };
template <int CH> struct Frame { float x[CH]; };

template <int CH>
__device__ __forceinline__ Frame<CH> ld_frame(const DView& d, const float* sbase, long long a)
{
    Frame<CH> f;
    if (a >= d.v.valid_end) {
#pragma unroll
        for (int c = 0; c < CH; c++) f.x[c] = 0.0f;
    } else if (CH == 2 && d.vec2) {
        const float2 t = *reinterpret_cast<const float2*>(sbase + (a - d.v.origin) * 2);
        f.x[0] = t.x;
        f.x[CH - 1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) f.x[c] = sbase[c * d.v.cs + (a - d.v.origin) * d.v.fs];
    }
    return f;
}

template <int CH>
__device__ __forceinline__ void st_frame(const DOut& d, float* obase, long long a, const Frame<CH>& f)
{
    if (CH == 2 && d.vec2) {
        *reinterpret_cast<float2*>(obase + (a - d.o.origin) * 2) = float2{f.x[0], f.x[CH - 1]};
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) obase[c * d.o.cs + (a - d.o.origin) * d.o.fs] = f.x[c];
    }
}

static DView dview(const StView& v, int ch)
{
    const bool vec = ch == 2 && v.cs == 1 && v.fs == 2 && (reinterpret_cast<uintptr_t>(v.base) & 7) == 0 && (v.ss & 1) == 0;
    return DView{v, vec ? 1 : 0};
}
static DOut dout(const StOut& o, int ch)
{
    const bool vec = ch == 2 && o.cs == 1 && o.fs == 2 && (reinterpret_cast<uintptr_t>(o.base) & 7) == 0 && (o.ss & 1) == 0;
    return DOut{o, vec ? 1 : 0};
}

struct TdParams {
    int ovl, seekl, body, first_skip;
    double nominal_skip;
    long long ip0, op0, nseq, out_limit;
    double skip0;
    int begin0;
    int S;          // LDS row stride of the window, in units
};

// geometry of the transposed window.  A thread owns NC candidates P units apart (P = 4 / CH units make one 4-float
// group: 2 frames for stereo, 4 samples for mono), a workgroup of 1024 / NC threads covers 1024 candidates.
// Unit u lives at row u mod R, column u div R with R = NC * P; rows R .. R+P-2 repeat rows 0 .. P-2 one column on, so
// that a group that wraps around the rows is still read at a fixed row offset.
//   NC = 4: 256 threads, one window read feeds 4 candidates (throughput shape: several streams per CU)
//   (round 5 measured two shapes that waste fewer of the 1024 candidate slots on C5's 912-candidate seek window — 5 candidates on 192 threads and 3 on
//   320 threads, 960 slots each — at 10.6-10.9 and 13.5 ms against 9.6 ms for NC = 4: three waves per SIMD hide less, three candidates per window
//   read load the LDS more; profiles/r05_wsola_sq.md.  Not kept.)
//   NC = 2: 512 threads, 8 waves per stream (latency shape: at most ~2 streams per CU, down to a lone stream)
constexpr int td_threads(int nc) { return 1024 / nc; }
template <int CH, int NC> struct TdGeo {
    static constexpr int P = 4 / CH;
    static constexpr int R = NC * P;
    static constexpr int ROWS = R + P - 1;
    static constexpr int LANES = 64 / P;          // lanes that share one p
    static constexpr int THREADS = td_threads(NC);
    static constexpr int WAVES = THREADS / 64;
    static constexpr int PF = (1344 + THREADS - 1) / THREADS;   // window frames one thread stages (seek + overlap <= 1344)
    static constexpr int CP = 4;                  // frames one thread has in flight while copying the body
};

struct Best {
    double v;
    int i;
};
__device__ __forceinline__ Best better(Best a, Best b)
{
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__device__ unsigned g_td_arrivals[8 * 4 * 16];

// KS: the LDS row stride as a compile-time value (0: p.S at run time).  With it the eight rows a search trip reads sit at immediate offsets of ONE
// address register instead of eight registers that are each advanced per trip: 7 of a trip's 229 vector instructions (round 5; instantiated for the
// stereo 48-kHz geometry — overlap 384 frames — that BASELINE.json's configs run).
template <int CH, int NC, int KS = 0>
__global__ __launch_bounds__(td_threads(NC), 4) void st_td_kernel(DView in, TdParams p_in, DOut out, float* __restrict__ mid_state,
                                                         int32_t* __restrict__ offs_dbg, long long offs_stride)
{
    TdParams p = p_in;
    if (KS != 0) p.S = KS;
    using G = TdGeo<CH, NC>;
    constexpr int T = G::THREADS;
    extern __shared__ __attribute__((aligned(16))) float td_smem[];
    const int ng = p.ovl * CH / 4;                       // 4-float groups per candidate
    // kReuse (throughput shape): two window buffers.  The frames a sequence copies to the output (body + new tail) are most
    // of the NEXT sequence's seek window, so they are dropped into the other buffer on their way through the registers
    // and only the part of the next window outside the copied range is fetched again: every input frame is read about
    // once instead of twice.
    constexpr bool kReuse = NC >= 4;
    const int wsz = G::ROWS * p.S * CH;
    float* win = td_smem;                                // [1 or 2][ROWS][S] units
    float* mid = win + (kReuse ? 2 : 1) * wsz;           // [ovl*CH] (+16 pad)
    float* ramp1 = mid + p.ovl * CH + 16;                // [ovl] cross-fade weights (stereo)
    float* ramp2 = ramp1 + p.ovl;
    __shared__ Best wave_best[G::WAVES];
    __shared__ int s_off, s_nan0;

    int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long s = blockIdx.x;
    const float* sbase = in.v.base + s * in.v.ss;
    float* obase = out.o.base + s * out.o.ss;
    const int nunits = p.seekl + p.ovl;

    if (p.begin0 || !mid_state) {
        for (int i = tid; i < p.ovl * CH; i += T) mid[i] = 0.0f;
    } else {
        for (int i = tid; i < p.ovl * CH; i += T) mid[i] = mid_state[s * (long long)(p.ovl * CH) + i];
    }
    for (int i = tid; i < 16; i += T) mid[p.ovl * CH + i] = 0.0f;
    if (CH == 2 && tid == 0) {
        const float step = 1.0f / (float)p.ovl;
        float f1 = 0.0f, f2 = 1.0f;
        for (int i = 0; i < p.ovl; i++) {
            ramp1[i] = f1;
            ramp2[i] = f2;
            f1 += step;
            f2 -= step;
        }
    }

    const int pp = lane / G::LANES, bl = lane % G::LANES;
    const int b = w * G::LANES + bl;
    const int c0 = G::R * b + pp;                        // this thread's candidates: c0 + P*q, q < NC
    const int wave_c0 = G::R * (w * G::LANES);           // smallest candidate of the wave
    const int xoff = (pp * p.S + b) * CH;

    // window of the sequence that starts at frame `at`: global -> registers -> LDS.  The loads are issued one
    // sequence ahead (the positions do not depend on the audio) so their latency hides behind the scoring loop.
    Frame<CH> pf[G::PF];
    auto window_load = [&](long long at) {
#pragma unroll
        for (int j = 0; j < G::PF; j++) {
            const int u = tid + j * T;
            if (u < nunits) pf[j] = ld_frame<CH>(in, sbase, at + u);
        }
    };
    auto put_unit = [&](float* wn, int u, const Frame<CH>& f) {
        const int row = u % G::R, col = u / G::R;
#pragma unroll
        for (int c = 0; c < CH; c++) wn[(row * p.S + col) * CH + c] = f.x[c];
        if (row < G::P - 1 && col >= 1) {
#pragma unroll
            for (int c = 0; c < CH; c++) wn[((G::R + row) * p.S + col - 1) * CH + c] = f.x[c];
        }
    };
    auto window_store = [&](float* wn) {
#pragma unroll
        for (int j = 0; j < G::PF; j++) {
            const int u = tid + j * T;
            if (u < nunits) put_unit(wn, u, pf[j]);
        }
    };

    long long ip = p.ip0, op = p.op0;
    double skip = p.skip0;
    bool begin = p.begin0 != 0;
    float* wcur = win;
    if (!begin && p.nseq > 0) {
        window_load(ip);
        window_store(wcur);
    }
    if (tid == 0) s_nan0 = 0;
    // The workgroups of a CU (four at C5) take turns at the higher issue priority, two at a time, in time slices of 2^21 shader
    // cycles (~1 ms): at equal priority the hardware serves them by age, they crowd the 4-waves-per-SIMD issue point together
    // and the oldest finishes first; with two favoured at a time the search ran 11.0 -> 9.7 ms at 1024 streams, 8.6 -> 8.1 at
    // 512, unchanged below (slices of 2^16 .. 2^20 and four-way turns: 10.2-10.3; 2^23: 11.3).  A workgroup learns its turn
    // from an arrival counter per physical CU (never reset: only the parity is used); both read the same clock.
    constexpr int kTurnBit = 21;
    __shared__ int s_turn;
    if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
        s_turn = (int)(atomicAdd(&g_td_arrivals[(xcc * 4 + ((hw >> 13) & 3u)) * 16 + ((hw >> 8) & 15u)], 1u) & 3u);
    }
    __syncthreads();
    const int my_turn = __builtin_amdgcn_readfirstlane(s_turn) & 1;

#pragma unroll 1
    for (long long k = 0; k < p.nseq; k++) {
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if ((int)((now >> kTurnBit) & 1) == my_turn) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        // Values derived from the thread index (LDS offsets of the copy and window stores, ramp addresses ...) are invariant
        // across sequences; hoisted out of this loop they exceed the register budget and are spilled — and a scratch line
        // re-read once per sequence has left the L2 by then (measured: +10 KB of HBM reads per sequence and stream).
        // Hiding the index from the optimiser once per sequence makes it recompute them instead (a few dozen instructions).
        asm volatile("" : "+v"(tid));
        // where the next sequence starts (the library's skip bookkeeping, in its order of operations)
        double skip_n = skip;
        if (begin) {
            skip_n -= (double)p.first_skip;
            if (skip_n <= -p.nominal_skip) skip_n = -p.nominal_skip;
        }
        skip_n += p.nominal_skip;
        const int adv = (int)skip_n;
        skip_n -= (double)adv;
        const long long ip_n = ip + adv;
        const bool more = k + 1 < p.nseq;
        // few streams per CU: fetch the next window now, so that its latency hides behind the scoring loop; with many
        // streams other workgroups hide it and the 12 registers are better spent on occupancy
        constexpr bool kAhead = NC < 4;
        if (more && kAhead) window_load(ip_n);
        float* wnext = kReuse ? (wcur == win ? win + wsz : win) : win;
        const float* xrow = wcur + xoff;

        int offset = 0;
        if (!begin) {
            // ---- score the candidates
            Best mine{-__builtin_inf(), 0x7fffffff};
            if (wave_c0 < p.seekl) {
                float sc[NC][4], sn[NC][4], M[NC][4];
#pragma unroll
                for (int q = 0; q < NC; q++)
#pragma unroll
                    for (int l = 0; l < 4; l++) sc[q][l] = sn[q][l] = M[q][l] = 0.0f;
                const int gsteps = (ng + NC - 1 + NC - 1) / NC;
                // one trip = NC group steps.  Trips whose steps all lie inside every candidate's range run without
                // predicates (the compiler otherwise turns each guarded accumulate into add + v_cndmask: +30 %).
                auto trip = [&](int g, auto guarded) {
#pragma unroll
                    for (int e = 0; e < NC; e++) {
                        const int Gs = NC * g + e;
                        float X[4], Q[4];
#pragma unroll
                        for (int m = 0; m < G::P; m++) {
                            if (CH == 2) {
                                // one 8-byte read per stereo frame (the ds_read2_b32 the compiler picks for two floats of unknown alignment is banked per dword:
                                // 32 lanes 8 bytes apart collide 2-way — 0.39 of the kernel's LDS cycles in round 4); with a compile-time stride it sits at an
                                // IMMEDIATE offset of the trip's one address register (ds_read_b64 takes 16 bits of offset, ds_read2_b32 8 bits of dwords)
                                typedef __attribute__((address_space(3))) const volatile unsigned long long lds_u64;   // volatile: two of them must not be paired into a ds_read2_b64 (twice the LDS time per byte)
                                const unsigned long long u = *(lds_u64*)(xrow + ((G::P * e + m) * (KS != 0 ? KS : p.S) + g) * 2);
                                X[m * 2] = __uint_as_float((unsigned)u);
                                X[m * 2 + 1] = __uint_as_float((unsigned)(u >> 32));
                            } else {
#pragma unroll
                                for (int c = 0; c < CH; c++) X[m * CH + c] = xrow[((G::P * e + m) * p.S + g) * CH + c];
                            }
                        }
#pragma unroll
                        for (int l = 0; l < 4; l++) Q[l] = X[l] * X[l];
                        {
                            const float4 mv = *reinterpret_cast<const float4*>(mid + 4 * (decltype(guarded)::value ? (Gs < ng ? Gs : ng) : Gs));
                            M[e][0] = mv.x; M[e][1] = mv.y; M[e][2] = mv.z; M[e][3] = mv.w;
                        }
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            const int j = Gs - q;
                            if (!decltype(guarded)::value || (j >= 0 && j < ng)) {
                                const int me = ((e - q) % NC + NC) % NC;
#pragma unroll
                                for (int l = 0; l < 4; l++) {
                                    sc[q][l] = sc[q][l] + X[l] * M[me][l];
                                    sn[q][l] = sn[q][l] + Q[l];
                                }
                            }
                        }
                    }
                };
                const int g_lo = NC > 1 ? 1 : 0;                    // first trip with NC * g >= NC - 1
                int g_hi = (ng - NC) / NC + 1;                      // one past the last trip with NC * g + NC - 1 < ng
                if (g_hi < g_lo) g_hi = g_lo;
                if (g_hi > gsteps) g_hi = gsteps;
                for (int g = 0; g < g_lo && g < gsteps; g++) trip(g, std::true_type{});
#pragma unroll 1
                for (int g = g_lo; g < g_hi; g++) trip(g, std::false_type{});
#pragma unroll 1
                for (int g = g_hi; g < gsteps; g++) trip(g, std::true_type{});
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    const int cand = c0 + G::P * q;
                    const float norm = ((sn[q][0] + sn[q][1]) + sn[q][2]) + sn[q][3];
                    const float sum = ((sc[q][0] + sc[q][1]) + sc[q][2]) + sc[q][3];
                    double c = (double)sum / __builtin_sqrt(norm < 1e-9 ? 1.0 : (double)norm);
                    const double u = (double)(2 * cand - p.seekl) / (double)p.seekl;
                    c = (c + 0.1) * (1.0 - 0.25 * u * u);
                    if (cand < p.seekl) {
                        if (cand == 0 && c != c) s_nan0 = 1;
                        if (c > mine.v) mine = Best{c, cand};
                    }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                Best o;
                o.v = __shfl_xor(mine.v, d);
                o.i = __shfl_xor(mine.i, d);
                mine = better(mine, o);
            }
            if (lane == 0) wave_best[w] = mine;
            __syncthreads();
            if (tid == 0) {
                Best r = wave_best[0];
                for (int i = 1; i < G::WAVES; i++) r = better(r, wave_best[i]);
                // a NaN score at offset 0 can never be beaten (every later comparison is false)
                const int o = (s_nan0 || r.i == 0x7fffffff) ? 0 : r.i;
                s_off = o;
                s_nan0 = 0;
                if (offs_dbg) offs_dbg[s * offs_stride + k - 1 + (p.begin0 ? 0 : 1)] = o;
            }
            __syncthreads();
            const int best = s_off;

            // ---- cross-fade the candidate into the stored tail
            for (int i = tid; i < p.ovl && op + i < p.out_limit; i += T) {
                const int u = best + i;
                const float* xu = wcur + ((u % G::R) * p.S + u / G::R) * CH;
                Frame<CH> y;
                if (CH == 2) {
                    const float f1 = ramp1[i], f2 = ramp2[i];
#pragma unroll
                    for (int c = 0; c < CH; c++) y.x[c] = xu[c] * f1 + mid[i * CH + c] * f2;
                } else {
                    const float m1 = (float)i, m2 = (float)(p.ovl - i);
                    y.x[0] = (xu[0] * m1 + mid[i] * m2) / (float)p.ovl;
                }
                st_frame<CH>(out, obase, op + i, y);
            }
            op += p.ovl;
            offset = best + p.ovl;
        } else {
            // first sequence of a stream: nothing to fade into; the missing overlap was charged to the skip accumulator
            begin = false;
        }
        // ---- body of the sequence and the new tail: loads of CP frames per thread in flight before their stores
        const long long bsrc = ip + offset;
        // copied frame i (0 <= i < body + ovl) is unit i + un0 of the next window
        const int un0 = (int)(bsrc - ip_n);
        const bool keep = kReuse && more;
        auto keep_unit = [&](int i, const Frame<CH>& f) {
            const int u = i + un0;
            if (keep && u >= 0 && u < nunits) put_unit(wnext, u, f);
        };
        Frame<CH> tail{};
        if (tid < p.ovl) tail = ld_frame<CH>(in, sbase, bsrc + p.body + tid);
        Frame<CH> tail2{};
        if (tid + T < p.ovl) tail2 = ld_frame<CH>(in, sbase, bsrc + p.body + tid + T);
        for (int i0 = 0; i0 < p.body; i0 += G::CP * T) {
            Frame<CH> cp[G::CP];
#pragma unroll
            for (int j = 0; j < G::CP; j++) {
                const int i = i0 + j * T + tid;
                if (i < p.body) cp[j] = ld_frame<CH>(in, sbase, bsrc + i);
            }
#pragma unroll
            for (int j = 0; j < G::CP; j++) {
                const int i = i0 + j * T + tid;
                if (i < p.body && op + i < p.out_limit) st_frame<CH>(out, obase, op + i, cp[j]);
                if (i < p.body) keep_unit(i, cp[j]);
            }
        }
        op += p.body;
        if (tid < p.ovl) keep_unit(p.body + tid, tail);
        if (tid + T < p.ovl) keep_unit(p.body + tid + T, tail2);
        __syncthreads();                                 // the cross-fade has read the old tail and the old window
        if (tid < p.ovl) {
#pragma unroll
            for (int c = 0; c < CH; c++) mid[tid * CH + c] = tail.x[c];
        }
        if (tid + T < p.ovl) {
#pragma unroll
            for (int c = 0; c < CH; c++) mid[(tid + T) * CH + c] = tail2.x[c];
        }
        if (more) {
            if (kReuse) {
                // the units of the next window that the copy did not pass: in front of it (a large offset) or behind it
                const int copied = p.body + p.ovl;
                for (int u = tid; u < nunits; u += T) {
                    const int i = u - un0;
                    if (i < 0 || i >= copied) put_unit(wnext, u, ld_frame<CH>(in, sbase, ip_n + u));
                }
            } else {
                if (!kAhead) window_load(ip_n);
                window_store(wnext);
            }
        }
        wcur = wnext;
        skip = skip_n;
        ip = ip_n;
        __syncthreads();
    }
    if (mid_state) {
        for (int i = tid; i < p.ovl * CH; i += T) mid_state[s * (long long)(p.ovl * CH) + i] = mid[i];
    }
}

static int td_row_stride(int ch, int ovl, int nc)
{
    const int P = 4 / ch, R = nc * P;
    const int ng = ovl * ch / 4;
    // furthest column any thread touches: candidate base up to 1024 plus the sliding steps
    const int max_unit = 1024 + P * (ng + 2 * nc) + P;
    int S = max_unit / R + 2;
    if (ch == 1) {
        while (S % 64 != 16) S++;                        // quarter-waves on disjoint bank ranges
    }
    return S;
}

int st_launch_td(nae_ctx* ctx, const StCfg& c, const StView& in, const TdRange& r, const StOut& out, size_t n_streams,
                 float* mid_state, int32_t* offs_dbg, long long offs_stride)
{
    if (r.nseq <= 0 || n_streams == 0) return NAE_OK;
    if (c.seekl > 1024 || c.seekl < 1 || c.ovl % 8 != 0 || c.ovl < 16 || c.ovl > 512 || c.seekl + c.ovl > 1344)
        return nae_fail(ctx, NAE_ERR_INVALID, "WSOLA geometry");
    // several workgroups per CU: 4 candidates per thread (fewest instructions per candidate).  Up to about two
    // workgroups per CU the search is latency-bound and 2 candidates per thread (twice the waves) is 1.2-1.6x faster;
    // 1 per thread loses everywhere (measured: tools/td_sweep.sh) and stays only as a test shape.
    int nc = n_streams >= 640 ? 4 : 2;
    // five candidates per thread on three waves (960 slots) where that wastes fewer slots than four on up to four waves (256 per wave); stereo only
    // (the mono window's bank layout is built for power-of-two row counts)
    if (ctx->dbg_td_nc == 1 || ctx->dbg_td_nc == 2 || ctx->dbg_td_nc == 4) nc = ctx->dbg_td_nc;
    TdParams p;
    p.ovl = c.ovl; p.seekl = c.seekl; p.body = c.body; p.first_skip = c.first_skip;
    p.nominal_skip = c.nominal_skip;
    p.ip0 = r.ip0; p.op0 = r.op0; p.nseq = r.nseq; p.skip0 = r.skip0; p.begin0 = r.begin0; p.out_limit = r.out_limit;
    p.S = td_row_stride(c.ch, c.ovl, nc);
    const int P = 4 / c.ch, rows = nc * P + P - 1;
    const size_t lds = ((size_t)(nc >= 4 ? 2 : 1) * rows * p.S * c.ch + (size_t)c.ovl * c.ch + 16 + 2 * (size_t)c.ovl) * sizeof(float);
    if (lds > 60 * 1024) return nae_fail(ctx, NAE_ERR_INVALID, "WSOLA window does not fit LDS");
    const dim3 grid((unsigned)n_streams);
#define NAE_TD(CHN, NCN) NAE_KLAUNCH(ctx, "st_td_kernel", (st_td_kernel<CHN, NCN>), grid, dim3(td_threads(NCN)), lds, ctx->stream, \
                                     dview(in, CHN), p, dout(out, CHN), mid_state, offs_dbg, offs_stride)
    if (c.ch == 2 && nc == 4 && p.S == 180) NAE_KLAUNCH(ctx, "st_td_kernel", (st_td_kernel<2, 4, 180>), grid, dim3(256), lds, ctx->stream, dview(in, 2), p, dout(out, 2), mid_state, offs_dbg, offs_stride);
    else if (c.ch == 2) { if (nc == 4) NAE_TD(2, 4); else if (nc == 2) NAE_TD(2, 2); else NAE_TD(2, 1); }
    else { if (nc == 4) NAE_TD(1, 4); else if (nc == 2) NAE_TD(1, 2); else NAE_TD(1, 1); }
#undef NAE_TD
    return nae_check(ctx, hipGetLastError(), "st_td_kernel");
}

// ------------------------------------------------------------------ AA: 64-tap FIR
constexpr int kAaTile = 1024;                 // outputs per workgroup (4 per thread)
constexpr int kAaRow = (kAaTile + kAaLen) / 4 + 1;
struct AaParams {
    float h[kAaLen];
    long long j0, j1;
};
// the cubic transposer fused behind the filter (block mode, the two stages adjacent): a workgroup filters 1024 outputs,
// keeps them in LDS and emits every cubic output whose 4 taps lie inside them; tiles therefore advance by 1021 and the
// host tells each tile its first cubic output (tile_n, tiles + 1 entries).  The filtered signal never touches HBM.
struct CuFuse {
    const long long* pos;
    const float* fr;
    const int* tile_n;
    long long n_limit;       // cubic outputs at or beyond this index are not wanted
};
constexpr int kAaTileFused = kAaTile - 3;

// kCubicIn: the cubic transposer fused IN FRONT of the filter (rate < 1): the tile's input frames are cubic outputs
// cu.pos / cu.fr [jt + u], evaluated from the raw signal while staging (cu.n_limit = number of cubic outputs that
// exist; beyond it the filter reads zeros); the transposed signal never touches HBM.
// (__launch_bounds__(256, 8): left to itself the compiler spent 85 VGPRs — 5 waves per SIMD — on a body that needs 49; at 8 waves
// per SIMD the vector unit issues an instruction per 1.26 cycles (measured, profiles/r02_valu_issue.md) instead of ~1.8 (interpolated
// between the measured 4- and 6-wave rows): 4.58 -> 3.95 ms at C5)
template <int CH, bool kCubic, bool kCubicIn = false>
__global__ __launch_bounds__(256, 8) void st_aa_kernel(DView in, AaParams p, DOut out, CuFuse cu)
{
    __shared__ float tile[4 * kAaRow * CH];
    __shared__ __attribute__((aligned(16))) float hs[kAaLen];
    const int tid = threadIdx.x;
    const long long s = blockIdx.y;
    const long long jt = p.j0 + (long long)blockIdx.x * (kCubic ? kAaTileFused : kAaTile);
    const float* sbase = in.v.base + s * in.v.ss;
    float* obase = out.o.base + s * out.o.ss;
    for (int u = tid; u < kAaTile + kAaLen; u += 256) {
        Frame<CH> x;
        if (kCubicIn) {
            const long long n = jt + u;
#pragma unroll
            for (int c = 0; c < CH; c++) x.x[c] = 0.0f;
            if (n < cu.n_limit) {
                const long long a = cu.pos[n];
                const float x2 = cu.fr[n], x1 = x2 * x2, x0 = x1 * x2, x3 = 1.0f;
                const float y0 = ((-0.5f * x0 + 1.0f * x1) + -0.5f * x2) + 0.0f * x3;
                const float y1 = ((1.5f * x0 + -2.5f * x1) + 0.0f * x2) + 1.0f * x3;
                const float y2 = ((-1.5f * x0 + 2.0f * x1) + 0.5f * x2) + 0.0f * x3;
                const float y3 = ((0.5f * x0 + -0.5f * x1) + 0.0f * x2) + 0.0f * x3;
                const Frame<CH> p0 = ld_frame<CH>(in, sbase, a), p1 = ld_frame<CH>(in, sbase, a + 1);
                const Frame<CH> p2 = ld_frame<CH>(in, sbase, a + 2), p3 = ld_frame<CH>(in, sbase, a + 3);
#pragma unroll
                for (int c = 0; c < CH; c++) x.x[c] = ((y0 * p0.x[c] + y1 * p1.x[c]) + y2 * p2.x[c]) + y3 * p3.x[c];
            }
        } else {
            x = ld_frame<CH>(in, sbase, jt + u);
        }
#pragma unroll
        for (int c = 0; c < CH; c++) tile[((u & 3) * kAaRow + (u >> 2)) * CH + c] = x.x[c];
    }
    if (tid < kAaLen) hs[tid] = p.h[tid];
    __syncthreads();
    const long long j = jt + 4 * tid;
    if (!kCubic && j >= p.j1) return;
    // 8 taps per trip: frames m = kb .. kb+10 feed outputs i = 0..3 with tap k = m - i.  Every accumulator still sees
    // its taps in increasing order (the order is observable: results are compared bit for bit).
    float ev[4][CH], od[4][CH];
    float acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc[i] = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; c++) ev[i][c] = od[i][c] = 0.0f;
    }
#pragma unroll 1
    for (int kb = 0; kb < kAaLen; kb += 8) {
        float x[11][CH], h[8];
#pragma unroll
        for (int t = 0; t < 11; t++) {
            if (CH == 2) {
                // one 8-byte read per stereo frame, kept from being paired: a ds_read2_b64 moves its 16 bytes per lane in twice the LDS time of two ds_read_b64
                typedef __attribute__((address_space(3))) const volatile unsigned long long lds_u64;
                const unsigned long long u = *(lds_u64*)(tile + ((t & 3) * kAaRow + tid + (kb >> 2) + (t >> 2)) * 2);
                x[t][0] = __uint_as_float((unsigned)u);
                x[t][CH - 1] = __uint_as_float((unsigned)(u >> 32));
            } else {
#pragma unroll
                for (int c = 0; c < CH; c++) x[t][c] = tile[((t & 3) * kAaRow + tid + (kb >> 2) + (t >> 2)) * CH + c];
            }
        }
        {
            const float4 h0 = *reinterpret_cast<const float4*>(hs + kb), h1 = *reinterpret_cast<const float4*>(hs + kb + 4);
            h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
        }
#pragma unroll
        for (int t = 0; t < 11; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = t - i;
                if (k >= 0 && k < 8) {
                    if (CH == 2) {
                        // SSE stereo order: even and odd taps are summed separately
#pragma unroll
                        for (int c = 0; c < CH; c++) {
                            if ((k & 1) == 0) ev[i][c] = ev[i][c] + x[t][c] * h[k];
                            else od[i][c] = od[i][c] + x[t][c] * h[k];
                        }
                    } else {
                        acc[i] = acc[i] + x[t][0] * h[k];        // generic mono order: one float accumulator, taps in order
                    }
                }
            }
    }
    Frame<CH> y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (CH == 2) {
#pragma unroll
            for (int c = 0; c < CH; c++) y[i].x[c] = od[i][c] + ev[i][c];
        } else {
            y[i].x[0] = acc[i];
        }
    }
    if (!kCubic) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (j + i < p.j1) st_frame<CH>(out, obase, j + i, y[i]);
        return;
    }
    // ---- fused cubic stage: the 1024 filtered frames go to LDS in natural order (the input tile is dead by now)
    __syncthreads();
    float* filt = tile;                                   // [1024][CH]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < CH; c++) filt[(4 * tid + i) * CH + c] = y[i].x[c];
    __syncthreads();
    const long long n_lo = cu.tile_n[blockIdx.x];
    long long n_hi = cu.tile_n[blockIdx.x + 1];
    if (n_hi > cu.n_limit) n_hi = cu.n_limit;
    for (long long n = n_lo + tid; n < n_hi; n += 256) {
        const int a = (int)(cu.pos[n] - jt);              // 0 <= a and a + 3 < 1024 by construction of tile_n
        const float x2 = cu.fr[n], x1 = x2 * x2, x0 = x1 * x2, x3 = 1.0f;
        const float y0 = ((-0.5f * x0 + 1.0f * x1) + -0.5f * x2) + 0.0f * x3;
        const float y1 = ((1.5f * x0 + -2.5f * x1) + 0.0f * x2) + 1.0f * x3;
        const float y2 = ((-1.5f * x0 + 2.0f * x1) + 0.5f * x2) + 0.0f * x3;
        const float y3 = ((0.5f * x0 + -0.5f * x1) + 0.0f * x2) + 0.0f * x3;
        Frame<CH> o;
#pragma unroll
        for (int c = 0; c < CH; c++)
            o.x[c] = ((y0 * filt[a * CH + c] + y1 * filt[(a + 1) * CH + c]) + y2 * filt[(a + 2) * CH + c]) + y3 * filt[(a + 3) * CH + c];
        st_frame<CH>(out, obase, n, o);
    }
}

int st_launch_aa(nae_ctx* ctx, const StCfg& c, const StView& in, long long j0, long long j1, const StOut& out,
                 size_t n_streams)
{
    if (j1 <= j0 || n_streams == 0) return NAE_OK;
    AaParams p;
    for (int k = 0; k < kAaLen; k++) p.h[k] = c.aa[k];
    p.j0 = j0; p.j1 = j1;
    const unsigned tiles = (unsigned)((j1 - j0 + kAaTile - 1) / kAaTile);
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        StView vin = in;
        vin.base += (long long)s0 * in.ss;
        StOut vout = out;
        vout.base += (long long)s0 * out.ss;
        if (c.ch == 2)
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<2, false>), dim3(tiles, ns), dim3(256), 0, ctx->stream, dview(vin, 2), p,
                        dout(vout, 2), CuFuse{});
        else
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<1, false>), dim3(tiles, ns), dim3(256), 0, ctx->stream, dview(vin, 1), p,
                        dout(vout, 1), CuFuse{});
    }
    return nae_check(ctx, hipGetLastError(), "st_aa_kernel");
}

// filter + cubic in one launch: cubic outputs [0, n_limit) of every stream from filter input `in`; d_tile_n has tiles + 1
// entries (st_tile_starts)
int st_launch_aa_cu(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract, const int* d_tile_n,
                    size_t tiles, long long n_limit, const StOut& out, size_t n_streams)
{
    if (tiles == 0 || n_limit <= 0 || n_streams == 0) return NAE_OK;
    AaParams p;
    for (int k = 0; k < kAaLen; k++) p.h[k] = c.aa[k];
    p.j0 = 0; p.j1 = 0;
    const CuFuse cu{d_pos, d_fract, d_tile_n, n_limit};
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        StView vin = in;
        vin.base += (long long)s0 * in.ss;
        StOut vout = out;
        vout.base += (long long)s0 * out.ss;
        if (c.ch == 2)
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<2, true>), dim3((unsigned)tiles, ns), dim3(256), 0, ctx->stream,
                        dview(vin, 2), p, dout(vout, 2), cu);
        else
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<1, true>), dim3((unsigned)tiles, ns), dim3(256), 0, ctx->stream,
                        dview(vin, 1), p, dout(vout, 1), cu);
    }
    return nae_check(ctx, hipGetLastError(), "st_aa_cu_kernel");
}

// cubic + filter in one launch (rate < 1): filter outputs [0, j1) of every stream; the filter's input frame n is the
// cubic output (d_pos[n], d_fract[n]) of the raw signal `in`, n < n_cu
int st_launch_cu_aa(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract, long long n_cu,
                    long long j1, const StOut& out, size_t n_streams)
{
    if (j1 <= 0 || n_streams == 0) return NAE_OK;
    AaParams p;
    for (int k = 0; k < kAaLen; k++) p.h[k] = c.aa[k];
    p.j0 = 0; p.j1 = j1;
    const CuFuse cu{d_pos, d_fract, nullptr, n_cu};
    const unsigned tiles = (unsigned)((j1 + kAaTile - 1) / kAaTile);
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        StView vin = in;
        vin.base += (long long)s0 * in.ss;
        StOut vout = out;
        vout.base += (long long)s0 * out.ss;
        if (c.ch == 2)
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<2, false, true>), dim3(tiles, ns), dim3(256), 0, ctx->stream, dview(vin, 2),
                        p, dout(vout, 2), cu);
        else
            NAE_KLAUNCH(ctx, "st_aa_kernel", (st_aa_kernel<1, false, true>), dim3(tiles, ns), dim3(256), 0, ctx->stream, dview(vin, 1),
                        p, dout(vout, 1), cu);
    }
    return nae_check(ctx, hipGetLastError(), "st_cu_aa_kernel");
}

// (position, fraction) of `count` consecutive cubic outputs, continued on the device from the state (pos0, fract0):
// the library's own recurrence (fract += rate; whole = int(fract); fract -= whole) in IEEE double, so the entries
// equal the host's bit for bit.  Sequential by nature; used by the streaming handle, whose puts add a few thousand
// outputs at a time, so that a put never has to wait for a host-to-device copy.
__global__ void st_cu_table_kernel(long long pos0, double fract0, double rate, long long count, long long* __restrict__ pos,
                                   float* __restrict__ fr)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    long long p = pos0;
    double f = fract0;
    for (long long i = 0; i < count; i++) {
        pos[i] = p;
        fr[i] = (float)f;
        f += rate;
        const int whole = (int)f;
        f -= (double)whole;
        p += whole;
    }
}

int st_launch_cu_table(nae_ctx* ctx, long long pos0, double fract0, double rate, long long count, long long* d_pos, float* d_fract)
{
    if (count <= 0) return NAE_OK;
    NAE_KLAUNCH(ctx, "st_cu_table_kernel", st_cu_table_kernel, dim3(1), dim3(64), 0, ctx->stream, pos0, fract0, rate, count, d_pos, d_fract);
    return nae_check(ctx, hipGetLastError(), "st_cu_table_kernel");
}

// first cubic output of every fused tile: tile t covers filter outputs [1021 t, 1021 t + 1024)
void st_tile_starts(const CuTable& tab, long long n_limit, std::vector<int>& tile_n)
{
    tile_n.clear();
    if (n_limit <= 0) return;
    const long long last_pos = tab.pos[(size_t)n_limit - 1];
    const size_t tiles = (size_t)(last_pos / kAaTileFused) + 1;
    size_t n = 0;
    for (size_t t = 0; t <= tiles; t++) {
        const long long lo = (long long)t * kAaTileFused;
        while (n < (size_t)n_limit && tab.pos[n] < lo) n++;
        tile_n.push_back((int)n);
    }
}

// ------------------------------------------------------------------ CU: cubic transposer
// One thread = one output index for kCuGroup streams: the (position, fraction) entry and the 4 weights depend on the
// index only, so the table — as large as the output itself — is read once per group instead of once per stream
// (the kernel is bound by memory traffic: table + 4 input frames + 1 output frame per stream-output).
constexpr int kCuGroup = 4;

template <int CH>
__global__ __launch_bounds__(256) void st_cu_kernel(DView in, const long long* __restrict__ pos, const float* __restrict__ fr,
                                                    long long tab_origin, long long n0, long long n1, DOut out, long long n_streams,
                                                    unsigned blocks)
{
    // output index fastest over the grid: each stream's reads and writes stay contiguous
    const long long s0 = (long long)(blockIdx.x / blocks) * kCuGroup;
    const long long n = n0 + (long long)(blockIdx.x % blocks) * 256 + threadIdx.x;
    if (n >= n1) return;
    const long long a = pos[n - tab_origin];
    const float x2 = fr[n - tab_origin], x1 = x2 * x2, x0 = x1 * x2, x3 = 1.0f;
    const float y0 = ((-0.5f * x0 + 1.0f * x1) + -0.5f * x2) + 0.0f * x3;
    const float y1 = ((1.5f * x0 + -2.5f * x1) + 0.0f * x2) + 1.0f * x3;
    const float y2 = ((-1.5f * x0 + 2.0f * x1) + 0.5f * x2) + 0.0f * x3;
    const float y3 = ((0.5f * x0 + -0.5f * x1) + 0.0f * x2) + 0.0f * x3;
    Frame<CH> p0[kCuGroup], p1[kCuGroup], p2[kCuGroup], p3[kCuGroup];
#pragma unroll
    for (int k = 0; k < kCuGroup; k++) {
        if (s0 + k < n_streams) {
            const float* sbase = in.v.base + (s0 + k) * in.v.ss;
            p0[k] = ld_frame<CH>(in, sbase, a);
            p1[k] = ld_frame<CH>(in, sbase, a + 1);
            p2[k] = ld_frame<CH>(in, sbase, a + 2);
            p3[k] = ld_frame<CH>(in, sbase, a + 3);
        }
    }
#pragma unroll
    for (int k = 0; k < kCuGroup; k++) {
        if (s0 + k < n_streams) {
            Frame<CH> y;
#pragma unroll
            for (int c = 0; c < CH; c++) y.x[c] = ((y0 * p0[k].x[c] + y1 * p1[k].x[c]) + y2 * p2[k].x[c]) + y3 * p3[k].x[c];
            st_frame<CH>(out, out.o.base + (s0 + k) * out.o.ss, n, y);
        }
    }
}

int st_launch_cu(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract,
                 long long tab_origin, long long n0, long long n1, const StOut& out, size_t n_streams)
{
    if (n1 <= n0 || n_streams == 0) return NAE_OK;
    const long long blocks = (n1 - n0 + 255) / 256;
    if (blocks > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "cubic transposer: range too long for one launch");
    // one launch covers at most 2^31 - 1 workgroups: chunk the stream groups if a batch is larger than that
    const size_t max_groups = (size_t)(0x7fffffffll / blocks);
    const size_t max_ns = max_groups * kCuGroup;
    for (size_t s0 = 0; s0 < n_streams; s0 += max_ns) {
        const size_t ns = (n_streams - s0 < max_ns) ? n_streams - s0 : max_ns;
        StView vin = in;
        vin.base += (long long)s0 * in.ss;
        StOut vout = out;
        vout.base += (long long)s0 * out.ss;
        const unsigned grid = (unsigned)(blocks * ((ns + kCuGroup - 1) / kCuGroup));
        if (c.ch == 2)
            NAE_KLAUNCH(ctx, "st_cu_kernel", (st_cu_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, dview(vin, 2), d_pos, d_fract,
                        tab_origin, n0, n1, dout(vout, 2), (long long)ns, (unsigned)blocks);
        else
            NAE_KLAUNCH(ctx, "st_cu_kernel", (st_cu_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, dview(vin, 1), d_pos, d_fract,
                        tab_origin, n0, n1, dout(vout, 1), (long long)ns, (unsigned)blocks);
    }
    return nae_check(ctx, hipGetLastError(), "st_cu_kernel");
}

} // namespace nae
