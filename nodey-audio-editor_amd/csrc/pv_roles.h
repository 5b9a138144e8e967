// pv_roles.h — the role bodies of the phase vocoder's wave pipeline (kernels_pvpipe.hip).  A frame passes through four waves — R1 analysis FFT,
// R2a / R2b phases of one half of the bins each, R3 synthesis — and the two schedules of that file (two barriers per step with one buffer per
// hand-off; one barrier per step with two) differ only in WHEN a role runs and WHICH buffer generation it touches.  What a role computes is here,
// once: the arithmetic is the canonical one of DESIGN.md §3, so every schedule gives the same bits (tests/test_gpu_stft.py:
// test_k7_pipeline_modes_agree_bit_for_bit).
#pragma once
#include "stft_common.h"

namespace nae {

constexpr int kYCf = 520;                                   // Y[0..512] natural order
constexpr int kOlaQuarter = 256;                            // floats

// every LDS operation of this wave has completed, then the workgroup barrier (vector-memory operations stay in flight:
// the frame prefetch of R1 and the block stores of R3 must not be drained at every barrier)
__device__ __forceinline__ void pipe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the lane index as a value the optimiser cannot see through: addresses derived from it are recomputed where they are used
// (one to four instructions each) instead of being hoisted out of the frame loop, where each would pin a VGPR of the 64.
// Only where that matters: R1 and R3 of the 64-VGPR build; the phase roles have registers to spare (their hoisted addresses
// stay below the kernel's maximum: -1.7 % kernel time), and the 128-VGPR builds hide nothing
template <bool kHide = true>
__device__ __forceinline__ int pipe_lane(int lane)
{
    if (kHide) asm volatile("" : "+v"(lane));
    return lane;
}

// exact phase increment of one hop for bin k: adv + round(dw * R / 2^24)
__device__ __forceinline__ uint32_t pipe_inc(uint32_t qa, uint32_t qp, unsigned k, unsigned d, unsigned R)
{
    const uint32_t e = ((k * d) & (NAE_FFT_N - 1)) << 22;
    const int32_t dw = (int32_t)(qa - qp - e);
    const uint32_t adv = ((k * NAE_HOP) & (NAE_FFT_N - 1)) << 22;
    const long long scaled = ((long long)dw * (long long)(int32_t)R + (1ll << (NAE_R_FRAC_BITS - 1))) >> NAE_R_FRAC_BITS;
    return adv + (uint32_t)scaled;
}

// synthesis bin X e^{i (qs - qa)} (tolerance path: v_sin / v_cos take turns)
__device__ __forceinline__ cf pipe_rotate(cf x, uint32_t qs, uint32_t qa)
{
    const float ph = (float)(int32_t)(qs - qa) * (1.0f / 4294967296.0f);
    const float cs = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
    return cf{__builtin_fmaf(x.x, cs, -(x.y * sn)), __builtin_fmaf(x.x, sn, x.y * cs)};
}

// issue priority of a wave when two workgroups share a CU (kernels_pvpipe.hip, "Issue priority"): the workgroups take turns at the higher level in
// time slices of 2^18 shader cycles, R3 sits one level above its workgroup's
constexpr int kPrioSliceBit = 18;
__device__ __forceinline__ void pipe_prio(int slot, int role, unsigned long long now)
{
    const int turn = (int)((now >> kPrioSliceBit) & 1);
    const int lvl = (((turn + slot) & 1) ? 2 : 0) + (role == 3 ? 1 : 0);                // wave-uniform
    if (lvl == 0) __builtin_amdgcn_s_setprio(0);
    else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// which of the two workgroups of its CU this one is (0 / 1, by arrival; thread 0 of the workgroup calls it): the parity of a counter per physical CU
__device__ __forceinline__ int pipe_cu_arrival(unsigned* arrivals)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_ID: CU 8-11, SE 13-14
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;     // XCC_ID
    const unsigned key = (xcc * 4 + ((hw >> 13) & 3u)) * 16 + ((hw >> 8) & 15u);
    return (int)(atomicAdd(&arrivals[key], 1u) & 1u);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The work item of a unit: one (stream-channel, tile).  Stereo: units 2i, 2i+1 are the two channels of one (stream, tile), so their block stores of
// an interleaved destination happen in the same step and merge in L2 (or into one dense piece); mono: consecutive (stream, tile) items.
struct PipeItem {
    long long sc, s_idx;       // stream-channel, stream
    int c, tile;               // channel, time tile
    long long b0, b_end;       // output blocks [b0, b_end) == the tile's first frame .. one past its last
    long long f_first;         // first frame analysed: b0 - 1 only primes the previous phase
    int n;                     // frames f_first .. f_first + n - 1 feed the tile's blocks (frames up to b_end + 2)
};

__device__ __forceinline__ bool pipe_item(const PvParams& p, long long ug, long long n_sc, PipeItem& it)
{
    if (p.ch == 2) {
        const long long pair = ug >> 1;                          // (stream, tile), tile fastest
        it.sc = 2 * (pair / p.n_tiles) + (ug & 1);
        it.tile = (int)(pair % p.n_tiles);
    } else {
        it.sc = ug / p.n_tiles;
        it.tile = (int)(ug % p.n_tiles);
    }
    if (it.sc >= n_sc) return false;                             // (a terminated wave no longer counts at s_barrier)
    it.s_idx = it.sc / p.ch;
    it.c = (int)(it.sc % p.ch);
    it.b0 = p.f_origin + (long long)it.tile * p.tile;
    it.b_end = it.b0 + p.tile;
    if (it.b_end > p.f_stop) it.b_end = p.f_stop;
    long long f_end = it.b_end + 3;                              // frames b0 .. b_end+2 feed blocks b0 .. b_end-1
    if (f_end > p.frames) f_end = p.frames;
    it.f_first = (it.b0 > 0 ? it.b0 - 1 : 0);
    it.n = (int)(f_end - it.f_first);
    return it.n > 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// R1: Hann window and pass A of the forward FFT — register-only, so a schedule may run it while another wave still reads this wave's scratch.
// kRich (one workgroup per CU: 128 VGPRs per wave): window and both twiddle sets live in registers; else they are requested from the LDS tables in ONE
// round trip (the accesses are volatile, so the compiler keeps them where they are written; one read per product would cost a round trip each).
struct FftRegs { cf w[8], ta[7], tb[7]; };                  // window of samples 2 (lane + 64 r), +1; pass-A twiddles W512^(lane q); pass-B twiddles

__device__ __forceinline__ void load_fft_regs_lds(FftRegs& R, const float* hann, const cf* twa, const cf* w64, int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) R.w[r] = lds_ld(reinterpret_cast<const cf*>(hann) + lane + 64 * r);
#pragma unroll
    for (int q = 0; q < 7; q++) { R.ta[q] = lds_ld(twa + lane + 64 * q); R.tb[q] = lds_ld(w64 + 8 * (lane & 7) + q + 1); }
}
__device__ __forceinline__ void load_fft_regs_global(FftRegs& R, const Tables& tb, int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) R.w[r] = reinterpret_cast<const cf*>(tb.hann)[lane + 64 * r];
#pragma unroll
    for (int q = 0; q < 7; q++) { R.ta[q] = tb.w512[lane * (q + 1)]; R.tb[q] = tb.w512[8 * (lane & 7) * (q + 1)]; }
}

template <bool kRich>
__device__ __forceinline__ void r1_window_pass_a(cf (&va)[8], const cf (&raw)[8], const FftRegs& R, const float* hann, const cf* twa, int lane)
{
    if (kRich) {
#pragma unroll
        for (int r = 0; r < 8; r++) va[r] = cf{raw[r].x * R.w[r].x, raw[r].y * R.w[r].y};
        fft512_pad_a_tw(va, R.ta);
    } else {
        const int la = pipe_lane<true>(lane);
        const cf* hw = reinterpret_cast<const cf*>(hann) + la;
        const cf* ta = twa + la;
        cf w[8], tw[7];
#pragma unroll
        for (int r = 0; r < 8; r++) w[r] = lds_ld(hw + 64 * r);
#pragma unroll
        for (int q = 0; q < 7; q++) tw[q] = lds_ld(ta + 64 * q);
#pragma unroll
        for (int r = 0; r < 8; r++) va[r] = cf{raw[r].x * w[r].x, raw[r].y * w[r].y};
        fft512_pad_a_tw(va, tw);
    }
}

// the finished spectrum in natural order where R2 reads it (entry 512 = Z[0]: the mirror of bin 0 is read like any other)
__device__ __forceinline__ void r1_store_z(cf* Z, const FftLds& L, const cf (&va)[8], int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, va[r]);
    if (lane == 0) Z[512] = va[0];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// R2a / R2b: the lane's bins.  Bins in mirror pairs: a lane owns k = lane + 64 (2 h + i) and 512 - k, i = 0, 1 (items 2 i and 2 i + 1) — 0..127 and
// 385..512 for h = 0, 128..255 and 257..384 for h = 1.  One pair of reads (A = Z[k], B = Z[512 - k]) gives both spectra: the mirror's E and O are
// (Ex, -Ey) and (-Ox, Oy) — exact negations and commuted sums of the canonical formula, so every phase keeps its bits — and both rotated bins of a
// pair meet in one lane, where the c2r pre-twiddle of R3's FFT input needs them: what goes to R3 is that input, not Y.
// 513 bins are 512 items and one more.  The odd one is bin 512, whose phase is just the sign of a real number (no atan2): in lane 0 of h = 0 the pair
// would be (0, 512); there item 1 carries the self-mirrored bin 256 instead (its own A = B = Z[256]), and bin 512 rides along in that wave as a
// fifth, cheap item (sign, increment, one cosine).
struct PhaseLane {
    int h, k0, km0;            // half; items 0 / 2: k0, k0 + 64; items 1 / 3: km0 = 512 - k0 (256 in lane 0 of h = 0), 448 - k0
    bool dc;                   // the lane of bins 0, 256 and 512
    cf tk[2], tm[2], tms;      // split twiddles of the items' bins; tms: of the pair's mirror (differs from tm[0] in lane 0 of h = 0: bin 512)

    __device__ __forceinline__ void init(const Tables& tb, int half, int lane)
    {
        h = half;
        k0 = lane + 128 * h;
        dc = (h == 0) && (lane == 0);
        km0 = dc ? 256 : 512 - k0;
#pragma unroll
        for (int i = 0; i < 2; i++) { tk[i] = tb.t1024[k0 + 64 * i]; tm[i] = tb.t1024[i == 0 ? km0 : 448 - k0]; }
        tms = tb.t1024[512 - k0];
    }
    // the lane's five records of a [520] phase array ([4] = bin 512: every lane reads it, the dc lane writes it)
    __device__ __forceinline__ void read5(const uint32_t* q, uint32_t (&v)[5]) const
    {
        v[0] = q[k0]; v[1] = q[km0]; v[2] = q[k0 + 64]; v[3] = q[448 - k0];
        v[4] = q[512];
    }
    __device__ __forceinline__ void write5(uint32_t* q, const uint32_t (&v)[5]) const
    {
        q[k0] = v[0]; q[km0] = v[1]; q[k0 + 64] = v[2]; q[448 - k0] = v[3];
        if (dc) q[512] = v[4];
    }
    // base phase of the tile (pass 2's record, or zero for a lone tile with nothing carried in)
    __device__ __forceinline__ void load_base(uint32_t (&qs)[5], const uint32_t* base_phase, const PvParams& p, const PipeItem& it) const
    {
        if (p.base_zero) {
#pragma unroll
            for (int q = 0; q < 5; q++) qs[q] = 0;
        } else {
            read5(base_phase + (it.sc * p.phase_tiles + (long long)it.tile * p.phase_step) * kT1024Pad, qs);
        }
    }
    // A = Z[k], B = Z[512 - k] of both pairs (+ Z[256] in half 0)
    __device__ __forceinline__ void read_z(const cf* Z, cf (&A)[2], cf (&B)[2], cf& z256) const
    {
        const cf* Zk = Z + k0;
        const cf* Zm = Z + 512 - k0;
#pragma unroll
        for (int i = 0; i < 2; i++) { A[i] = lds_ld(Zk + 64 * i); B[i] = lds_ld(Zm - 64 * i); }
        if (h == 0) z256 = lds_ld(Z + 256);
    }
    // r2c split -> 2 X (phases are scale-invariant; the factor is undone in R3's output gain: a factor 2 is exact in every product on the way)
    __device__ __forceinline__ void split(const cf (&A)[2], const cf (&B)[2], cf z256, cf (&va)[5]) const
    {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const cf E = cf{A[i].x + B[i].x, A[i].y - B[i].y};
            const cf O = cf{A[i].x - B[i].x, A[i].y + B[i].y};
            const cf P = cmul_tw(O, tk[i]);
            va[2 * i] = cf{E.x + P.y, E.y - P.x};
            const cf Em = cf{E.x, -E.y};
            const cf Om = cf{-O.x, O.y};
            const cf Pm = cmul_tw(Om, i == 0 ? tms : tm[i]);
            va[2 * i + 1] = cf{Em.x + Pm.y, Em.y - Pm.x};
        }
        va[4] = cf{0.0f, 0.0f};
        if (h == 0) {
            // lane 0: item 1 so far is bin 512 (from Z[0] alone) -> the fifth item; bin 256 takes its place
            va[4] = va[1];
            const cf E = cf{z256.x + z256.x, z256.y - z256.y};
            const cf O = cf{z256.x - z256.x, z256.y + z256.y};
            const cf P = cmul_tw(O, tm[0]);
            if (dc) va[1] = cf{E.x + P.y, E.y - P.x};
        }
    }
    __device__ __forceinline__ void phases(const cf (&va)[5], uint32_t (&qa)[5]) const
    {
#pragma unroll
        for (int q = 0; q < 4; q++) qa[q] = atan2_q32(va[q].y, va[q].x);
        // bin N/2 of a real signal is real: its phase is 0 or 1/2 turn by the sign of the real part
        if (h == 0) qa[4] = (va[4].x < 0.0f) ? 0x80000000u : 0u;
    }
    // phase increment of all items of this lane (exact integers)
    __device__ __forceinline__ void inc_items(const uint32_t (&qa)[5], const uint32_t (&qv)[5], unsigned d, unsigned R, uint32_t (&inc)[5]) const
    {
        inc[0] = pipe_inc(qa[0], qv[0], (unsigned)k0, d, R);
        inc[1] = pipe_inc(qa[1], qv[1], (unsigned)km0, d, R);
        inc[2] = pipe_inc(qa[2], qv[2], (unsigned)(k0 + 64), d, R);
        inc[3] = pipe_inc(qa[3], qv[3], (unsigned)(448 - k0), d, R);
        inc[4] = (h == 0) ? pipe_inc(qa[4], qv[4], 512u, d, R) : 0u;
    }
    // increment of frame f against its predecessor's analysis phases (frame 0: its analysis phase itself)
    __device__ __forceinline__ void inc_of_frame(const PvParams& p, long long f, const uint32_t (&qa)[5], const uint32_t (&qprev)[5], uint32_t (&inc)[5]) const
    {
        if (f == 0) {
#pragma unroll
            for (int q = 0; q < 5; q++) inc[q] = qa[q];
        } else {
            const unsigned d = (unsigned)(frame_start(p, f) - frame_start(p, f - 1));
            const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
            inc_items(qa, qprev, d, R, inc);
        }
    }
    // a continued stream carries the synthesis phase behind its segment's last frame on (nae_stream.hip)
    __device__ __forceinline__ void carry_store(const PvParams& p, long long sc, const uint32_t (&q)[5]) const { write5(p.carry_out + sc * kT1024Pad, q); }
    // rotation by the phase difference and the in-lane c2r pre-twiddle: R3's FFT input Zin[k], Zin[512 - k]
    // (conjugated, inverse = conj(FFT(conj Z)) / 512; 2E, 2D: see kSynthGain)
    __device__ __forceinline__ void synth_items(cf* Y, const cf (&x)[5], const uint32_t (&qsv)[5], const uint32_t (&qav)[5]) const
    {
        const int kk = k0, km = km0;
        cf* Yk = Y + kk;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            cf yk = pipe_rotate(x[2 * i], qsv[2 * i], qav[2 * i]);
            const cf ym = pipe_rotate(x[2 * i + 1], qsv[2 * i + 1], qav[2 * i + 1]);
            cf a = ym, b = yk;                            // index 512 - k: the roles of the two bins swap
            cf mk = ym;                                   // the partner of bin k
            if (i == 0 && h == 0) {
                // lane 0: bin 0 pairs with bin 512 (both real: c2r ignores their imaginary parts), bin 256 with itself
                const float y512 = pipe_rotate(x[4], qsv[4], qav[4]).x;
                if (dc) { yk.y = 0.0f; mk = cf{y512, 0.0f}; b = ym; }
            }
            const cf E{yk.x + mk.x, yk.y - mk.y};
            const cf D{yk.x - mk.x, yk.y + mk.y};
            const cf Q{__builtin_fmaf(tk[i].x, D.x, tk[i].y * D.y), __builtin_fmaf(tk[i].x, D.y, -(tk[i].y * D.x))};
            lds_st(Yk + 64 * i, cf{E.x - Q.y, -(E.y + Q.x)});
            const cf Em{a.x + b.x, a.y - b.y};
            const cf Dm{a.x - b.x, a.y + b.y};
            const cf Qm{__builtin_fmaf(tm[i].x, Dm.x, tm[i].y * Dm.y), __builtin_fmaf(tm[i].x, Dm.y, -(tm[i].y * Dm.x))};
            lds_st(Y + (i == 0 ? km : 448 - kk), cf{Em.x - Qm.y, -(Em.y + Qm.x)});
        }
    }
};

// frame-interleaved shapes: the running phase of slot j = the unit's base + the increments of slots 0..j; all kG of them move the base on
template <int kG>
__device__ __forceinline__ void r2_running_phase(uint32_t (&qs)[5], uint32_t (&mine)[5], const uint32_t (&iv)[kG][5], int j)
{
    uint32_t base[5];
#pragma unroll
    for (int q = 0; q < 5; q++) { mine[q] = qs[q]; base[q] = qs[q]; }
#pragma unroll
    for (int i2 = 0; i2 < kG; i2++) {
#pragma unroll
        for (int q = 0; q < 5; q++) {
            base[q] += iv[i2][q];
            if (i2 <= j) mine[q] += iv[i2][q];
        }
    }
#pragma unroll
    for (int q = 0; q < 5; q++) qs[q] = base[q];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// R3: overlap-add.  Sample n = 2 (lane + 64 r) + {0,1} of a frame falls into hop block r >> 1 at offset 2 lane + 128 (r & 1) + {0,1}: a lane touches
// the same 4 offsets of every block.  One frame per step: the 3 open blocks are 12 VGPRs (the 4th block a frame touches is new); block fz-3 is complete
// once frame fz is in; contributions arrive in increasing frame order, as in the oracle.  Frame-interleaved: quarters 1..3 go to LDS, quarter 0 waits in
// 4 VGPRs for the next step, where the other three quarters of its block are fetched from the slots of the three frames before.
// Sums are kept unscaled; the constants of the tolerance path — 1/512 (inverse FFT), 1/2 (c2r pre-twiddle), 1/2 (analysis split) and 2/3 (overlap-add
// gain) — scale the finished block.
constexpr float kSynthGain = NAE_OLA_GAIN / 2048.0f;

// zs[r] = conj(z[n]) * 512 (x 4), n = lane + 64 r  ->  time samples 2n, 2n+1, windowed: y[quarter][offset]
__device__ __forceinline__ void r3_window(const cf (&zs)[8], const cf (&wn)[8], float (&y)[4][4])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cf w = wn[r];
        y[r >> 1][2 * (r & 1)] = zs[r].x * w.x;
        y[r >> 1][2 * (r & 1) + 1] = -(zs[r].y * w.y);   // the sign undoes the conjugation
    }
}
// one frame per step: the block this frame completes, the three open ones move on
__device__ __forceinline__ void r3_ola_shift(float (&r0)[4], float (&r1)[4], float (&r2)[4], const float (&y)[4][4], float (&o)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o[i] = (r0[i] + y[0][i]) * kSynthGain;
        r0[i] = r1[i] + y[1][i];
        r1[i] = r2[i] + y[2][i];
        r2[i] = y[3][i];
    }
}
// frame-interleaved: quarters 3, 2, 1 of the three frames before the block's last one (oq[2], oq[1], oq[0]), then its own quarter 0 — the frame order
__device__ __forceinline__ void r3_ola_finish(const float4 (&oq)[3], const float (&y0)[4], float (&o)[4])
{
    o[0] = oq[2].x; o[1] = oq[2].y; o[2] = oq[2].z; o[3] = oq[2].w;
    o[0] += oq[1].x; o[1] += oq[1].y; o[2] += oq[1].z; o[3] += oq[1].w;
    o[0] += oq[0].x; o[1] += oq[0].y; o[2] += oq[0].z; o[3] += oq[0].w;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (o[i] + y0[i]) * kSynthGain;
}
// where quarter i (1..3) of the block that slot j's previous frame completed lies: slot (j - i) mod kG of the unit, floor((j - i) / kG) steps earlier
template <int kG>
__device__ __forceinline__ void r3_quarter_source(int slot, int j, int i, int t, int gens, int& sl, int& gen)
{
    const int rel = j - i;                                   // < 0: an earlier step
    const int back = rel >= 0 ? 0 : (-rel + kG - 1) / kG;
    sl = slot - j + rel + back * kG;
    gen = (t - 1 - back + 2 * gens) % gens;
}

// a finished hop block to the destination: buffer stores off a scalar descriptor of the block + one 32-bit lane offset (plain pointer stores made
// hipcc hoist four 64-bit per-lane addresses out of the frame loop); the last block of a stream may be partial
struct BlockOut { float* optr; long long out_fs; bool out_vec; };
__device__ __forceinline__ void r3_store_block(const PvParams& p, const PipeItem& it, const BlockOut& bo, long long be, const float (&o)[4], int ls)
{
    if (be >= it.b0 && be < it.b_end && be * NAE_HOP < p.mid_len) {
        float* pb = bo.optr + be * NAE_HOP * bo.out_fs;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, -1, 0x00020000);
        auto st = [&](unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 0); };
        const unsigned fs4 = 4u * (unsigned)bo.out_fs;              // bytes between consecutive samples
        const unsigned oa = 2u * (unsigned)ls * fs4;                // sample 2 lane of the block
        if ((be + 1) * NAE_HOP <= p.mid_len) {
            if (bo.out_vec) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o[0]), __float_as_uint(o[1])}, rs, (int)(8u * ls), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o[2]), __float_as_uint(o[3])}, rs, (int)(512u + 8u * ls), 0, 0);
            } else {
                st(oa, o[0]); st(oa + fs4, o[1]); st(oa + 128u * fs4, o[2]); st(oa + 129u * fs4, o[3]);
            }
        } else {
            const int rem = (int)(p.mid_len - be * NAE_HOP);
            if (2 * ls + 0 < rem) st(oa, o[0]);
            if (2 * ls + 1 < rem) st(oa + fs4, o[1]);
            if (128 + 2 * ls < rem) st(oa + 128u * fs4, o[2]);
            if (129 + 2 * ls < rem) st(oa + 129u * fs4, o[3]);
        }
    }
}
// interleaved stereo, one frame per step: the wave of channel c writes HALF c of the interleaved block — samples 128 c + 2 lane, + 1 of both channels = 16
// contiguous bytes per lane, 1 KiB per wave — from its own two values (ch_c) and its partner's (ch_o), instead of four dwords that fill a quarter of every 16 bytes
__device__ __forceinline__ void r3_store_dense_half(float* stream_base, int pend_be, int c, cf ch0, cf ch1, int lx)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(stream_base + (long long)pend_be * (2 * NAE_HOP), 0, -1, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(ch0.x), __float_as_uint(ch1.x), __float_as_uint(ch0.y), __float_as_uint(ch1.y)}, rs, 16 * lx, 1024 * c, 0);
}

} // namespace nae
