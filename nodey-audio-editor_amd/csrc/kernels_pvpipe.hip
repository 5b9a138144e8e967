// kernels_pvpipe.hip — pass 3 of the phase vocoder (K7) as a three-role wave pipeline for gfx950.
//
// Why a pipeline.  On gfx950 one wave issues at most one vector instruction per 4.5-5 cycles, while a SIMD with 6-8
// resident waves issues one per 1.0-1.6 cycles (profiles/r02_valu_issue.md).  A stream-channel of the vocoder is a serial
// chain of frames (integer phase accumulator, overlap-add), so "one wave per stream-channel" (the round-1 kernel:
// 239 VGPRs, 2 waves per SIMD on the 2048 stream-channels of BASELINE.json configs[4]) leaves more than half of the
// vector issue slots empty.  Here each stream-channel is served by THREE waves, one per stage, that hand a frame on
// through LDS once per step, and every stage fits 80 VGPRs (6 waves per SIMD):
//
//   step t:   R1  frame t    load, Hann window, forward FFT                         -> Z   (its own FFT scratch)
//             R2  frame t-1  r2c split, atan2 -> Q0.32, exact phase advance, rotate  -> Y   (hand-off buffer)
//             R3  frame t-2  c2r pre-twiddle, inverse FFT (by forward FFT), overlap-add, store the finished hop block
//
// A step has two workgroup barriers: after A the consumers (R2, R3) read what the producers left in step t-1 into
// registers; after B the producers overwrite.  So the hand-off buffers need no double buffering and one stream-channel
// costs 2 x 4608 B (FFT scratch of R1 / R3) + 4160 B (Y) of LDS; four 384-thread workgroups (one stereo stream, or two
// mono streams, each) fit a CU: 4 x (12352 B tables + 2 x 13376 B) = 156416 B of the 160 KiB.
//
// The arithmetic — and therefore every integer phase — is the canonical one of DESIGN.md §3 (same dft8_fwd / cmul_tw /
// atan2_q32 / phase_inc as the other kernels); only where data waits between operations differs.
// Replaces: SoundTouch behind /root/reference/src/processor/audio-velocity.cpp:369-428 (algorithm differs: DESIGN.md §3).
#include "stft_common.h"

namespace nae {

constexpr int kPipeSc = 4;                                  // stream-channels per workgroup: 12 waves = 3 per SIMD, so two
                                                            // workgroups pack a CU exactly (6-wave workgroups left one in four out)
constexpr int kPipeThreads = 64 * 3 * kPipeSc;              // 768: waves 0-3 = R1, 4-7 = R2, 8-11 = R3 (of sc 0..3)
constexpr int kYCf = 520;                                   // Y[0..512] natural order
constexpr size_t kPipeLdsTables = NAE_FFT_N * sizeof(float) + (kT1024Pad + 64 + kTwaCf) * sizeof(cf);
constexpr size_t kPipeLdsPerSc = (2 * kPadScratchCf + kYCf) * sizeof(cf);
constexpr size_t kPipeLds = kPipeLdsTables + kPipeSc * kPipeLdsPerSc;
static_assert(2 * kPipeLds <= 160 * 1024, "two workgroups per CU");

// every LDS operation of this wave has completed, then the workgroup barrier (vector-memory operations stay in flight:
// the frame prefetch of R1 and the block stores of R3 must not be drained twice per step)
__device__ __forceinline__ void pipe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef NAE_PIPE_STAMPS
// diagnostic build only (tools/pipe_stamps.sh): s_memtime around both barriers of steps 200..207, workgroup 0
__device__ unsigned long long g_pipe_stamps[12 * 8 * 4];
__device__ unsigned long long g_pipe_total[4 * 64];      // {cycles, realtime ticks, steps, xcc/hw id} of wave 0 of 64 sampled workgroups
#define PIPE_BARRIER(t, which)                                                                                              \
    do {                                                                                                                    \
        const bool st_ = blockIdx.x == 0 && (t) >= 200 && (t) < 208;                                                        \
        const unsigned long long a_ = __builtin_amdgcn_s_memtime();                                                         \
        pipe_barrier();                                                                                                     \
        const unsigned long long b_ = __builtin_amdgcn_s_memtime();                                                         \
        if (st_ && lane == 0) { g_pipe_stamps[(wave * 8 + ((t) - 200)) * 4 + 2 * (which)] = a_; g_pipe_stamps[(wave * 8 + ((t) - 200)) * 4 + 2 * (which) + 1] = b_; } \
    } while (0)
#else
#define PIPE_BARRIER(t, which) pipe_barrier()
#endif

// Issue priority.  Two workgroups share a CU and the hardware arbitrates equal priorities by age: measured (tools/pipe_stamps.py)
// the workgroup that arrived first runs a step in 4700 cycles, its neighbour in 7060, on every CU — the first finishes a
// third earlier and the CU then runs half empty.  So the two take turns: a workgroup learns whether it was the first or the
// second on its CU (g_cu_arrivals, counted per physical CU, never reset: only the parity is used) and raises its priority on
// alternate steps (6.95-7.03 ms against 7.14-7.17 on one box; raising the role on a step's critical path as well — R1 between
// barriers A and B, R3 between B and A — made it 7.4).
__device__ unsigned g_cu_arrivals[8 * 4 * 16];
__device__ __forceinline__ void pipe_prio(int t, int slot)
{
    if ((t + slot) & 1) __builtin_amdgcn_s_setprio(2);      // wave-uniform
    else __builtin_amdgcn_s_setprio(0);
}

template <bool kUnit>
__global__ __launch_bounds__(kPipeThreads, 6) void pv_pipe_kernel(SigViewD src, PvParams p, long long n_sc,
                                                                 const uint32_t* __restrict__ base_phase, OutViewD out, Tables tb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kPipeThreads) hann[i] = tb.hann[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kPipeThreads) t1024[i] = tb.t1024[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = tb.w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, tb.w512, threadIdx.x, kPipeThreads);
    __shared__ int s_slot;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_ID: CU 8-11, SE 13-14
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;     // XCC_ID
        const unsigned key = (xcc * 4 + ((hw >> 13) & 3u)) * 16 + ((hw >> 8) & 15u);
        s_slot = (int)(atomicAdd(&g_cu_arrivals[key], 1u) & 1u);
    }
    __syncthreads();
    const int slot = __builtin_amdgcn_readfirstlane(s_slot);

    const int wave = wave_id();
    const int role = wave / kPipeSc, half = wave % kPipeSc;  // scalars
    const int lane = threadIdx.x & 63;
    // stereo: a workgroup owns both channels of (stream, tile), so their block stores of an interleaved destination
    // happen in the same step and merge in L2; mono: two consecutive (stream, tile) items
    long long sc;
    int tile;
    if (p.ch == 2) {
        const long long pair = 2 * (long long)blockIdx.x + (half >> 1);      // (stream, tile), tile fastest
        sc = 2 * (pair / p.n_tiles) + (half & 1);
        tile = (int)(pair % p.n_tiles);
    } else {
        const long long item = kPipeSc * (long long)blockIdx.x + half;
        sc = item / p.n_tiles;
        tile = (int)(item % p.n_tiles);
    }
    if (sc >= n_sc) return;                                  // a terminated wave no longer counts at s_barrier
    const long long s_idx = sc / p.ch;
    const int c = (int)(sc % p.ch);

    cf* S1 = reinterpret_cast<cf*>(smem + kPipeLdsTables + half * kPipeLdsPerSc);
    cf* Y = S1 + kPadScratchCf;
    cf* S3 = Y + kYCf;

    const long long b0 = p.f_origin + (long long)tile * p.tile;      // first output block == first frame of the tile
    long long b_end = b0 + p.tile;
    if (b_end > p.f_stop) b_end = p.f_stop;
    long long f_end = b_end + 3;                                       // frames b0 .. b_end+2 feed blocks b0 .. b_end-1
    if (f_end > p.frames) f_end = p.frames;
    const long long f_first = (b0 > 0 ? b0 - 1 : 0);                   // frame b0-1 only primes the previous phase
    const int n = (int)(f_end - f_first);
    if (n <= 0) return;

#ifdef NAE_PIPE_STAMPS
    const unsigned long long tot_c0 = __builtin_amdgcn_s_memtime(), tot_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (role == 0) {
        // ------------------------------------------------------------------------------------------ R1: analysis FFT
        ChanView in{src.base + s_idx * src.ss + c * src.cs, src.fs, p.in_len};
        const FftLds L = make_fft_lds(S1, twa, w64, lane);
        const cf* hw = reinterpret_cast<const cf*>(hann) + lane;             // window of samples 2 (lane + 64 j), +1
        cf nxt[8], va[8];
        load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first), lane);
#pragma unroll 1
        for (int t = 0; t < n + 2; t++) {
            PIPE_BARRIER(t, 0);                               // A
            pipe_prio(t, slot);
            if (t < n) {
                // register-only part while R2 reads Z of frame t-1 out of this wave's scratch
                // (all reads of a phase are requested before the first one is used: the accesses are volatile, so the compiler
                // keeps them where they are written, and one read per product would cost one LDS round trip each)
                cf w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = lds_ld(hw + 64 * j);
#pragma unroll
                for (int j = 0; j < 8; j++) va[j] = cf{nxt[j].x * w[j].x, nxt[j].y * w[j].y};
                fft512_pad_a(va, L);
            }
            PIPE_BARRIER(t, 1);                               // B: R2 holds X of frame t-1 in registers
            if (t < n) {
                fft512_pad_bc(va, L);
#pragma unroll
                for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, va[r]);
                if (lane == 0) S1[512] = va[0];               // so that the mirror of bin 0 is read like any other
                // request the next frame now: R1 is the shortest role, the loads land while it waits at the barriers
                if (t + 1 < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + t + 1), lane);
            }
        }
#ifdef NAE_PIPE_STAMPS
        if (wave == 0 && lane == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64) {
            unsigned long long* o = g_pipe_total + 4 * (blockIdx.x >> 3);
            o[0] = __builtin_amdgcn_s_memtime() - tot_c0; o[1] = __builtin_amdgcn_s_memrealtime() - tot_r0; o[2] = n + 2;
            o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
#endif
    } else if (role == 1) {
        // ------------------------------------------------------------------------------------------ R2: phases
        const cf* Zn = S1 + lane;
        const cf* Zm = S1 + 64 - lane;
        cf* Yn = Y + lane;
        const cf* tsp = t1024 + lane;
        uint32_t qs[9], qp[9];
        {
            const uint32_t* bp = base_phase + (sc * p.phase_tiles + (long long)tile * p.phase_step) * kT1024Pad;
#pragma unroll
            for (int r = 0; r < 8; r++) { qs[r] = bp[lane + 64 * r]; qp[r] = 0; }
            qs[8] = bp[512];
            qp[8] = 0;
        }
        long long s_prev = 0;
#pragma unroll 1
        for (int t = 0; t < n + 2; t++) {
            const bool active = (t >= 1) && (t <= n);
            const long long f = f_first + t - 1;
            PIPE_BARRIER(t, 0);                               // A: Z of frame f is complete
            pipe_prio(t, slot);
            cf va[8], nyq{0.0f, 0.0f};
            if (active) {
                cf vb[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { va[r] = lds_ld(Zn + 64 * r); vb[r] = lds_ld(Zm + 448 - 64 * r); }
                const cf z0 = S1[0];
                // r2c split -> 2 X (phases are scale-invariant; the factor is undone in R3's output gain: a factor 2 is
                // exact in every product on the way)
                {
                    const cf E = cf{z0.x + z0.x, z0.y - z0.y};
                    const cf O = cf{z0.x - z0.x, z0.y + z0.y};
                    const cf P = cmul_tw(O, t1024[512]);
                    nyq = cf{E.x + P.y, E.y - P.x};
                }
                cf tw[8];
#pragma unroll
                for (int r = 0; r < 8; r++) tw[r] = lds_ld(tsp + 64 * r);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const cf A = va[r], B = vb[r];
                    const cf E = cf{A.x + B.x, A.y - B.y};
                    const cf O = cf{A.x - B.x, A.y + B.y};
                    const cf P = cmul_tw(O, tw[r]);
                    va[r] = cf{E.x + P.y, E.y - P.x};
                }
            }
            PIPE_BARRIER(t, 1);                               // B: R1 may overwrite its scratch
            if (active) {
                const long long s = frame_start(p, f);
                uint32_t qa[9];
                phases_of(va, nyq, qa);
                if (f >= b0) {
                    if (f == 0) {
#pragma unroll
                        for (int r = 0; r < 9; r++) qs[r] += qa[r];
                    } else {
                        const unsigned d = (unsigned)(s - s_prev);
                        const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
                        phase_inc(qa, qp, qs, lane, d, R);
                    }
                }
#pragma unroll
                for (int r = 0; r < 9; r++) qp[r] = qa[r];
                s_prev = s;
                if (f >= b0) {
                    // synthesis spectrum |X| e^{2 pi i qs} == X e^{i (qs - qa)}: rotate by the phase difference (tolerance path)
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const float ph = (float)(int32_t)(qs[r] - qa[r]) * (1.0f / 4294967296.0f);
                        const float cs = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
                        cf y{__builtin_fmaf(va[r].x, cs, -(va[r].y * sn)), __builtin_fmaf(va[r].x, sn, va[r].y * cs)};
                        if (r == 0 && lane == 0) y.y = 0.0f;   // c2r ignores Im Y[0]
                        lds_st(Yn + 64 * r, y);
                    }
                    if (lane == 0) {
                        const float ph = (float)(int32_t)(qs[8] - qa[8]) * (1.0f / 4294967296.0f);
                        const float cs = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
                        Y[512] = cf{__builtin_fmaf(nyq.x, cs, -(nyq.y * sn)), 0.0f};
                    }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------ R3: synthesis
        float* optr = out.base + s_idx * out.ss + c * out.cs;
        const bool out_vec = (out.fs == 1) && ((reinterpret_cast<uintptr_t>(optr) & 15) == 0);
        const FftLds L = make_fft_lds(S3, twa, w64, lane);
        const cf* Yn = Y + lane;
        const cf* Ym = Y + 64 - lane;
        const cf* tsp = t1024 + lane;
        const cf* hw = reinterpret_cast<const cf*>(hann) + lane;
        // Overlap-add in registers.  Sample n = 2 (lane + 64 r) + {0,1} of a frame falls into hop block r >> 1 at offset
        // 2 lane + 128 (r & 1) + {0,1}: a lane touches the same 4 offsets of every block, so the 3 open blocks are 12 VGPRs
        // (the 4th block a frame touches is new).  Block fz-3 is complete once frame fz is in; contributions arrive in
        // increasing frame order, as in the oracle.  Sums are kept unscaled; the constants of the tolerance path — 1/512
        // (inverse FFT), 1/2 (c2r pre-twiddle), 1/2 (analysis split) and 2/3 (overlap-add gain) — scale the finished block.
        constexpr float kGain = NAE_OLA_GAIN / 2048.0f;
        float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int t = 0; t < n + 2; t++) {
            const long long fz = f_first + t - 2;
            const bool active = (t >= 2) && (fz >= b0);
            PIPE_BARRIER(t, 0);                               // A: Y of frame fz is complete
            pipe_prio(t, slot);
            cf zs[8];
            if (active) {
                // c2r pre-twiddle into FFT input layout, conjugated (inverse = conj(FFT(conj Z)) / 512); 2E, 2D: see kGain
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cf Xk[4], Xm[4], T[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int r = 4 * h + q;
                        Xk[q] = lds_ld(Yn + 64 * r);
                        Xm[q] = lds_ld(Ym + 448 - 64 * r);
                        T[q] = lds_ld(tsp + 64 * r);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const cf E{Xk[q].x + Xm[q].x, Xk[q].y - Xm[q].y};
                        const cf D{Xk[q].x - Xm[q].x, Xk[q].y + Xm[q].y};
                        const cf Q{__builtin_fmaf(T[q].x, D.x, T[q].y * D.y), __builtin_fmaf(T[q].x, D.y, -(T[q].y * D.x))};
                        zs[4 * h + q] = cf{E.x - Q.y, -(E.y + Q.x)};
                    }
                }
            }
            PIPE_BARRIER(t, 1);                               // B: R2 may overwrite Y
            if (active) {
                fft512_pad_a(zs, L);
                // (the synthesis window is requested now: its round trip hides behind passes B and C)
                cf wn[8];
#pragma unroll
                for (int r = 0; r < 8; r++) wn[r] = lds_ld(hw + 64 * r);
                fft512_pad_bc(zs, L);
                // zs[r] = conj(z[n]) * 512 (x 4), n = lane + 64 r  ->  time samples 2n, 2n+1, windowed
                float y[4][4];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const cf w = wn[r];
                    y[r >> 1][2 * (r & 1)] = zs[r].x * w.x;
                    y[r >> 1][2 * (r & 1) + 1] = -(zs[r].y * w.y);   // the sign undoes the conjugation
                }
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    o[i] = (r0[i] + y[0][i]) * kGain;
                    r0[i] = r1[i] + y[1][i];
                    r1[i] = r2[i] + y[2][i];
                    r2[i] = y[3][i];
                }
                const long long be = fz - 3;                  // wave-uniform: the block's base pointer stays scalar
                if (be >= b0 && be < b_end && be * NAE_HOP < p.mid_len) {
                    // buffer stores: scalar descriptor of the block + one 32-bit lane offset (plain pointer stores made hipcc
                    // hoist four 64-bit per-lane addresses out of the frame loop: 8 VGPRs, spilled at 80)
                    float* pb = optr + be * NAE_HOP * out.fs;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, -1, 0x00020000);
                    auto st = [&](unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 0); };
                    const unsigned fs4 = 4u * (unsigned)out.fs;                 // bytes between consecutive samples
                    const unsigned oa = 2u * (unsigned)lane * fs4;              // sample 2 lane of the block
                    if ((be + 1) * NAE_HOP <= p.mid_len) {
                        if (out_vec) {
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o[0]), __float_as_uint(o[1])}, rs, (int)(8u * lane), 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(o[2]), __float_as_uint(o[3])}, rs, (int)(512u + 8u * lane), 0, 0);
                        } else {
                            st(oa, o[0]); st(oa + fs4, o[1]); st(oa + 128u * fs4, o[2]); st(oa + 129u * fs4, o[3]);
                        }
                    } else {
                        const int rem = (int)(p.mid_len - be * NAE_HOP);
                        if (2 * lane + 0 < rem) st(oa, o[0]);
                        if (2 * lane + 1 < rem) st(oa + fs4, o[1]);
                        if (128 + 2 * lane < rem) st(oa + 128u * fs4, o[2]);
                        if (129 + 2 * lane < rem) st(oa + 129u * fs4, o[3]);
                    }
                }
            }
        }
    }
}

} // namespace nae

using namespace nae;

#ifdef NAE_PIPE_STAMPS
extern "C" int nae_debug_read_pipe_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pipe_stamps), sizeof(unsigned long long) * 12 * 8 * 4); }
extern "C" int nae_debug_read_pipe_total(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pipe_total), sizeof(unsigned long long) * 4 * 64); }
#endif

int nae_launch_pv_pipe(nae_ctx* ctx, const PvParams& p, const SigViewD& src, long long n_sc, const uint32_t* phase_ws,
                       const OutViewD& out, bool unit_stride)
{
    const long long items = n_sc * p.n_tiles;
    if (items == 0) return NAE_OK;
    const long long groups = (items + kPipeSc - 1) / kPipeSc;
    if (groups > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "pv_pipe_kernel: grid too large");
    Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    if (unit_stride)
        NAE_KLAUNCH(ctx, "pv_synth_kernel", (pv_pipe_kernel<true>), dim3((unsigned)groups), dim3(kPipeThreads), kPipeLds, ctx->stream, src, p, n_sc,
                    phase_ws, out, tb);
    else
        NAE_KLAUNCH(ctx, "pv_synth_kernel", (pv_pipe_kernel<false>), dim3((unsigned)groups), dim3(kPipeThreads), kPipeLds, ctx->stream, src, p, n_sc,
                    phase_ws, out, tb);
    return nae_check(ctx, hipGetLastError(), "pv_pipe_kernel");
}
