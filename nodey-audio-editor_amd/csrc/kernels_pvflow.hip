// kernels_pvflow.hip — pass 3 of the phase vocoder (K7) for batches that give a CU ONE workgroup: the four-role wave pipeline of
// kernels_pvpipe.hip with ONE workgroup barrier per step instead of two.
//
// Why.  With one 16-wave workgroup per CU (4 waves per SIMD: the 128 to 512 streams a rank of a multi-GPU job owns) the two-barrier step of
// kernels_pvpipe.hip leaves the vector ALU busy 54 % of the time (profiles/r05_pipe_stamps_128_streams.txt): every barrier interval starts
// with all 16 waves waiting for their LDS reads and ends with the fast roles waiting for the slow one — R1 and R2b wait 2000 of a step's
// 5030 cycles — and a wave that runs alone on its SIMD issues at 4.6 cycles per instruction instead of 2.15 (profiles/r05_valu_wallclock.md).
// A CU with one workgroup has LDS to spare, so here every hand-off buffer exists TWICE: in step t a producer writes generation t & 1 while
// its consumer reads generation (t - 1) & 1, one barrier per step separates the two, and between two barriers every wave does a whole
// step's work of its role (270 to 330 vector instructions: level).  The price is pipeline depth (a frame needs 5 steps instead of 4 to
// leave, of ~470) and LDS (144 / 156 KB of the CU's 160).
//
//   step t, slot j of a unit (kG consecutive frames of ONE stream-channel per step; kG = 1: the slot IS a stream-channel):
//     R1   frame kG t + j        window, forward FFT (scratch = its own Z buffer of generation t)             -> Z[t]
//     R2a  frame kG (t-1) + j    bins of half 0 (+ 256, 512): r2c split, atan2 -> Q0.32                          -> QA[t]   (kG > 1)
//          frame kG (t-2) + j    exact phase increment against the predecessor's analysis phase QA[t-1]          -> INC[t]  (kG > 1)
//          frame kG (t-3) + j    running phase = base + the unit's increments INC[t-1] up to slot j; rotation;
//                                c2r pre-twiddle                                                                  -> Y[t]
//     R2b  the same for the bins of half 1
//     R3   frame kG (t-4) + j    inverse FFT (by forward FFT; scratch = the Y buffer it has just read), synthesis window;
//                                quarters 1..3 of the windowed frame                                              -> OLA[t]  (kG > 1)
//          frame kG (t-5) + j    the hop block this frame completed: quarters of the three frames before it (OLA[t-1], OLA[t-2], ...)
//                                in frame order + its own quarter 0 (registers), gain, store
//   kG = 1: R2 analyses, advances and rotates frame t - 1 in one step (phases stay in registers), R3 works on frame t - 2 and keeps the
//   three open hop blocks in registers — the schedule of kernels_pvpipe.hip's kG = 1, with one barrier.
//
// The arithmetic, its order and therefore every bit of the output are those of kernels_pvpipe.hip (tests/test_gpu_stft.py:
// test_k7_pipeline_modes_agree_bit_for_bit); only where data waits between operations differs.
// Replaces: SoundTouch behind /root/reference/src/processor/audio-velocity.cpp:369-428 (algorithm differs: DESIGN.md §3).
#include "stft_common.h"
#include "pv_roles.h"

namespace nae {

constexpr int kFlowSlots = 4, kFlowThreads = 64 * 4 * kFlowSlots;
constexpr int flow_ola_gens(int kG) { return kG == 1 ? 0 : kG == 2 ? 4 : 3; }     // generations read (t-1 ... t-1-ceil(3/kG)) + the one written
constexpr size_t kFlowBuf = kPadScratchCf * sizeof(cf);                             // one Z / Y generation (doubles as FFT scratch)
constexpr size_t flow_lds_per_slot(int kG)
{
    return 4 * kFlowBuf + (kG == 1 ? 0 : 4 * kPhasePad * sizeof(uint32_t) + (size_t)flow_ola_gens(kG) * 3 * kOlaQuarter * sizeof(float));
}
constexpr size_t flow_lds(int kG) { return kFlowSlots * flow_lds_per_slot(kG); }
static_assert(flow_lds(1) <= 160 * 1024 && flow_lds(2) <= 160 * 1024 && flow_lds(4) <= 160 * 1024, "one workgroup per CU");
// (A 64-VGPR build of the one-frame-per-step shape — tables in LDS, two workgroups per CU at exactly 80 KiB each, the priority time slices of
// kernels_pvpipe.hip — was measured at 1024 streams: 6.39-6.44 against 6.10-6.18 ms; tools/experiments/r05_flow_lean.patch, profiles/r05_flow.md.)

template <bool kUnit, int kG>
__global__ __launch_bounds__(kFlowThreads, 4) void pv_flow_kernel(SigViewD src, PvParams p, long long n_sc, const uint32_t* __restrict__ base_phase,
                                                                  OutViewD out, Tables tb)
{
    static_assert(kFlowSlots % kG == 0, "a unit's frames share a workgroup");
    constexpr int kUnits = kFlowSlots / kG;                  // stream-channels (x tile) per workgroup
    constexpr int kGens = flow_ola_gens(kG);
    constexpr size_t kPer = flow_lds_per_slot(kG);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave = wave_id();
    const int role = wave / kFlowSlots, slot = wave % kFlowSlots;   // scalars
    const int unit = slot / kG, j = slot % kG;                      // j: frame of the step this slot works on
    const int lane = threadIdx.x & 63;
    // stereo: units 2i, 2i+1 are the two channels of one (stream, tile); mono: consecutive (stream, tile) items (as kernels_pvpipe.hip)
    const long long ug = (long long)kUnits * blockIdx.x + unit;
    long long sc;
    int tile;
    if (p.ch == 2) {
        const long long pair = ug >> 1;
        sc = 2 * (pair / p.n_tiles) + (ug & 1);
        tile = (int)(pair % p.n_tiles);
    } else {
        sc = ug / p.n_tiles;
        tile = (int)(ug % p.n_tiles);
    }
    if (sc >= n_sc) return;                                  // a terminated wave no longer counts at s_barrier
    const long long s_idx = sc / p.ch;
    const int c = (int)(sc % p.ch);

    // per slot: Z[2] | Y[2] | QA[2] | INC[2] | OL[kGens][3][256]
    auto slot_base = [&](int sl) { return smem + sl * kPer; };
    auto x_z = [&](int sl, int gen) { return reinterpret_cast<cf*>(slot_base(sl)) + gen * kPadScratchCf; };
    auto x_y = [&](int sl, int gen) { return reinterpret_cast<cf*>(slot_base(sl)) + (2 + gen) * kPadScratchCf; };
    auto x_qa = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(slot_base(sl) + 4 * kFlowBuf) + gen * kPhasePad; };
    auto x_inc = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(slot_base(sl) + 4 * kFlowBuf) + (2 + gen) * kPhasePad; };
    auto x_ola = [&](int sl, int gen) {
        return reinterpret_cast<float*>(slot_base(sl) + 4 * kFlowBuf + 4 * kPhasePad * sizeof(uint32_t)) + gen * 3 * kOlaQuarter;
    };

    const long long b0 = p.f_origin + (long long)tile * p.tile;      // first output block == first frame of the tile
    long long b_end = b0 + p.tile;
    if (b_end > p.f_stop) b_end = p.f_stop;
    long long f_end = b_end + 3;                                       // frames b0 .. b_end+2 feed blocks b0 .. b_end-1
    if (f_end > p.frames) f_end = p.frames;
    const long long f_first = (b0 > 0 ? b0 - 1 : 0);                   // frame b0-1 only primes the previous phase
    const int n = (int)(f_end - f_first);
    if (n <= 0) return;
    const int steps = (n + kG - 1) / kG;
    constexpr int kR3Lag = kG == 1 ? 2 : 4;                            // steps between a frame's R1 step and its R3 step
    // interleaved stereo output, one frame per step: a finished block leaves one barrier later, as a dense piece assembled with the other channel's wave (R3)
    const bool dense_shape = kG == 1 && p.ch == 2 && out.fs == 2 && out.cs == 1;
    const int T = steps + kR3Lag + (kG == 1 ? (dense_shape ? 1 : 0) : 1);   // kG > 1: a block leaves one step behind its last frame
    /*pipe:begin*/

    if (role == 0) {
        // ------------------------------------------------------------------------------------------ R1: analysis FFT
        ChanView in{src.base + s_idx * src.ss + c * src.cs, src.fs, p.in_len};
        cf nxt[8], va[8];
        if (j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + j), lane);
        cf r_w[8], r_ta[7], r_tb[7];                          // window and both twiddle sets stay in registers (128 VGPRs per wave)
#pragma unroll
        for (int r = 0; r < 8; r++) r_w[r] = reinterpret_cast<const cf*>(tb.hann)[lane + 64 * r];
#pragma unroll
        for (int q = 0; q < 7; q++) { r_ta[q] = tb.w512[lane * (q + 1)]; r_tb[q] = tb.w512[8 * (lane & 7) * (q + 1)]; }
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const bool cur = kG * t + j < n;
            pipe_barrier();
            if (cur) {
#pragma unroll
                for (int r = 0; r < 8; r++) va[r] = cf{nxt[r].x * r_w[r].x, nxt[r].y * r_w[r].y};
                fft512_pad_a_tw(va, r_ta);
                cf* Z = x_z(slot, t & 1);
                const FftLds L = make_fft_lds(Z, nullptr, nullptr, lane);
                cf none[8];
                fft512_pad_bc_g<true, false>(va, L, r_tb, nullptr, none);
#pragma unroll
                for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, va[r]);
                if (lane == 0) Z[512] = va[0];                // so that the mirror of bin 0 is read like any other
                // request the next frame now: the loads land while the wave waits at the barrier
                if (kG * (t + 1) + j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + kG * (t + 1) + j), lane);
            }
        }
        /*pipe:r1-end*/
    } else if (role <= 2) {
        // ------------------------------------------------------------------------------------------ R2a / R2b: phases
        // bins in mirror pairs, bin 512 as a fifth item in half 0: kernels_pvpipe.hip
        const int h = role - 1;
        const int k0 = lane + 128 * h;                        // items 0 / 2: k0, k0 + 64; items 1 / 3: 512 - k0 (256 in lane 0 of h = 0), 448 - k0
        const bool dc = (h == 0) && (lane == 0);              // the lane of bins 0, 256 and 512
        const int km0 = dc ? 256 : 512 - k0;                  // bin of item 1
        cf tk[2], tm[2];                                      // split twiddles of the items' bins
#pragma unroll
        for (int i = 0; i < 2; i++) { tk[i] = tb.t1024[k0 + 64 * i]; tm[i] = tb.t1024[i == 0 ? km0 : 448 - k0]; }
        const cf tms = tb.t1024[512 - k0];                    // split twiddle of the pair's mirror (differs from tm[0] in lane 0 of h = 0: bin 512)
        uint32_t qs[5], qp[5];                                // [4]: bin 512 (h = 0)
        {
            const uint32_t* bp = base_phase + (sc * p.phase_tiles + (long long)tile * p.phase_step) * kT1024Pad;
            if (p.base_zero) {
#pragma unroll
                for (int q = 0; q < 5; q++) qs[q] = 0;
            } else {
                qs[0] = bp[k0]; qs[1] = bp[km0]; qs[2] = bp[k0 + 64]; qs[3] = bp[448 - k0];
                qs[4] = bp[512];
            }
#pragma unroll
            for (int q = 0; q < 5; q++) qp[q] = 0;
        }
        // kG > 1: H1 = the frame analysed one step ago (waits for its increment), H2 = two steps ago (waits for the running phase)
        cf x1[5], x2[5];
        uint32_t qa1[5] = {0, 0, 0, 0, 0}, qa2[5] = {0, 0, 0, 0, 0};
        uint32_t pcur[5] = {0, 0, 0, 0, 0};                   // slot 0 of a unit: the analysis phases of the frame before H1 (the unit's last slot, a step earlier)
#pragma unroll
        for (int q = 0; q < 5; q++) { x1[q] = cf{0.0f, 0.0f}; x2[q] = cf{0.0f, 0.0f}; }
        auto inc_items = [&](const uint32_t (&qa)[5], const uint32_t (&qv)[5], unsigned d, unsigned R, uint32_t (&inc)[5]) {
            const int kk = k0, km = km0;
            inc[0] = pipe_inc(qa[0], qv[0], (unsigned)kk, d, R);
            inc[1] = pipe_inc(qa[1], qv[1], (unsigned)km, d, R);
            inc[2] = pipe_inc(qa[2], qv[2], (unsigned)(kk + 64), d, R);
            inc[3] = pipe_inc(qa[3], qv[3], (unsigned)(448 - kk), d, R);
            inc[4] = (h == 0) ? pipe_inc(qa[4], qv[4], 512u, d, R) : 0u;
        };
        auto carry_store = [&](const uint32_t (&q)[5]) {
            uint32_t* co = p.carry_out + sc * kT1024Pad;
            co[k0] = q[0]; co[km0] = q[1]; co[k0 + 64] = q[2]; co[448 - k0] = q[3];
            if (dc) co[512] = q[4];
        };
        // rotation by the phase difference and the in-lane c2r pre-twiddle: R3's FFT input Zin[k], Zin[512 - k]
        auto synth_items = [&](cf* Y, const cf (&x)[5], const uint32_t (&qsv)[5], const uint32_t (&qav)[5]) {
            const int kk = k0, km = km0;
            cf* Yk = Y + kk;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                cf yk = pipe_rotate(x[2 * i], qsv[2 * i], qav[2 * i]);
                const cf ym = pipe_rotate(x[2 * i + 1], qsv[2 * i + 1], qav[2 * i + 1]);
                cf a = ym, b = yk;                            // index 512 - k: the roles of the two bins swap
                cf mk = ym;                                   // the partner of bin k
                if (i == 0 && h == 0) {
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
        };
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const int ia = kG * (t - 1) + j;                  // index (from f_first) of the frame analysed in this step
            const bool act_a = t >= 1 && ia < n;
            const long long fa = f_first + ia;
            const int ib = ia - kG;                           // kG > 1: the frame whose increment is formed in this step (H1)
            const bool act_b = kG > 1 && t >= 2 && ib < n;
            const long long fb = f_first + ib;
            const int ic = ib - kG;                           // kG > 1: the frame whose phase is advanced and that is rotated in this step (H2)
            const bool act_c = kG > 1 && t >= 3 && ic < n;
            const long long fc = f_first + ic;
            const int gen = t & 1, old = gen ^ 1;
            pipe_barrier();                                   // everything written in step t - 1 is complete
            const int kk = k0, km = km0;
            // ---- all of this step's reads in one round trip
            cf A[2], B[2], z256{0.0f, 0.0f};
            if (act_a) {
                const cf* Z = x_z(slot, old);
                const cf* Zk = Z + kk;
                const cf* Zm = Z + 512 - kk;
#pragma unroll
                for (int i = 0; i < 2; i++) { A[i] = lds_ld(Zk + 64 * i); B[i] = lds_ld(Zm - 64 * i); }
                if (h == 0) z256 = lds_ld(Z + 256);
            }
            uint32_t pv[5] = {0, 0, 0, 0, 0}, pnext[5] = {0, 0, 0, 0, 0}, iv[kG > 1 ? kG : 1][5];
            if (kG > 1) {
                if (j > 0) {
                    // the predecessor of H1 sits in the slot before, analysed in the same step as H1: one step ago
                    const uint32_t* pq = x_qa(slot - 1, old);
                    pv[0] = pq[kk]; pv[1] = pq[km]; pv[2] = pq[kk + 64]; pv[3] = pq[448 - kk];
                    pv[4] = pq[512];
                } else {
                    // slot 0: the predecessor of the frame analysed NOW is what the unit's last slot analysed one step ago; it is H1's
                    // predecessor in the next step, when that buffer is being overwritten — fetched now, kept in registers
                    const uint32_t* pq = x_qa(slot + kG - 1, old);
                    pnext[0] = pq[kk]; pnext[1] = pq[km]; pnext[2] = pq[kk + 64]; pnext[3] = pq[448 - kk];
                    pnext[4] = pq[512];
#pragma unroll
                    for (int q = 0; q < 5; q++) pv[q] = pcur[q];
                }
                if (act_c) {
#pragma unroll
                    for (int i2 = 0; i2 < kG; i2++) {
                        const uint32_t* pi = x_inc(slot - j + i2, old);
                        iv[i2][0] = pi[kk]; iv[i2][1] = pi[km]; iv[i2][2] = pi[kk + 64]; iv[i2][3] = pi[448 - kk];
                        iv[i2][4] = pi[512];
                    }
                }
            }
            // ---- analysis of frame fa: r2c split -> 2 X (the factor is undone in R3's output gain), phases
            cf va[5];
            uint32_t qa[5] = {0, 0, 0, 0, 0};
            if (act_a) {
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
#pragma unroll
                for (int q = 0; q < 4; q++) qa[q] = atan2_q32(va[q].y, va[q].x);
                // bin N/2 of a real signal is real: its phase is 0 or 1/2 turn by the sign of the real part
                if (h == 0) qa[4] = (va[4].x < 0.0f) ? 0x80000000u : 0u;
            }
            if (kG == 1) {
                if (act_a) {
                    if (fa >= b0) {
                        if (fa == 0) {
#pragma unroll
                            for (int q = 0; q < 5; q++) qs[q] += qa[q];
                        } else {
                            const unsigned d = (unsigned)(frame_start(p, fa) - frame_start(p, fa - 1));
                            const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
                            uint32_t inc[5];
                            inc_items(qa, qp, d, R, inc);
#pragma unroll
                            for (int q = 0; q < 5; q++) qs[q] += inc[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 5; q++) qp[q] = qa[q];
                    if (fa == p.carry_frame) carry_store(qs);
                    if (fa >= b0) synth_items(x_y(slot, gen), va, qs, qa);
                }
            } else {
                if (act_a) {
                    uint32_t* pq = x_qa(slot, gen);
                    pq[kk] = qa[0]; pq[km] = qa[1]; pq[kk + 64] = qa[2]; pq[448 - kk] = qa[3];
                    if (dc) pq[512] = qa[4];
                }
                // ---- increment of H1 (frame fb); zero for a frame that only primes the phase or lies beyond the tile
                uint32_t inc[5] = {0, 0, 0, 0, 0};
                if (act_b && fb >= b0) {
                    if (fb == 0) {
#pragma unroll
                        for (int q = 0; q < 5; q++) inc[q] = qa1[q];   // the "increment" of frame 0 is its analysis phase
                    } else {
                        const unsigned d = (unsigned)(frame_start(p, fb) - frame_start(p, fb - 1));
                        const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
                        inc_items(qa1, pv, d, R, inc);
                    }
                }
                {
                    uint32_t* pi = x_inc(slot, gen);
                    pi[kk] = inc[0]; pi[km] = inc[1]; pi[kk + 64] = inc[2]; pi[448 - kk] = inc[3];
                    if (dc) pi[512] = inc[4];
                }
                // ---- running phase of H2 (frame fc): the increments of the unit's slots up to this one; all of them move the base on
                if (act_c) {
                    uint32_t mine[5], base[5];
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
                    if (fc == p.carry_frame) carry_store(mine);
                    if (fc >= b0) synth_items(x_y(slot, gen), x2, mine, qa2);
                }
#pragma unroll
                for (int q = 0; q < 5; q++) { x2[q] = x1[q]; qa2[q] = qa1[q]; pcur[q] = pnext[q]; }
                if (act_a) {
#pragma unroll
                    for (int q = 0; q < 5; q++) { x1[q] = va[q]; qa1[q] = qa[q]; }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------ R3: synthesis
        float* optr = out.base + s_idx * out.ss + c * out.cs;
        const bool out_vec = (out.fs == 1) && ((reinterpret_cast<uintptr_t>(optr) & 15) == 0);
        // overlap-add, gain: kernels_pvpipe.hip (same order of the sums)
        constexpr float kGain = NAE_OLA_GAIN / 2048.0f;
        float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float y0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        cf r_w[8], r_ta[7], r_tb[7];                          // synthesis window and twiddles in registers
#pragma unroll
        for (int r = 0; r < 8; r++) r_w[r] = reinterpret_cast<const cf*>(tb.hann)[lane + 64 * r];
#pragma unroll
        for (int q = 0; q < 7; q++) { r_ta[q] = tb.w512[lane * (q + 1)]; r_tb[q] = tb.w512[8 * (lane & 7) * (q + 1)]; }
        bool had = false;                                     // kG > 1: a frame of this slot went through the previous step
        // Dense stores (kG = 1, interleaved stereo, 16-byte aligned stream).  A wave holds its channel's finished block as samples 2 lane, + 1 (first half) and
        // 128 + 2 lane, + 1 (second half); the wave of channel c stores HALF c of the interleaved block, 16 contiguous bytes per lane, 1 KiB per wave, instead of
        // four dwords that fill half of every 8 bytes.  So it keeps its own half c in two registers and leaves the other half where its partner finds it one
        // barrier later: entries 512..575 of the Y buffer it has just used up as FFT scratch.  Nobody else touches them: the transposes that reach up to entry
        // 567 are this wave's own and are over by then, the phase waves fill entries 0..511 of that buffer in the next step (while the partner reads), and
        // the buffer's next FFT — which overwrites them — is this wave's, two steps on.
        const bool dense = dense_shape && ((reinterpret_cast<uintptr_t>(optr - c) & 15) == 0);   // (optr - c: channel 0 of the stream)
        int pend_be = -1;                                     // block waiting for its other half (wave-uniform; the same in both channel waves)
        float keep0 = 0.0f, keep1 = 0.0f;
        auto store_block = [&](long long be, const float (&o)[4]) {
            if (be >= b0 && be < b_end && be * NAE_HOP < p.mid_len) {
                float* pb = optr + be * NAE_HOP * out.fs;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, -1, 0x00020000);
                auto st = [&](unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 0); };
                const int ls = lane;
                const unsigned fs4 = 4u * (unsigned)out.fs;                 // bytes between consecutive samples
                const unsigned oa = 2u * (unsigned)ls * fs4;                // sample 2 lane of the block
                if ((be + 1) * NAE_HOP <= p.mid_len) {
                    if (out_vec) {
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
        };
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const int iz = kG * (t - kR3Lag) + j;
            const long long fz = f_first + iz;
            const bool active = iz >= 0 && iz < n && fz >= b0;
            pipe_barrier();                                   // the FFT input of frame fz is complete
            cf zs[8];
            cf* Yi = x_y(slot, (t & 1) ^ 1);
            cf xh{0.0f, 0.0f};
            if (kG == 1 && pend_be >= 0) xh = lds_ld(x_y(slot ^ 1, t & 1) + 512 + lane);   // the partner's half c of the previous step's block
            // kG > 1: the three quarters that complete the previous step's block, in the same round trip as the FFT input
            float4 oq[3];
            const bool finish = kG > 1 && had && fz - kG - 3 >= b0;
            if (finish) {
#pragma unroll
                for (int i = 3; i >= 1; i--) {
                    const int rel = j - i;                                   // < 0: an earlier step
                    const int back = rel >= 0 ? 0 : (-rel + kG - 1) / kG;
                    const int sl = slot - j + rel + back * kG;
                    const int gen = (t - 1 - back + 2 * kGens) % (kGens > 0 ? kGens : 1);
                    oq[i - 1] = *reinterpret_cast<const float4*>(x_ola(sl, gen) + (i - 1) * kOlaQuarter + 4 * lane);
                }
            }
            if (active) {
                const int la = lane;
#pragma unroll
                for (int r = 0; r < 8; r++) zs[r] = lds_ld(Yi + la + 64 * r);
            }
            if (kG == 1 && pend_be >= 0) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(optr - c + (long long)pend_be * (2 * NAE_HOP), 0, -1, 0x00020000);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = c == 0 ? u32x4{__float_as_uint(keep0), __float_as_uint(xh.x), __float_as_uint(keep1), __float_as_uint(xh.y)}
                                       : u32x4{__float_as_uint(xh.x), __float_as_uint(keep0), __float_as_uint(xh.y), __float_as_uint(keep1)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * lane, 1024 * c, 0);
                pend_be = -1;
            }
            if (finish) {
                // quarters 3, 2, 1 of the three frames before it, then the own quarter 0 — the frame order
                float o[4] = {oq[2].x, oq[2].y, oq[2].z, oq[2].w};
                o[0] += oq[1].x; o[1] += oq[1].y; o[2] += oq[1].z; o[3] += oq[1].w;
                o[0] += oq[0].x; o[1] += oq[0].y; o[2] += oq[0].z; o[3] += oq[0].w;
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (o[i] + y0[i]) * kGain;
                store_block(fz - kG - 3, o);
            }
            had = active;
            if (active) {
                fft512_pad_a_tw(zs, r_ta);
                // scratch: the buffer just read (its next writer, R2 of step t + 1, is a barrier away)
                const FftLds L = make_fft_lds(Yi, nullptr, nullptr, lane);
                cf none[8];
                fft512_pad_bc_g<true, false>(zs, L, r_tb, nullptr, none);
                // zs[r] = conj(z[n]) * 512 (x 4), n = lane + 64 r  ->  time samples 2n, 2n+1, windowed
                float y[4][4];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const cf w = r_w[r];
                    y[r >> 1][2 * (r & 1)] = zs[r].x * w.x;
                    y[r >> 1][2 * (r & 1) + 1] = -(zs[r].y * w.y);   // the sign undoes the conjugation
                }
                if (kG == 1) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        o[i] = (r0[i] + y[0][i]) * kGain;
                        r0[i] = r1[i] + y[1][i];
                        r1[i] = r2[i] + y[2][i];
                        r2[i] = y[3][i];
                    }
                    const long long be = fz - 3;
                    if (dense && be >= b0 && be < b_end && (be + 1) * NAE_HOP <= p.mid_len) {
                        lds_st(Yi + 512 + lane, c == 0 ? cf{o[2], o[3]} : cf{o[0], o[1]});   // the half the partner stores
                        keep0 = c == 0 ? o[0] : o[2];
                        keep1 = c == 0 ? o[1] : o[3];
                        pend_be = (int)be;
                    } else {
                        store_block(be, o);                   // wave-uniform: the block's base pointer stays scalar
                    }
                } else {
                    float* po = x_ola(slot, t % (kGens > 0 ? kGens : 1)) + 4 * lane;
#pragma unroll
                    for (int q = 1; q < 4; q++) *reinterpret_cast<float4*>(po + (q - 1) * kOlaQuarter) = float4{y[q][0], y[q][1], y[q][2], y[q][3]};
#pragma unroll
                    for (int i = 0; i < 4; i++) y0[i] = y[0][i];
                }
            }
        }
    }
}

} // namespace nae

using namespace nae;

template <int kG>
static int flow_launch(nae_ctx* ctx, unsigned groups, const SigViewD& src, const PvParams& p, long long n_sc, const uint32_t* phase_ws,
                       const OutViewD& out, const Tables& tb, bool unit_stride)
{
    // more than 64 KiB of dynamic LDS needs the attribute: once per instantiation and DEVICE, so the flag lives in the context
    constexpr unsigned bit = 1u << (8 + (kG == 1 ? 0 : kG == 2 ? 1 : 2));
    if (!(ctx->pv_attr_done & bit)) {
        (void)nae_use_device(ctx);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_flow_kernel<true, kG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flow_lds(kG));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_flow_kernel<false, kG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flow_lds(kG));
        if (e != hipSuccess) return nae_check(ctx, e, "hipFuncSetAttribute(pv_flow_kernel)");
        ctx->pv_attr_done |= bit;
    }
    if (unit_stride)
        NAE_KLAUNCH(ctx, "pv_flow_kernel", (pv_flow_kernel<true, kG>), dim3(groups), dim3(kFlowThreads), flow_lds(kG), ctx->stream, src, p, n_sc, phase_ws, out, tb);
    else
        NAE_KLAUNCH(ctx, "pv_flow_kernel", (pv_flow_kernel<false, kG>), dim3(groups), dim3(kFlowThreads), flow_lds(kG), ctx->stream, src, p, n_sc, phase_ws, out, tb);
    return NAE_OK;
}

// frames_per_step: 1 = one stream-channel per slot; 2 / 4 = frame-interleaved (two / one stream-channel per four slots).  For launches of at
// most one workgroup per CU (the caller's choice: nae_launch_pv_pipe sends them here)
int nae_launch_pv_flow(nae_ctx* ctx, const PvParams& p, const SigViewD& src, long long n_sc, const uint32_t* phase_ws,
                       const OutViewD& out, bool unit_stride, int frames_per_step)
{
    const long long items = n_sc * p.n_tiles;
    if (items == 0) return NAE_OK;
    if (frames_per_step != 1 && frames_per_step != 2 && frames_per_step != 4) return nae_fail(ctx, NAE_ERR_INVALID, "pv_flow_kernel: frames per step");
    const int units = kFlowSlots / frames_per_step;
    const long long groups = (items + units - 1) / units;
    if (groups > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "pv_flow_kernel: grid too large");
    Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    int rc;
    if (groups > (long long)ctx->n_cu) return nae_fail(ctx, NAE_ERR_INVALID, "pv_flow_kernel: more than one workgroup per CU");
    if (frames_per_step == 1) rc = flow_launch<1>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    else if (frames_per_step == 2) rc = flow_launch<2>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    else rc = flow_launch<4>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    if (rc) return rc;
    return nae_check(ctx, hipGetLastError(), "pv_flow_kernel");
}
