// kernels_pvpipe.hip — pass 3 of the phase vocoder (K7) as a four-role wave pipeline for gfx950, in two schedules.
//
// Why a pipeline.  On gfx950 one wave issues at most one vector instruction per 4.5-5 cycles, while a SIMD with two or more
// resident waves retires one per 2.15 cycles (profiles/r05_valu_wallclock.md) — and more waves hide the LDS round trips and barrier
// waits.  A stream-channel of the vocoder is a serial chain of frames (integer phase accumulator, overlap-add), so "one wave per
// stream-channel" leaves most issue slots empty.  Here a frame passes through FOUR waves, one per role (bodies: pv_roles.h), that hand
// it on through LDS once per step:
//
//   R1   load, Hann window, forward FFT                                                                  -> Z   (natural order)
//   R2a  bins of half 0 (+ 256, 512): r2c split, atan2 -> Q0.32, exact phase advance, rotation, c2r pre-twiddle -> Y   (R3's FFT input)
//   R2b  the same for the bins of half 1
//   R3   inverse FFT (by forward FFT), synthesis window, overlap-add, store of the finished hop block
//
// Slots.  kG = 1: the four slots of a workgroup are four stream-channels (two stereo streams) — the regime of large batches
// (>= 1024 stream-channels: one tile per stream-channel, no pass 1, nothing analysed twice).
// kG = 2 / 4 ("frame-interleaved"): the slots of a unit are kG CONSECUTIVE FRAMES of ONE stream-channel and a step advances it by kG
// frames.  FFTs, split, atan2, rotation are independent per frame; what is sequential is
//   * the Q0.32 phase: frame f needs Qa of f-1 and Qs of f-1.  R2 analyses frame f and leaves Qa in LDS; a step later it fetches its
//     predecessor's Qa, forms its increment (exact integers), leaves it in LDS, and behind the next barrier every R2 wave sums the
//     increments of the slots up to its own onto the running phase (one add per bin);
//   * the overlap-add: a hop block sums quarters of four consecutive frames = four different slots.  R3 leaves quarters 1..3 of its
//     windowed frame in LDS and, one step later, the wave that holds the block's LAST frame adds them in frame order — the order of
//     the kG = 1 registers and of the oracle, so all modes give the same samples.
// This is what small batches run (the 128 streams one rank of an 8-GPU job owns are 256 stream-channels = one workgroup per CU): no
// time tiles, no pass 1, no second analysis.  Fewer stream-channels still get tiles (nae_pick_pv_shape).
//
// Schedules.
//   pv_pipe_kernel — TWO barriers per step, ONE buffer per hand-off: after A the consumers read what the producers left in the step
//     before into registers; after B the producers overwrite.  One slot costs 2 x 4608 B (FFT scratch of R1 / R3) + 4160 B (Y) of LDS:
//     every role fits 64 VGPRs, a 1024-thread workgroup is 4 slots x 4 roles, two workgroups fill a CU (32 waves, 8 per SIMD — each
//     SIMD hosts one wave of every role of two slots).  The headline batch (2048 stream-channels) runs this.
//   pv_flow_kernel — ONE barrier per step, every hand-off buffer TWICE: in step t a producer writes generation t & 1 while its
//     consumer reads generation (t - 1) & 1, and between two barriers every wave does a whole step's work of its role (270 to 330
//     vector instructions: level).  For launches of at most one workgroup per CU (the 128 to 512 streams a rank of a multi-GPU job
//     owns), where the two-barrier step leaves the vector ALU busy 54 % of the time (profiles/r05_pipe_stamps_128_streams.txt): a CU
//     with one workgroup has the LDS to spare (144 / 156 KB of 160).  The price is pipeline depth (kG > 1: a frame needs 5 steps
//     instead of 4 to leave, of ~470).  Shipped for one frame per step (512 streams: 3.5 % faster); in the frame-interleaved shapes it
//     saves 3.5 % of the cycles and the chip returns them as a lower clock (profiles/r05_flow.md).
//     (A 64-VGPR build of it for the full batch — two workgroups per CU at 80 KiB each — was measured: 6.39-6.44 against 6.10-6.18 ms.)
//
// The arithmetic — and therefore every integer phase — is the canonical one of DESIGN.md §3 (same dft8_fwd / cmul_tw / atan2_q32 /
// phase increment as the other kernels); only where data waits between operations differs between the schedules and shapes.
// Replaces: SoundTouch behind /root/reference/src/processor/audio-velocity.cpp:369-428 (algorithm differs: DESIGN.md §3).
#include "stft_common.h"
#include "pv_roles.h"

namespace nae {

constexpr int kPipeSlots = 4;                                 // slots per workgroup: waves [0,4) = R1, [4,8) = R2a, [8,12) = R2b, [12,16) = R3
constexpr int kPipeThreads = 64 * 4 * kPipeSlots;
constexpr size_t kPipeLdsTables = NAE_FFT_N * sizeof(float) + (kT1024Pad + 64 + kTwaCf) * sizeof(cf);

// ======================================================================================================= two barriers per step
constexpr size_t kPipeLdsPerSlot = (2 * kPadScratchCf + kYCf) * sizeof(cf);
// frame-interleaved modes: per slot, analysis phases of the last two steps, one phase increment, and the quarters 1..3 of
// the windowed frames of the last kOlaGens steps (a block's oldest frame lies ceil(3 / kG) steps back)
constexpr int pipe_ola_gens(int kG) { return kG >= 3 ? 2 : 3; }
// kG = 1: 1 KiB per slot — the two channel waves of a stereo stream (adjacent slots) exchange their finished hop blocks there, so that each can write
// one DENSE 1-KiB piece of the interleaved output (16 bytes per lane) instead of four dword stores that fill a quarter of every 16 bytes
// (Measured for the two-frames-per-step shape too — exchange behind barrier A, stores behind barrier B: 2.14-2.19 against 2.10-2.11 ms at 256 streams: in
// that latency-bound regime the extra LDS round trip costs more than the denser stores give; four frames per step hold ONE channel per workgroup.)
constexpr size_t kPipeXchgPerSlot = 1024;
constexpr size_t pipe_lds_x_per_slot(int kG) { return kG == 1 ? kPipeXchgPerSlot : 3 * kPhasePad * sizeof(uint32_t) + (size_t)pipe_ola_gens(kG) * 3 * kOlaQuarter * sizeof(float); }
constexpr size_t pipe_lds(int kG) { return kPipeLdsTables + kPipeSlots * (kPipeLdsPerSlot + pipe_lds_x_per_slot(kG)); }
static_assert(2 * pipe_lds(1) <= 160 * 1024, "two workgroups per CU");
static_assert(pipe_lds(2) <= 160 * 1024 && pipe_lds(4) <= 160 * 1024, "one workgroup per CU");

// Issue priority (kG = 1).  Two workgroups share a CU and the hardware arbitrates equal priorities by age, so the workgroup that
// arrived first runs its steps 30-40 % faster than its neighbour on every CU, finishes early and leaves the CU half empty
// (profiles/r02_pipe_stamps_per_cu.txt, r03_pipe_stamps_1024_streams.txt).  An uneven pair is not the problem — the favoured
// workgroup runs a step in 4800 cycles against 4140 alone on the CU while its neighbour still advances: more frames per cycle
// than two workgroups at equal priority — the early finish is.  So the two take turns at the higher priority in TIME SLICES of
// 2^18 shader cycles (0.13 ms, ~50 steps): both read the same clock (s_memtime, requested behind barrier B and used behind
// the next barrier A, where the wave has waited for lgkmcnt(0) anyway), a workgroup learns whether it was the first or the second
// on its CU from an arrival counter per physical CU (never reset: only the parity is used), and the two progress at the same
// average pace.  The role that ends a step's critical path most often (R3) sits one level above its workgroup's.
// Measured on one box (vocoder kernel, ms): step-parity turns 6.56, slices of 2^12 cycles 6.6, 2^15 6.38, 2^16 6.26, 2^18 6.19,
// 2^19 6.19, 2^21 6.31, 2^23 6.75; with 2^18: no role up 6.50, R1 and R3 up 6.33, R2a / R2b up 6.69, R1 up 6.61, R3 up 6.31.
// Measured and dropped: keeping the pair level by feedback (each workgroup publishing its step counter, whoever is behind at the
// higher priority: both then run at the pace of the slower, 7.8 against 7.3 ms); two-slot (eight-wave) workgroups, four per CU
// (within 1 %).
__device__ unsigned g_cu_arrivals[8 * 4 * 16];

// kRich: at most one workgroup per CU anyway -> 128 VGPRs per wave, window and twiddles in registers (30 LDS reads less per frame and one round trip
// less on the step's critical path)
// (Pass 1 — the per-tile sums of the phase increments, kernels_stft.hip — was tried on this pipeline too, with R3 idle and nothing rotated: on one hour of stereo
// 1.69-1.77 ms against the one-wave-per-tile kernel's 1.38-1.40: a step's two barriers and LDS round trips cost the same with half the arithmetic.
// profiles/r06_pass1.md.)
template <bool kUnit, int kG, bool kRich>
__global__ __launch_bounds__(kPipeThreads, kRich ? 4 : 8) void pv_pipe_kernel(SigViewD src, PvParams p, long long n_sc,
                                                                            const uint32_t* __restrict__ base_phase, OutViewD out, Tables tb)
{
    static_assert(kPipeSlots % kG == 0, "a unit's frames share a workgroup");
    static_assert(kG == 1 || kRich, "the frame-interleaved modes run one workgroup per CU");
    constexpr int kUnits = kPipeSlots / kG;                  // stream-channels (x tile) per workgroup
    constexpr int kDepth = kG == 1 ? 2 : 4;                  // steps a frame needs beyond its R1 step
    constexpr int kGens = pipe_ola_gens(kG);
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
    if (kG == 1 && threadIdx.x == 0) s_slot = pipe_cu_arrival(g_cu_arrivals);
    __syncthreads();
    const int prio_slot = kG == 1 ? __builtin_amdgcn_readfirstlane(s_slot) : 0;

    const int wave = wave_id();
    const int role = wave / kPipeSlots, slot = wave % kPipeSlots;   // scalars
    const int unit = slot / kG, j = slot % kG;                      // j: frame of the step this slot works on (kG > 1)
    const int lane = threadIdx.x & 63;
    PipeItem it;
    if (!pipe_item(p, (long long)kUnits * blockIdx.x + unit, n_sc, it)) return;
    const int c = it.c, n = it.n;
    const long long sc = it.sc, b0 = it.b0, b_end = it.b_end, f_first = it.f_first;

    cf* S1 = reinterpret_cast<cf*>(smem + kPipeLdsTables + slot * kPipeLdsPerSlot);
    cf* Y = S1 + kPadScratchCf;
    cf* S3 = Y + kYCf;
    // frame-interleaved exchange areas, indexed by slot: QA[2][520] | INC[520] | OL[kGens][3][256]
    unsigned char* xbase = smem + kPipeLdsTables + kPipeSlots * kPipeLdsPerSlot;
    constexpr size_t kXPer = pipe_lds_x_per_slot(kG);
    auto x_qa = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(xbase + sl * kXPer) + gen * kPhasePad; };
    auto x_inc = [&](int sl) { return reinterpret_cast<uint32_t*>(xbase + sl * kXPer) + 2 * kPhasePad; };
    auto x_ola = [&](int sl, int gen) { return reinterpret_cast<float*>(xbase + sl * kXPer + 3 * kPhasePad * sizeof(uint32_t)) + gen * 3 * kOlaQuarter; };

    const int steps = (n + kG - 1) / kG;
    // interleaved stereo output, one frame per step: a finished block leaves one barrier later, through the exchange area (R3) — one more step for everybody
    const bool dense_shape = kG == 1 && p.ch == 2 && out.fs == 2 && out.cs == 1;
    const int T = steps + kDepth + (dense_shape ? 1 : 0);
    /*pipe:begin*/
    unsigned long long now = 0;                              // shader clock, read behind barrier B, used behind the next barrier A
    if (kG > 1 && role == 2) __builtin_amdgcn_s_setprio(1);   // frame-interleaved (one workgroup per CU): R2b one level up (1.10 against 1.15 ms at 128 streams)

    if (role == 0) {
        // ------------------------------------------------------------------------------------------ R1: analysis FFT
        ChanView in{src.base + it.s_idx * src.ss + c * src.cs, src.fs, p.in_len};
        cf nxt[8], va[8];
        if (j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + j), lane);
        FftRegs R;
        if (kRich) load_fft_regs_lds(R, hann, twa, w64, lane);
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const bool cur = kG * t + j < n;
            pipe_barrier();                                   /*A*/
            if (kG == 1) pipe_prio(prio_slot, role, now);
            // register-only part while R2 reads Z of the previous step out of this wave's scratch
            if (cur) r1_window_pass_a<kRich>(va, nxt, R, hann, twa, lane);
            pipe_barrier();                                   /*B*/
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            if (cur) {
                const FftLds L = make_fft_lds(S1, twa, w64, pipe_lane<!kRich>(lane));
                cf none[8];
                fft512_pad_bc_g<kRich, false>(va, L, R.tb, nullptr, none);
                r1_store_z(S1, L, va, lane);
                // request the next frame now: the loads land while the wave waits at the barriers
                if (kG * (t + 1) + j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + kG * (t + 1) + j), lane);
            }
        }
        /*pipe:r1-end*/
    } else if (role <= 2) {
        // ------------------------------------------------------------------------------------------ R2a / R2b: phases
        PhaseLane P;
        P.init(tb, role - 1, lane);
        uint32_t qs[5], qp[5];                                // [4]: bin 512 (h = 0)
        P.load_base(qs, base_phase, p, it);
#pragma unroll
        for (int q = 0; q < 5; q++) qp[q] = 0;
        cf hx[5];                                             // kG > 1: the frame analysed in the previous step ([4]: bin 512)
        uint32_t hqa[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 5; q++) hx[q] = cf{0.0f, 0.0f};
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const int ia = kG * (t - 1) + j;                  // index (from f_first) of the frame analysed in this step
            const bool act_a = t >= 1 && ia < n;
            const long long fa = f_first + ia;
            const int ib = ia - kG;                           // kG > 1: the frame whose phase is advanced in this step
            const bool act_b = kG > 1 && t >= 2 && ib < n;
            const long long fb = f_first + ib;
            pipe_barrier();                                   /*A*/  // Z of frame fa is complete
            if (kG == 1) pipe_prio(prio_slot, role, now);
            cf va[5];
            if (act_a) {
                cf A[2], B[2], z256{0.0f, 0.0f};
                P.read_z(S1, A, B, z256);
                P.split(A, B, z256, va);
            }
            if (kG > 1) {
                // increment of the held frame fb: its predecessor's analysis phases were left in LDS one step ago (slot j-1), or
                // two steps ago by the last slot (j = 0)
                uint32_t inc[5] = {0, 0, 0, 0, 0};
                if (act_b && fb >= b0) {
                    uint32_t pv[5] = {0, 0, 0, 0, 0};
                    if (fb != 0) P.read5(j > 0 ? x_qa(slot - 1, (t - 1) & 1) : x_qa(slot + kG - 1, t & 1), pv);
                    P.inc_of_frame(p, fb, hqa, pv, inc);
                }
                P.write5(x_inc(slot), inc);
            }
            pipe_barrier();                                   /*B*/  // R1 may overwrite its scratch
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            uint32_t qa[5] = {0, 0, 0, 0, 0};
            if (act_a) P.phases(va, qa);
            if (kG == 1) {
                if (act_a) {
                    // (written out rather than through PhaseLane::inc_of_frame: the same instructions, but this form keeps the headline kernel's schedule — 0.5 % of it)
                    if (fa >= b0) {
                        if (fa == 0) {
#pragma unroll
                            for (int q = 0; q < 5; q++) qs[q] += qa[q];
                        } else {
                            const unsigned d = (unsigned)(frame_start(p, fa) - frame_start(p, fa - 1));
                            const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
                            uint32_t inc[5];
                            P.inc_items(qa, qp, d, R, inc);
#pragma unroll
                            for (int q = 0; q < 5; q++) qs[q] += inc[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 5; q++) qp[q] = qa[q];
                    if (fa == p.carry_frame) P.carry_store(p, sc, qs);
                    if (fa >= b0) P.synth_items(Y, va, qs, qa);
                }
            } else {
                if (act_a) P.write5(x_qa(slot, t & 1), qa);
                if (act_b) {
                    // running phase: the increments of the unit's slots up to this one; all of them move the base on
                    uint32_t iv[kG][5], mine[5];
#pragma unroll
                    for (int i2 = 0; i2 < kG; i2++) P.read5(x_inc(slot - j + i2), iv[i2]);
                    r2_running_phase<kG>(qs, mine, iv, j);
                    if (fb == p.carry_frame) P.carry_store(p, sc, mine);
                    if (fb >= b0) P.synth_items(Y, hx, mine, hqa);
                }
                if (act_a) {
#pragma unroll
                    for (int q = 0; q < 5; q++) { hx[q] = va[q]; hqa[q] = qa[q]; }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------ R3: synthesis
        float* optr = out.base + it.s_idx * out.ss + c * out.cs;
        const BlockOut bo{optr, out.fs, (out.fs == 1) && ((reinterpret_cast<uintptr_t>(optr) & 15) == 0)};
        float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float y0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        FftRegs R;                                            // kRich: synthesis window and twiddles in registers
        if (kRich) load_fft_regs_lds(R, hann, twa, w64, lane);
        bool had = false;                                     // kG > 1: a frame of this slot went through the previous step
        // dense stores (kG = 1, interleaved stereo, 16-byte aligned stream): XB[channel][half][lane] = the lane's two sample pairs of the block
        const bool dense = dense_shape && ((reinterpret_cast<uintptr_t>(optr - c) & 15) == 0);   // (optr - c: channel 0 of the stream)
        int pend_be = -1;                                     // block waiting in the exchange area (wave-uniform; the same in both channel waves)
        auto xchg = [&]() { return reinterpret_cast<cf*>(xbase + (size_t)(slot & ~1) * kPipeXchgPerSlot); };
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const int iz = kG * (t - (kDepth - (kG == 1 ? 0 : 1))) + j;      // kG = 1: t - 2;  kG > 1: t - 3
            const long long fz = f_first + iz;
            const bool active = iz >= 0 && iz < n && fz >= b0;
            pipe_barrier();                                   /*A*/  // the FFT input of frame fz is complete
            if (kG == 1) pipe_prio(prio_slot, role, now);
            if (kG == 1 && pend_be >= 0) {
                // both channels' blocks of the previous step are in XB: this wave writes half c of the interleaved block
                const int lx = pipe_lane<!kRich>(lane);
                const cf* XB = xchg();
                r3_store_dense_half(optr - c, pend_be, c, lds_ld(XB + (0 + c) * 64 + lx), lds_ld(XB + (2 + c) * 64 + lx), lx);
                pend_be = -1;
            }
            cf zs[8];
            // kG > 1: the three quarters that complete the previous step's block are requested in the SAME round trip as the FFT input (round 5: read
            // behind pass A they cost the light barrier interval a second LDS round trip: R3 was the role it waited for, 1650 against R2a's 1400 cycles)
            float4 oq[3];
            const bool finish = kG > 1 && had && fz - kG - 3 >= b0;
            if (finish) {
                const int lq = pipe_lane<!kRich>(lane);
#pragma unroll
                for (int i = 3; i >= 1; i--) {
                    int sl, gen;
                    r3_quarter_source<kG>(slot, j, i, t, kGens, sl, gen);
                    oq[i - 1] = *reinterpret_cast<const float4*>(x_ola(sl, gen) + (i - 1) * kOlaQuarter + 4 * lq);
                }
            }
            if (active) {
                // FFT input and pass-A twiddles in one round trip; pass A is register-only, so it runs on this side of barrier B
                const int la = pipe_lane<!kRich>(lane);
                const cf* Zi = Y + la;
                const cf* ta = twa + la;
#pragma unroll
                for (int r = 0; r < 8; r++) zs[r] = lds_ld(Zi + 64 * r);
                if (kRich) {
                    fft512_pad_a_tw(zs, R.ta);
                } else {
                    cf tw[7];
#pragma unroll
                    for (int q = 0; q < 7; q++) tw[q] = lds_ld(ta + 64 * q);
                    fft512_pad_a_tw(zs, tw);
                }
            }
            if (finish) {
                float o[4];
                r3_ola_finish(oq, y0, o);
                r3_store_block(p, it, bo, fz - kG - 3, o, pipe_lane<!kRich>(lane));
            }
            pipe_barrier();                                   /*B*/  // R2 may overwrite the FFT input
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            had = active;
            if (active) {
                const int lb = pipe_lane<!kRich>(lane);
                const FftLds L = make_fft_lds(S3, twa, w64, lb);
                const cf* hw = reinterpret_cast<const cf*>(hann) + lb;
                cf wn[8];                                     // synthesis window: requested behind the second transpose
                fft512_pad_bc_g<kRich, !kRich>(zs, L, R.tb, hw, wn);
                if (kRich) {
#pragma unroll
                    for (int r = 0; r < 8; r++) wn[r] = R.w[r];
                }
                float y[4][4];
                r3_window(zs, wn, y);
                if (kG == 1) {
                    float o[4];
                    r3_ola_shift(r0, r1, r2, y, o);
                    const long long be = fz - 3;
                    if (dense && be >= b0 && be < b_end && (be + 1) * NAE_HOP <= p.mid_len) {
                        const int lx = pipe_lane<!kRich>(lane);
                        cf* XB = xchg();
                        lds_st(XB + (2 * c + 0) * 64 + lx, cf{o[0], o[1]});
                        lds_st(XB + (2 * c + 1) * 64 + lx, cf{o[2], o[3]});
                        pend_be = (int)be;
                    } else {
                        r3_store_block(p, it, bo, be, o, pipe_lane<!kRich>(lane));   // wave-uniform: the block's base pointer stays scalar
                    }
                } else {
                    float* po = x_ola(slot, t % kGens) + 4 * lb;
#pragma unroll
                    for (int q = 1; q < 4; q++) *reinterpret_cast<float4*>(po + (q - 1) * kOlaQuarter) = float4{y[q][0], y[q][1], y[q][2], y[q][3]};
#pragma unroll
                    for (int i = 0; i < 4; i++) y0[i] = y[0][i];
                }
            }
        }
    }
}

// ======================================================================================================= one barrier per step
//   step t, slot j of a unit:
//     R1   frame kG t + j        window, forward FFT (scratch = its own Z buffer of generation t)             -> Z[t]
//     R2   frame kG (t-1) + j    r2c split, atan2 -> Q0.32                                                       -> QA[t]   (kG > 1)
//          frame kG (t-2) + j    exact phase increment against the predecessor's analysis phase QA[t-1]          -> INC[t]  (kG > 1)
//          frame kG (t-3) + j    running phase = base + the unit's increments INC[t-1] up to slot j; rotation;
//                                c2r pre-twiddle                                                                  -> Y[t]
//     R3   frame kG (t-4) + j    inverse FFT (scratch = the Y buffer it has just read), synthesis window;
//                                quarters 1..3 of the windowed frame                                              -> OLA[t]  (kG > 1)
//          frame kG (t-5) + j    the hop block this frame completed: quarters of the three frames before it (OLA[t-1], OLA[t-2], ...)
//                                in frame order + its own quarter 0 (registers), gain, store
//   kG = 1: R2 analyses, advances and rotates frame t - 1 in one step (phases stay in registers), R3 works on frame t - 2 and keeps the
//   three open hop blocks in registers — the two-barrier schedule's kG = 1 with one barrier.
constexpr int flow_ola_gens(int kG) { return kG == 1 ? 0 : kG == 2 ? 4 : 3; }     // generations read (t-1 ... t-1-ceil(3/kG)) + the one written
constexpr size_t kFlowBuf = kPadScratchCf * sizeof(cf);                             // one Z / Y generation (doubles as FFT scratch)
constexpr size_t flow_lds_per_slot(int kG)
{
    return 4 * kFlowBuf + (kG == 1 ? 0 : 4 * kPhasePad * sizeof(uint32_t) + (size_t)flow_ola_gens(kG) * 3 * kOlaQuarter * sizeof(float));
}
constexpr size_t flow_lds(int kG) { return kPipeSlots * flow_lds_per_slot(kG); }
static_assert(flow_lds(1) <= 160 * 1024 && flow_lds(2) <= 160 * 1024 && flow_lds(4) <= 160 * 1024, "one workgroup per CU");

template <bool kUnit, int kG>
__global__ __launch_bounds__(kPipeThreads, 4) void pv_flow_kernel(SigViewD src, PvParams p, long long n_sc, const uint32_t* __restrict__ base_phase,
                                                                  OutViewD out, Tables tb)
{
    static_assert(kPipeSlots % kG == 0, "a unit's frames share a workgroup");
    constexpr int kUnits = kPipeSlots / kG;                  // stream-channels (x tile) per workgroup
    constexpr int kGens = flow_ola_gens(kG);
    constexpr size_t kPer = flow_lds_per_slot(kG);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave = wave_id();
    const int role = wave / kPipeSlots, slot = wave % kPipeSlots;   // scalars
    const int unit = slot / kG, j = slot % kG;                      // j: frame of the step this slot works on
    const int lane = threadIdx.x & 63;
    PipeItem it;
    if (!pipe_item(p, (long long)kUnits * blockIdx.x + unit, n_sc, it)) return;
    const int c = it.c, n = it.n;
    const long long sc = it.sc, b0 = it.b0, b_end = it.b_end, f_first = it.f_first;

    // per slot: Z[2] | Y[2] | QA[2] | INC[2] | OL[kGens][3][256]
    auto slot_base = [&](int sl) { return smem + sl * kPer; };
    auto x_z = [&](int sl, int gen) { return reinterpret_cast<cf*>(slot_base(sl)) + gen * kPadScratchCf; };
    auto x_y = [&](int sl, int gen) { return reinterpret_cast<cf*>(slot_base(sl)) + (2 + gen) * kPadScratchCf; };
    auto x_qa = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(slot_base(sl) + 4 * kFlowBuf) + gen * kPhasePad; };
    auto x_inc = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(slot_base(sl) + 4 * kFlowBuf) + (2 + gen) * kPhasePad; };
    auto x_ola = [&](int sl, int gen) {
        return reinterpret_cast<float*>(slot_base(sl) + 4 * kFlowBuf + 4 * kPhasePad * sizeof(uint32_t)) + gen * 3 * kOlaQuarter;
    };

    const int steps = (n + kG - 1) / kG;
    constexpr int kR3Lag = kG == 1 ? 2 : 4;                            // steps between a frame's R1 step and its R3 step
    // interleaved stereo output, one frame per step: a finished block leaves one barrier later, as a dense piece assembled with the other channel's wave (R3)
    const bool dense_shape = kG == 1 && p.ch == 2 && out.fs == 2 && out.cs == 1;
    const int T = steps + kR3Lag + (kG == 1 ? (dense_shape ? 1 : 0) : 1);   // kG > 1: a block leaves one step behind its last frame
    /*pipe:begin*/

    if (role == 0) {
        // ------------------------------------------------------------------------------------------ R1: analysis FFT
        ChanView in{src.base + it.s_idx * src.ss + c * src.cs, src.fs, p.in_len};
        cf nxt[8], va[8];
        if (j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + j), lane);
        FftRegs R;                                            // window and both twiddle sets stay in registers (128 VGPRs per wave)
        load_fft_regs_global(R, tb, lane);
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const bool cur = kG * t + j < n;
            pipe_barrier();                                   /*A*/
            if (cur) {
                r1_window_pass_a<true>(va, nxt, R, nullptr, nullptr, lane);
                cf* Z = x_z(slot, t & 1);
                const FftLds L = make_fft_lds(Z, nullptr, nullptr, lane);
                cf none[8];
                fft512_pad_bc_g<true, false>(va, L, R.tb, nullptr, none);
                r1_store_z(Z, L, va, lane);
                // request the next frame now: the loads land while the wave waits at the barrier
                if (kG * (t + 1) + j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + kG * (t + 1) + j), lane);
            }
        }
        /*pipe:r1-end*/
    } else if (role <= 2) {
        // ------------------------------------------------------------------------------------------ R2a / R2b: phases
        PhaseLane P;
        P.init(tb, role - 1, lane);
        uint32_t qs[5], qp[5];                                // [4]: bin 512 (h = 0)
        P.load_base(qs, base_phase, p, it);
#pragma unroll
        for (int q = 0; q < 5; q++) qp[q] = 0;
        // kG > 1: H1 = the frame analysed one step ago (waits for its increment), H2 = two steps ago (waits for the running phase)
        cf x1[5], x2[5];
        uint32_t qa1[5] = {0, 0, 0, 0, 0}, qa2[5] = {0, 0, 0, 0, 0};
        uint32_t pcur[5] = {0, 0, 0, 0, 0};                   // slot 0 of a unit: the analysis phases of the frame before H1 (the unit's last slot, a step earlier)
#pragma unroll
        for (int q = 0; q < 5; q++) { x1[q] = cf{0.0f, 0.0f}; x2[q] = cf{0.0f, 0.0f}; }
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
            pipe_barrier();                                   /*A*/  // everything written in step t - 1 is complete
            // ---- all of this step's reads in one round trip
            cf A[2], B[2], z256{0.0f, 0.0f};
            if (act_a) P.read_z(x_z(slot, old), A, B, z256);
            uint32_t pv[5] = {0, 0, 0, 0, 0}, pnext[5] = {0, 0, 0, 0, 0}, iv[kG > 1 ? kG : 1][5];
            if (kG > 1) {
                if (j > 0) {
                    // the predecessor of H1 sits in the slot before, analysed in the same step as H1: one step ago
                    P.read5(x_qa(slot - 1, old), pv);
                } else {
                    // slot 0: the predecessor of the frame analysed NOW is what the unit's last slot analysed one step ago; it is H1's
                    // predecessor in the next step, when that buffer is being overwritten — fetched now, kept in registers
                    P.read5(x_qa(slot + kG - 1, old), pnext);
#pragma unroll
                    for (int q = 0; q < 5; q++) pv[q] = pcur[q];
                }
                if (act_c) {
#pragma unroll
                    for (int i2 = 0; i2 < kG; i2++) P.read5(x_inc(slot - j + i2, old), iv[i2]);
                }
            }
            // ---- analysis of frame fa
            cf va[5];
            uint32_t qa[5] = {0, 0, 0, 0, 0};
            if (act_a) {
                P.split(A, B, z256, va);
                P.phases(va, qa);
            }
            if (kG == 1) {
                if (act_a) {
                    if (fa >= b0) {
                        uint32_t inc[5];
                        P.inc_of_frame(p, fa, qa, qp, inc);
#pragma unroll
                        for (int q = 0; q < 5; q++) qs[q] += inc[q];
                    }
#pragma unroll
                    for (int q = 0; q < 5; q++) qp[q] = qa[q];
                    if (fa == p.carry_frame) P.carry_store(p, sc, qs);
                    if (fa >= b0) P.synth_items(x_y(slot, gen), va, qs, qa);
                }
            } else {
                if (act_a) P.write5(x_qa(slot, gen), qa);
                // ---- increment of H1 (frame fb); zero for a frame that only primes the phase or lies beyond the tile
                uint32_t inc[5] = {0, 0, 0, 0, 0};
                if (act_b && fb >= b0) P.inc_of_frame(p, fb, qa1, pv, inc);
                P.write5(x_inc(slot, gen), inc);
                // ---- running phase of H2 (frame fc): the increments of the unit's slots up to this one; all of them move the base on
                if (act_c) {
                    uint32_t mine[5];
                    r2_running_phase<(kG > 1 ? kG : 1)>(qs, mine, iv, j);
                    if (fc == p.carry_frame) P.carry_store(p, sc, mine);
                    if (fc >= b0) P.synth_items(x_y(slot, gen), x2, mine, qa2);
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
        float* optr = out.base + it.s_idx * out.ss + c * out.cs;
        const BlockOut bo{optr, out.fs, (out.fs == 1) && ((reinterpret_cast<uintptr_t>(optr) & 15) == 0)};
        float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float y0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        FftRegs R;                                            // synthesis window and twiddles in registers
        load_fft_regs_global(R, tb, lane);
        bool had = false;                                     // kG > 1: a frame of this slot went through the previous step
        // Dense stores (kG = 1, interleaved stereo, 16-byte aligned stream).  A wave holds its channel's finished block as samples 2 lane, + 1 (first half) and
        // 128 + 2 lane, + 1 (second half); the wave of channel c stores HALF c of the interleaved block.  So it keeps its own half c in two registers and leaves
        // the other half where its partner finds it one barrier later: entries 512..575 of the Y buffer it has just used up as FFT scratch.  Nobody else touches
        // them: the transposes that reach up to entry 567 are this wave's own and are over by then, the phase waves fill entries 0..511 of that buffer in the next
        // step (while the partner reads), and the buffer's next FFT — which overwrites them — is this wave's, two steps on.
        const bool dense = dense_shape && ((reinterpret_cast<uintptr_t>(optr - c) & 15) == 0);   // (optr - c: channel 0 of the stream)
        int pend_be = -1;                                     // block waiting for its other half (wave-uniform; the same in both channel waves)
        cf keep{0.0f, 0.0f};
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const int iz = kG * (t - kR3Lag) + j;
            const long long fz = f_first + iz;
            const bool active = iz >= 0 && iz < n && fz >= b0;
            pipe_barrier();                                   /*A*/  // the FFT input of frame fz is complete
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
                    int sl, gen;
                    r3_quarter_source<kG>(slot, j, i, t, kGens > 0 ? kGens : 1, sl, gen);
                    oq[i - 1] = *reinterpret_cast<const float4*>(x_ola(sl, gen) + (i - 1) * kOlaQuarter + 4 * lane);
                }
            }
            if (active) {
#pragma unroll
                for (int r = 0; r < 8; r++) zs[r] = lds_ld(Yi + lane + 64 * r);
            }
            if (kG == 1 && pend_be >= 0) {
                r3_store_dense_half(optr - c, pend_be, c, c == 0 ? keep : xh, c == 0 ? xh : keep, lane);
                pend_be = -1;
            }
            if (finish) {
                float o[4];
                r3_ola_finish(oq, y0, o);
                r3_store_block(p, it, bo, fz - kG - 3, o, lane);
            }
            had = active;
            if (active) {
                fft512_pad_a_tw(zs, R.ta);
                // scratch: the buffer just read (its next writer, R2 of step t + 1, is a barrier away)
                const FftLds L = make_fft_lds(Yi, nullptr, nullptr, lane);
                cf none[8];
                fft512_pad_bc_g<true, false>(zs, L, R.tb, nullptr, none);
                float y[4][4];
                r3_window(zs, R.w, y);
                if (kG == 1) {
                    float o[4];
                    r3_ola_shift(r0, r1, r2, y, o);
                    const long long be = fz - 3;
                    if (dense && be >= b0 && be < b_end && (be + 1) * NAE_HOP <= p.mid_len) {
                        lds_st(Yi + 512 + lane, c == 0 ? cf{o[2], o[3]} : cf{o[0], o[1]});   // the half the partner stores
                        keep = c == 0 ? cf{o[0], o[1]} : cf{o[2], o[3]};
                        pend_be = (int)be;
                    } else {
                        r3_store_block(p, it, bo, be, o, lane);   // wave-uniform: the block's base pointer stays scalar
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

// one launch of one instantiation.  More than 64 KiB of dynamic LDS needs the attribute: once per instantiation and DEVICE, so the flag lives in the
// context (no process-global launch state: contexts of different devices, or driven by different threads, do not share it)
template <typename K>
static int pv_launch(nae_ctx* ctx, const char* name, K kernel_unit, K kernel_strided, unsigned attr_bit, size_t lds, unsigned groups, const SigViewD& src,
                     const PvParams& p, long long n_sc, const uint32_t* phase_ws, const OutViewD& out, bool unit_stride)
{
    if (!(ctx->pv_attr_done & attr_bit)) {
        (void)nae_use_device(ctx);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_unit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_strided), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return nae_check(ctx, e, "hipFuncSetAttribute(vocoder pipeline)");
        ctx->pv_attr_done |= attr_bit;
    }
    const Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    NAE_KLAUNCH(ctx, name, unit_stride ? kernel_unit : kernel_strided, dim3(groups), dim3(kPipeThreads), lds, ctx->stream, src, p, n_sc, phase_ws, out, tb);
    return nae_check(ctx, hipGetLastError(), name);
}

template <int kG, bool kRich>
static int pipe_launch(nae_ctx* ctx, unsigned groups, const SigViewD& src, const PvParams& p, long long n_sc, const uint32_t* ws, const OutViewD& out, bool unit)
{
    return pv_launch(ctx, "pv_pipe_kernel", &pv_pipe_kernel<true, kG, kRich>, &pv_pipe_kernel<false, kG, kRich>,
                     1u << ((kG == 1 ? 0 : kG == 2 ? 1 : 2) * 2 + (kRich ? 1 : 0)), pipe_lds(kG), groups, src, p, n_sc, ws, out, unit);
}
template <int kG>
static int flow_launch(nae_ctx* ctx, unsigned groups, const SigViewD& src, const PvParams& p, long long n_sc, const uint32_t* ws, const OutViewD& out, bool unit)
{
    return pv_launch(ctx, "pv_flow_kernel", &pv_flow_kernel<true, kG>, &pv_flow_kernel<false, kG>, 1u << (8 + (kG == 1 ? 0 : kG == 2 ? 1 : 2)), flow_lds(kG),
                     groups, src, p, n_sc, ws, out, unit);
}

// frames_per_step: 1 = one stream-channel per slot; 2 / 4 = frame-interleaved (two / one stream-channel per four slots)
int nae_launch_pv_pipe(nae_ctx* ctx, const PvParams& p, const SigViewD& src, long long n_sc, const uint32_t* phase_ws,
                       const OutViewD& out, bool unit_stride, int frames_per_step)
{
    const long long items = n_sc * p.n_tiles;
    if (items == 0) return NAE_OK;
    if (frames_per_step != 1 && frames_per_step != 2 && frames_per_step != 4) return nae_fail(ctx, NAE_ERR_INVALID, "pv_pipe_kernel: frames per step");
    const int units = kPipeSlots / frames_per_step;
    // stereo units come in channel pairs of one (stream, tile): n_sc is even, so items is
    const long long groups = (items + units - 1) / units;
    if (groups > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "pv_pipe_kernel: grid too large");
    const unsigned g = (unsigned)groups;
    const bool one_per_cu = groups <= (long long)ctx->n_cu && !ctx->pv_lean;
    // at most one workgroup per CU, one frame per step (e.g. the 512 streams a rank of a 2-GPU job owns): the one-barrier schedule is 5 % faster; in the
    // frame-interleaved shapes it saves cycles and loses them to a lower clock (pv_flow = 2 forces it there)
    if (one_per_cu && (ctx->pv_flow >= 2 || (ctx->pv_flow == 1 && frames_per_step == 1))) {
        if (frames_per_step == 1) return flow_launch<1>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
        if (frames_per_step == 2) return flow_launch<2>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
        return flow_launch<4>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
    }
    // kRich: at most one workgroup per CU anyway (the frame-interleaved modes by their LDS; four slots per workgroup on a grid of at most n_cu workgroups)
    if (frames_per_step == 1) return one_per_cu ? pipe_launch<1, true>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride)
                                                : pipe_launch<1, false>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
    if (frames_per_step == 2) return pipe_launch<2, true>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
    return pipe_launch<4, true>(ctx, g, src, p, n_sc, phase_ws, out, unit_stride);
}
