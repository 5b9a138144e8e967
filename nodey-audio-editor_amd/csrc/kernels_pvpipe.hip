// kernels_pvpipe.hip — pass 3 of the phase vocoder (K7) as a four-role wave pipeline for gfx950.
//
// Why a pipeline.  On gfx950 one wave issues at most one vector instruction per 4.5-5 cycles, while a SIMD with two or more
// resident waves retires one per 2.15 cycles (profiles/r05_valu_wallclock.md; rounds 2-4 believed 1.0-1.3 at 8 waves) — and more
// waves hide the LDS round trips and barrier waits.  A stream-channel of the vocoder is a serial chain of
// frames (integer phase accumulator, overlap-add), so "one wave per stream-channel" leaves most issue slots empty.  Here a
// frame passes through FOUR waves, one per role, that hand it on through LDS once per step, and every role fits 64 VGPRs:
// a 1024-thread workgroup is 4 slots x 4 roles, two workgroups fill a CU (32 waves, 8 per SIMD — each SIMD hosts one wave
// of every role of two slots):
//
//   step t:   R1   frame t    load, Hann window, forward FFT                                         -> Z   (its own FFT scratch)
//             R2a  frame t-1  bins lane+64r, r = 0..3: r2c split, atan2 -> Q0.32, phase advance, rotate -> Y   (hand-off buffer)
//             R2b  frame t-1  the same for r = 4..7 and bin 512
//             R3   frame t-2  c2r pre-twiddle, inverse FFT (by forward FFT), overlap-add, store the finished hop block (interleaved stereo output:
//                             one barrier later, as a dense 16-byte-per-lane piece assembled with the other channel's wave through LDS)
//
// (round 2 ran three roles at 80 VGPRs / 6 waves per SIMD: the phase role carried half of a frame's ~1100 vector
// instructions and, issuing at the single-wave rate, set the step; profiles/r02_pipe_stamps*.txt.)
// A step has two workgroup barriers: after A the consumers read what the producers left in step t-1 into registers;
// after B the producers overwrite.  So the hand-off buffers need no double buffering and one slot costs
// 2 x 4608 B (FFT scratch of R1 / R3) + 4160 B (Y) of LDS.
//
// Slots.  kG = 1: the four slots of a workgroup are four stream-channels (two stereo streams) — the regime of large
// batches (>= 1024 stream-channels: one tile per stream-channel, no pass 1, nothing analysed twice).
// kG = 2 / 4 ("frame-interleaved"): the slots of a unit are kG CONSECUTIVE FRAMES of ONE stream-channel and a step
// advances it by kG frames.  FFTs, split, atan2, rotation are independent per frame; what is sequential is
//   * the Q0.32 phase: frame f needs Qa of f-1 and Qs of f-1.  R2 analyses frame f in step t and leaves Qa in LDS; in
//     step t+1 it fetches its predecessor's Qa, forms its increment (exact integers), leaves it in LDS, and behind the
//     next barrier every R2 wave sums the increments of the slots up to its own onto the running phase (one add per bin);
//   * the overlap-add: a hop block sums quarters of four consecutive frames = four different slots.  R3 leaves quarters
//     1..3 of its windowed frame in LDS and, one step later, the wave that holds the block's LAST frame adds them in
//     frame order — the order of the kG = 1 registers and of the oracle, so all modes give the same samples.
// This is what small batches run (the 128 streams one rank of an 8-GPU job owns are 256 stream-channels = one
// workgroup per CU): no time tiles, no pass 1, no second analysis.  Fewer stream-channels still get tiles (nae_pick_pv_shape).
//
// The arithmetic — and therefore every integer phase — is the canonical one of DESIGN.md §3 (same dft8_fwd / cmul_tw /
// atan2_q32 / phase increment as the other kernels); only where data waits between operations differs.
// Replaces: SoundTouch behind /root/reference/src/processor/audio-velocity.cpp:369-428 (algorithm differs: DESIGN.md §3).
#include "stft_common.h"
#include "pv_roles.h"

namespace nae {

constexpr int pipe_threads(int kS) { return 64 * 4 * kS; }   // kS slots: waves [0,kS) = R1, [kS,2kS) = R2a, [2kS,3kS) = R2b, [3kS,4kS) = R3
constexpr size_t kPipeLdsTables = NAE_FFT_N * sizeof(float) + (kT1024Pad + 64 + kTwaCf) * sizeof(cf);
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
constexpr size_t pipe_lds(int kG, int kS) { return kPipeLdsTables + kS * (kPipeLdsPerSlot + pipe_lds_x_per_slot(kG)); }
static_assert(2 * pipe_lds(1, 4) <= 160 * 1024, "two workgroups per CU");
static_assert(pipe_lds(2, 4) <= 160 * 1024 && pipe_lds(4, 4) <= 160 * 1024, "one workgroup per CU");

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

template <bool kUnit, int kG, int kS, bool kRich>
__global__ __launch_bounds__(pipe_threads(kS), kRich ? (kS == 4 ? 4 : 2) : 8) void pv_pipe_kernel(SigViewD src, PvParams p, long long n_sc,
                                                                               const uint32_t* __restrict__ base_phase, OutViewD out, Tables tb)
{
    constexpr int kPipeSlots = kS, kPipeThreads = pipe_threads(kS);
    static_assert(kS % kG == 0, "a unit's frames share a workgroup");
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
    if (kG == 1 && threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_ID: CU 8-11, SE 13-14
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;     // XCC_ID
        const unsigned key = (xcc * 4 + ((hw >> 13) & 3u)) * 16 + ((hw >> 8) & 15u);
        s_slot = (int)(atomicAdd(&g_cu_arrivals[key], 1u) & 1u);
    }
    __syncthreads();
    const int prio_slot = kG == 1 ? __builtin_amdgcn_readfirstlane(s_slot) : 0;

    const int wave = wave_id();
    const int role = wave / kPipeSlots, slot = wave % kPipeSlots;   // scalars
    const int unit = slot / kG, j = slot % kG;                      // j: frame of the step this slot works on (kG > 1)
    const int lane = threadIdx.x & 63;
    // stereo: units 2i, 2i+1 are the two channels of one (stream, tile), so with kG <= 2 their block stores of an interleaved
    // destination happen in the same step and merge in L2; mono: consecutive (stream, tile) items
    const long long ug = (long long)kUnits * blockIdx.x + unit;
    long long sc;
    int tile;
    if (p.ch == 2) {
        const long long pair = ug >> 1;                          // (stream, tile), tile fastest
        sc = 2 * (pair / p.n_tiles) + (ug & 1);
        tile = (int)(pair % p.n_tiles);
    } else {
        sc = ug / p.n_tiles;
        tile = (int)(ug % p.n_tiles);
    }
    if (sc >= n_sc) return;                                  // a terminated wave no longer counts at s_barrier
    const long long s_idx = sc / p.ch;
    const int c = (int)(sc % p.ch);

    cf* S1 = reinterpret_cast<cf*>(smem + kPipeLdsTables + slot * kPipeLdsPerSlot);
    cf* Y = S1 + kPadScratchCf;
    cf* S3 = Y + kYCf;
    // frame-interleaved exchange areas, indexed by slot: QA[2][520] | INC[520] | OL[kGens][3][256]
    unsigned char* xbase = smem + kPipeLdsTables + kPipeSlots * kPipeLdsPerSlot;
    constexpr size_t kXPer = pipe_lds_x_per_slot(kG);
    auto x_qa = [&](int sl, int gen) { return reinterpret_cast<uint32_t*>(xbase + sl * kXPer) + gen * kPhasePad; };
    auto x_inc = [&](int sl) { return reinterpret_cast<uint32_t*>(xbase + sl * kXPer) + 2 * kPhasePad; };
    auto x_ola = [&](int sl, int gen) { return reinterpret_cast<float*>(xbase + sl * kXPer + 3 * kPhasePad * sizeof(uint32_t)) + gen * 3 * kOlaQuarter; };

    const long long b0 = p.f_origin + (long long)tile * p.tile;      // first output block == first frame of the tile
    long long b_end = b0 + p.tile;
    if (b_end > p.f_stop) b_end = p.f_stop;
    long long f_end = b_end + 3;                                       // frames b0 .. b_end+2 feed blocks b0 .. b_end-1
    if (f_end > p.frames) f_end = p.frames;
    const long long f_first = (b0 > 0 ? b0 - 1 : 0);                   // frame b0-1 only primes the previous phase
    const int n = (int)(f_end - f_first);
    if (n <= 0) return;
    const int steps = (n + kG - 1) / kG;
    // interleaved stereo output, one frame per step: a finished block leaves one barrier later, through the exchange area (R3) — one more step for everybody
    const bool dense_shape = kG == 1 && p.ch == 2 && out.fs == 2 && out.cs == 1;
    const int T = steps + kDepth + (dense_shape ? 1 : 0);
    /*pipe:begin*/
    unsigned long long now = 0;                              // shader clock, read behind barrier B, used behind the next barrier A
    if (kG > 1 && role == 2) __builtin_amdgcn_s_setprio(1);   // frame-interleaved (one workgroup per CU): R2b one level up (1.10 against 1.15 ms at 128 streams)

    if (role == 0) {
        // ------------------------------------------------------------------------------------------ R1: analysis FFT
        ChanView in{src.base + s_idx * src.ss + c * src.cs, src.fs, p.in_len};
        cf nxt[8], va[8];
        if (j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + j), lane);
        // kRich (one workgroup per CU: 128 VGPRs per wave): window and both twiddle sets stay in registers — 30 LDS reads less
        // per frame and one round trip less on the step's critical path
        cf r_w[8], r_ta[7], r_tb[7];
        if (kRich) {
#pragma unroll
            for (int r = 0; r < 8; r++) r_w[r] = lds_ld(reinterpret_cast<const cf*>(hann) + lane + 64 * r);
#pragma unroll
            for (int q = 0; q < 7; q++) { r_ta[q] = lds_ld(twa + lane + 64 * q); r_tb[q] = lds_ld(w64 + 8 * (lane & 7) + q + 1); }
        }
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            const bool cur = kG * t + j < n;
            pipe_barrier();                                   /*A*/
            if (kG == 1) pipe_prio(prio_slot, role, now);
            if (cur) {
                // register-only part while R2 reads Z of the previous step out of this wave's scratch.  Window and pass-A
                // twiddles are requested together (one LDS round trip): the accesses are volatile, so the compiler keeps them
                // where they are written, and one read per product would cost one round trip each
                const int la = pipe_lane<!kRich>(lane);
                const cf* hw = reinterpret_cast<const cf*>(hann) + la;      // window of samples 2 (lane + 64 r), +1
                const cf* ta = twa + la;
                if (kRich) {
#pragma unroll
                    for (int r = 0; r < 8; r++) va[r] = cf{nxt[r].x * r_w[r].x, nxt[r].y * r_w[r].y};
                    fft512_pad_a_tw(va, r_ta);
                } else {
                    cf w[8], tw[7];
#pragma unroll
                    for (int r = 0; r < 8; r++) w[r] = lds_ld(hw + 64 * r);
#pragma unroll
                    for (int q = 0; q < 7; q++) tw[q] = lds_ld(ta + 64 * q);
#pragma unroll
                    for (int r = 0; r < 8; r++) va[r] = cf{nxt[r].x * w[r].x, nxt[r].y * w[r].y};
                    fft512_pad_a_tw(va, tw);
                }
            }
            pipe_barrier();                                   /*B*/
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            if (cur) {
                const FftLds L = make_fft_lds(S1, twa, w64, pipe_lane<!kRich>(lane));
                cf none[8];
                fft512_pad_bc_g<kRich, false>(va, L, r_tb, nullptr, none);
#pragma unroll
                for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, va[r]);
                if (lane == 0) S1[512] = va[0];               // so that the mirror of bin 0 is read like any other
                // request the next frame now: the loads land while the wave waits at the barriers
                if (kG * (t + 1) + j < n) load_frame_raw<kUnit>(nxt, in, frame_start(p, f_first + kG * (t + 1) + j), lane);
            }
        }
        /*pipe:r1-end*/
    } else if (role <= 2) {
        // ------------------------------------------------------------------------------------------ R2a / R2b: phases
        // Bins in mirror pairs: a lane owns k = lane + 64 (2 h + i) and 512 - k, i = 0, 1 (items 2 i and 2 i + 1) — 0..127 and
        // 385..512 for h = 0, 128..255 and 257..384 for h = 1.  One pair of reads (A = Z[k], B = Z[512 - k]) gives both spectra:
        // the mirror's E and O are (Ex, -Ey) and (-Ox, Oy) — exact negations and commuted sums of the canonical formula, so
        // every phase keeps its bits — and both rotated bins of a pair meet in one lane, where the c2r pre-twiddle of R3's FFT
        // input needs them: what goes to R3 is that input, not Y.
        // 513 bins are 512 items and one more.  The odd one is bin 512, whose phase is just the sign of a real number (no atan2):
        // in lane 0 of h = 0 the pair would be (0, 512); there item 1 carries the self-mirrored bin 256 instead (its own A = B =
        // Z[256]), and bin 512 rides along in that wave as a fifth, cheap item (sign, increment, one cosine).
        const int h = role - 1;
        const int k0 = lane + 128 * h;                        // items 0 / 2: k0, k0 + 64; items 1 / 3: 512 - k0 (256 in lane 0 of h = 0), 448 - k0
        const bool dc = (h == 0) && (lane == 0);              // the lane of bins 0, 256 and 512
        const int km0 = dc ? 256 : 512 - k0;                  // bin of item 1
        cf tk[2], tm[2];                                      // split twiddles of the items' bins (loop-invariant: 8 VGPRs)
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
        cf hx[5];                                             // kG > 1: the frame analysed in the previous step ([4]: bin 512)
        uint32_t hqa[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 5; q++) hx[q] = cf{0.0f, 0.0f};
        // phase increment of all items of this lane (exact integers)
        auto inc_items = [&](const uint32_t (&qa)[5], const uint32_t (&qv)[5], unsigned d, unsigned R, uint32_t (&inc)[5]) {
            const int kk = k0, km = km0;
            inc[0] = pipe_inc(qa[0], qv[0], (unsigned)kk, d, R);
            inc[1] = pipe_inc(qa[1], qv[1], (unsigned)km, d, R);
            inc[2] = pipe_inc(qa[2], qv[2], (unsigned)(kk + 64), d, R);
            inc[3] = pipe_inc(qa[3], qv[3], (unsigned)(448 - kk), d, R);
            inc[4] = (h == 0) ? pipe_inc(qa[4], qv[4], 512u, d, R) : 0u;
        };
        // a continued stream carries the synthesis phase behind its segment's last frame on (nae_stream.hip)
        auto carry_store = [&](const uint32_t (&q)[5]) {
            uint32_t* co = p.carry_out + sc * kT1024Pad;
            co[k0] = q[0]; co[km0] = q[1]; co[k0 + 64] = q[2]; co[448 - k0] = q[3];
            if (dc) co[512] = q[4];
        };
        // rotation by the phase difference and the in-lane c2r pre-twiddle: R3's FFT input Zin[k], Zin[512 - k]
        // (conjugated, inverse = conj(FFT(conj Z)) / 512; 2E, 2D: see kGain)
        auto synth_items = [&](const cf (&x)[5], const uint32_t (&qsv)[5], const uint32_t (&qav)[5]) {
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
        };
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
                const int kk = k0;
                const cf* Zk = S1 + kk;
                const cf* Zm = S1 + 512 - kk;
                cf A[2], B[2];
#pragma unroll
                for (int i = 0; i < 2; i++) { A[i] = lds_ld(Zk + 64 * i); B[i] = lds_ld(Zm - 64 * i); }
                cf z256{0.0f, 0.0f};
                if (h == 0) z256 = lds_ld(S1 + 256);
                // r2c split -> 2 X (phases are scale-invariant; the factor is undone in R3's output gain: a factor 2 is
                // exact in every product on the way)
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
            if (kG > 1) {
                // increment of the held frame fb: its predecessor's analysis phases were left in LDS one step ago (slot j-1), or
                // two steps ago by the last slot (j = 0)
                uint32_t inc[5] = {0, 0, 0, 0, 0};
                const int kk = k0, km = km0;
                if (act_b && fb >= b0) {
                    if (fb == 0) {
#pragma unroll
                        for (int q = 0; q < 5; q++) inc[q] = hqa[q];   // the "increment" of frame 0 is its analysis phase
                    } else {
                        const uint32_t* pq = j > 0 ? x_qa(slot - 1, (t - 1) & 1) : x_qa(slot + kG - 1, t & 1);
                        uint32_t pv[5];
                        pv[0] = pq[kk]; pv[1] = pq[km]; pv[2] = pq[kk + 64]; pv[3] = pq[448 - kk];
                        pv[4] = pq[512];
                        const unsigned d = (unsigned)(frame_start(p, fb) - frame_start(p, fb - 1));
                        const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
                        inc_items(hqa, pv, d, R, inc);
                    }
                }
                uint32_t* pi = x_inc(slot);
                pi[kk] = inc[0]; pi[km] = inc[1]; pi[kk + 64] = inc[2]; pi[448 - kk] = inc[3];
                if (dc) pi[512] = inc[4];
            }
            pipe_barrier();                                   /*B*/  // R1 may overwrite its scratch
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            uint32_t qa[5] = {0, 0, 0, 0, 0};
            if (act_a) {
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
                    if (fa >= b0) synth_items(va, qs, qa);
                }
            } else {
                if (act_a) {
                    const int kk = k0, km = km0;
                    uint32_t* pq = x_qa(slot, t & 1);
                    pq[kk] = qa[0]; pq[km] = qa[1]; pq[kk + 64] = qa[2]; pq[448 - kk] = qa[3];
                    if (dc) pq[512] = qa[4];
                }
                if (act_b) {
                    // running phase: the increments of the unit's slots up to this one; all of them move the base on
                    const int kk = k0, km = km0;
                    uint32_t mine[5], base[5];
#pragma unroll
                    for (int q = 0; q < 5; q++) { mine[q] = qs[q]; base[q] = qs[q]; }
#pragma unroll
                    for (int i2 = 0; i2 < kG; i2++) {
                        const uint32_t* pi = x_inc(slot - j + i2);
                        uint32_t v[5];
                        v[0] = pi[kk]; v[1] = pi[km]; v[2] = pi[kk + 64]; v[3] = pi[448 - kk];
                        v[4] = pi[512];
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            base[q] += v[q];
                            if (i2 <= j) mine[q] += v[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 5; q++) qs[q] = base[q];
                    if (fb == p.carry_frame) carry_store(mine);
                    if (fb >= b0) synth_items(hx, mine, hqa);
                }
                if (act_a) {
#pragma unroll
                    for (int q = 0; q < 5; q++) { hx[q] = va[q]; hqa[q] = qa[q]; }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------ R3: synthesis
        float* optr = out.base + s_idx * out.ss + c * out.cs;
        const bool out_vec = (out.fs == 1) && ((reinterpret_cast<uintptr_t>(optr) & 15) == 0);
        // Overlap-add.  Sample n = 2 (lane + 64 r) + {0,1} of a frame falls into hop block r >> 1 at offset
        // 2 lane + 128 (r & 1) + {0,1}: a lane touches the same 4 offsets of every block.  kG = 1: the 3 open blocks are 12
        // VGPRs (the 4th block a frame touches is new); block fz-3 is complete once frame fz is in; contributions arrive in
        // increasing frame order, as in the oracle.  kG > 1: quarters 1..3 go to LDS, quarter 0 waits in 4 VGPRs for the
        // next step, where the other three quarters of its block are fetched from the slots of the three frames before.
        // Sums are kept unscaled; the constants of the tolerance path — 1/512 (inverse FFT), 1/2 (c2r pre-twiddle), 1/2
        // (analysis split) and 2/3 (overlap-add gain) — scale the finished block.
        constexpr float kGain = NAE_OLA_GAIN / 2048.0f;
        float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float y0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        cf r_w[8], r_ta[7], r_tb[7];                          // kRich: synthesis window and twiddles in registers
        if (kRich) {
#pragma unroll
            for (int r = 0; r < 8; r++) r_w[r] = lds_ld(reinterpret_cast<const cf*>(hann) + lane + 64 * r);
#pragma unroll
            for (int q = 0; q < 7; q++) { r_ta[q] = lds_ld(twa + lane + 64 * q); r_tb[q] = lds_ld(w64 + 8 * (lane & 7) + q + 1); }
        }
        bool had = false;                                     // kG > 1: a frame of this slot went through the previous step
        // dense stores (kG = 1, interleaved stereo, 16-byte aligned stream): XB[channel][half][lane] = the lane's two sample pairs of the block
        const bool dense = dense_shape && ((reinterpret_cast<uintptr_t>(optr - c) & 15) == 0);   // (optr - c: channel 0 of the stream)
        int pend_be = -1;                                     // block waiting in the exchange area (wave-uniform; the same in both channel waves)
        auto xchg = [&]() { return reinterpret_cast<cf*>(xbase + (size_t)(slot & ~1) * kPipeXchgPerSlot); };
        auto store_block = [&](long long be, const float (&o)[4]) {
            if (be >= b0 && be < b_end && be * NAE_HOP < p.mid_len) {
                // buffer stores: scalar descriptor of the block + one 32-bit lane offset (plain pointer stores made hipcc
                // hoist four 64-bit per-lane addresses out of the frame loop)
                float* pb = optr + be * NAE_HOP * out.fs;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, -1, 0x00020000);
                auto st = [&](unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 0); };
                const int ls = pipe_lane<!kRich>(lane);
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
            const int iz = kG * (t - (kDepth - (kG == 1 ? 0 : 1))) + j;      // kG = 1: t - 2;  kG > 1: t - 3
            const long long fz = f_first + iz;
            const bool active = iz >= 0 && iz < n && fz >= b0;
            pipe_barrier();                                   /*A*/  // the FFT input of frame fz is complete
            if (kG == 1) pipe_prio(prio_slot, role, now);
            if (kG == 1 && pend_be >= 0) {
                // both channels' blocks of the previous step are in XB: this wave writes half c of the interleaved block — samples 128 c + 2 lane, + 1 of
                // both channels = 16 contiguous bytes per lane, 1 KiB per wave
                const int lx = pipe_lane<!kRich>(lane);
                const cf* XB = xchg();
                const cf P0 = lds_ld(XB + (0 + c) * 64 + lx), P1 = lds_ld(XB + (2 + c) * 64 + lx);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(optr - c + (long long)pend_be * (2 * NAE_HOP), 0, -1, 0x00020000);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(P0.x), __float_as_uint(P1.x), __float_as_uint(P0.y), __float_as_uint(P1.y)}, rs,
                                                       16 * lx, 1024 * c, 0);
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
                    const int rel = j - i;                                   // < 0: an earlier step
                    const int back = rel >= 0 ? 0 : (-rel + kG - 1) / kG;
                    const int sl = slot - j + rel + back * kG;
                    const int gen = (t - 1 - back + 2 * kGens) % kGens;
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
                    fft512_pad_a_tw(zs, r_ta);
                } else {
                    cf tw[7];
#pragma unroll
                    for (int q = 0; q < 7; q++) tw[q] = lds_ld(ta + 64 * q);
                    fft512_pad_a_tw(zs, tw);
                }
            }
            if (finish) {
                // the block that the frame of the previous step completed: quarters 3, 2, 1 of the three frames before it
                // (slot (j - i) mod kG, floor((j - i) / kG) steps earlier), then the own quarter 0 — the frame order
                float o[4] = {oq[2].x, oq[2].y, oq[2].z, oq[2].w};
                o[0] += oq[1].x; o[1] += oq[1].y; o[2] += oq[1].z; o[3] += oq[1].w;
                o[0] += oq[0].x; o[1] += oq[0].y; o[2] += oq[0].z; o[3] += oq[0].w;
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = (o[i] + y0[i]) * kGain;
                store_block(fz - kG - 3, o);
            }
            pipe_barrier();                                   /*B*/  // R2 may overwrite the FFT input
            if (kG == 1) now = __builtin_amdgcn_s_memtime();
            had = active;
            if (active) {
                const int lb = pipe_lane<!kRich>(lane);
                const FftLds L = make_fft_lds(S3, twa, w64, lb);
                const cf* hw = reinterpret_cast<const cf*>(hann) + lb;
                cf wn[8];                                     // synthesis window: requested behind the second transpose
                fft512_pad_bc_g<kRich, !kRich>(zs, L, r_tb, hw, wn);
                if (kRich) {
#pragma unroll
                    for (int r = 0; r < 8; r++) wn[r] = r_w[r];
                }
                // zs[r] = conj(z[n]) * 512 (x 4), n = lane + 64 r  ->  time samples 2n, 2n+1, windowed
                float y[4][4];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const cf w = wn[r];
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
                        const int lx = pipe_lane<!kRich>(lane);
                        cf* XB = xchg();
                        lds_st(XB + (2 * c + 0) * 64 + lx, cf{o[0], o[1]});
                        lds_st(XB + (2 * c + 1) * 64 + lx, cf{o[2], o[3]});
                        pend_be = (int)be;
                    } else {
                        store_block(be, o);                   // wave-uniform: the block's base pointer stays scalar
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

} // namespace nae

using namespace nae;

template <int kG, int kS, bool kRich>
static int pipe_launch(nae_ctx* ctx, unsigned groups, const SigViewD& src, const PvParams& p, long long n_sc, const uint32_t* phase_ws,
                        const OutViewD& out, const Tables& tb, bool unit_stride)
{
    // more than 64 KiB of dynamic LDS needs the attribute: once per instantiation and DEVICE, so the flag lives in the context
    // (no process-global launch state: contexts of different devices, or driven by different threads, do not share it)
    constexpr unsigned bit = 1u << ((kG == 1 ? 0 : kG == 2 ? 1 : 2) * 2 + (kRich ? 1 : 0));
    if (!(ctx->pv_attr_done & bit)) {
        (void)nae_use_device(ctx);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_pipe_kernel<true, kG, kS, kRich>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_lds(kG, kS));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_pipe_kernel<false, kG, kS, kRich>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_lds(kG, kS));
        if (e != hipSuccess) return nae_check(ctx, e, "hipFuncSetAttribute(pv_pipe_kernel)");
        ctx->pv_attr_done |= bit;
    }
    if (unit_stride)
        NAE_KLAUNCH(ctx, "pv_pipe_kernel", (pv_pipe_kernel<true, kG, kS, kRich>), dim3(groups), dim3(pipe_threads(kS)), pipe_lds(kG, kS), ctx->stream, src, p,
                    n_sc, phase_ws, out, tb);
    else
        NAE_KLAUNCH(ctx, "pv_pipe_kernel", (pv_pipe_kernel<false, kG, kS, kRich>), dim3(groups), dim3(pipe_threads(kS)), pipe_lds(kG, kS), ctx->stream, src, p,
                    n_sc, phase_ws, out, tb);
    return NAE_OK;
}

// frames_per_step: 1 = one stream-channel per slot; 2 / 4 = frame-interleaved (two / one stream-channel per four slots)
int nae_launch_pv_pipe(nae_ctx* ctx, const PvParams& p, const SigViewD& src, long long n_sc, const uint32_t* phase_ws,
                       const OutViewD& out, bool unit_stride, int frames_per_step)
{
    const long long items = n_sc * p.n_tiles;
    if (items == 0) return NAE_OK;
    if (frames_per_step != 1 && frames_per_step != 2 && frames_per_step != 4) return nae_fail(ctx, NAE_ERR_INVALID, "pv_pipe_kernel: frames per step");
    const int slots = 4;
    const int units = slots / frames_per_step;
    // stereo units come in channel pairs of one (stream, tile): n_sc is even, so items is
    const long long groups = (items + units - 1) / units;
    if (groups > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "pv_pipe_kernel: grid too large");
    // at most one workgroup per CU, one frame per step (e.g. the 512 streams a rank of a 2-GPU job owns): the one-barrier pipeline with doubled
    // hand-off buffers (kernels_pvflow.hip) is 5 % faster; in the frame-interleaved shapes it saves cycles and loses them to a lower clock
    if (groups <= (long long)ctx->n_cu && !ctx->pv_lean && (ctx->pv_flow >= 2 || (ctx->pv_flow == 1 && frames_per_step == 1)))
        return nae_launch_pv_flow(ctx, p, src, n_sc, phase_ws, out, unit_stride, frames_per_step);
    Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    // kRich: at most one workgroup per CU anyway (the frame-interleaved modes by their LDS; four slots per workgroup on a grid
    // of at most n_cu workgroups) -> 128 VGPRs per wave, tables in registers
    const bool rich1 = frames_per_step == 1 && groups <= (long long)ctx->n_cu && !ctx->pv_lean;
    int rc;
    if (frames_per_step == 1 && rich1) rc = pipe_launch<1, 4, true>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    else if (frames_per_step == 1) rc = pipe_launch<1, 4, false>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    else if (frames_per_step == 2) rc = pipe_launch<2, 4, true>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    else rc = pipe_launch<4, 4, true>(ctx, (unsigned)groups, src, p, n_sc, phase_ws, out, tb, unit_stride);
    if (rc) return rc;
    return nae_check(ctx, hipGetLastError(), "pv_pipe_kernel");
}
