// stft_device.h — wave-level STFT building blocks for gfx950 (CDNA4, wave64).
//
// One wavefront owns one 1024-sample frame: the 512 packed complex points live 8 per lane (16 VGPRs);
// the 512-point FFT is three register-resident radix-8 passes with two transposes through a wave-private
// LDS scratch (no workgroup barrier anywhere: DS operations of one wave execute in order).
//
// The operation order of everything up to the Q0.32 phase is the canonical one of DESIGN.md §3 so that the
// integer phases equal the CPU oracle's bit for bit.  This translation unit is compiled with
// -ffp-contract=off; fused multiply-adds appear only where __builtin_fmaf is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nae_dsp_spec.h"

namespace nae {

struct cf { float x, y; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; }; // 8-byte access at 4-byte alignment

constexpr int kScratchCf = 520;                 // per-wave LDS scratch, complex elements (512 for the transposes, 513 natural)

__device__ __forceinline__ void wave_lds_sync()
{
    // order this wave's LDS writes before its following LDS reads (other lanes' data); no instruction
    // beyond the waitcnt the compiler already tracks — DS ops of one wave are executed in issue order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------
// Complex arithmetic on (re, im) register pairs.  The product is built with NAE_PK = 0: plain f32 instructions.
// NAE_PK = 1 (a measured alternative, kept because it is bit-identical and small): hand-placed VOP3P instructions — one v_pk_add_f32 per
// complex add / subtract (a multiplication by -i is an operand swap + one sign: op_sel / neg_lo / neg_hi, no instruction), one v_pk_mul_f32
// + one v_pk_fma_f32 per twiddle product; no v_mov re-pairing (the compiler's own SLP packing needs those: Makefile).  Every component is
// the SAME rounded operation as in the plain form (x - y == x + (-y), (-a) b == -(a b), fma(-a, b, c): sign changes of operands are
// exact), and the two builds give the same bits on whole graphs (tools/lib_hash.py).  Round 6 measured it (profiles/r06_pk.md): the
// static vector-instruction count of the vocoder falls 1428 -> 1206, a packed instruction holds the arithmetic path for 2.02x the cycles of
// a plain one, and the kernels get SLOWER — vocoder 6.08 -> 6.21 ms, spectrum 2.75 -> 2.93 ms in the C5 step — at no better energy per
// frame (the +10 % flops per joule of a v_pk_fma_f32 stream over a v_fma_f32 stream does not carry over to add-dominated butterflies).
// NAE_PK = 2: only the twiddle products packed.
#ifndef NAE_PK
#define NAE_PK 0
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_v(cf a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ cf pk_c(v2f a) { return cf{a.x, a.y}; }
#define NAE_PK2(name, text)                                                                            \
    __device__ __forceinline__ cf name(cf a, cf b)                                                     \
    {                                                                                                  \
        v2f d;                                                                                         \
        asm(text : "=v"(d) : "v"(pk_v(a)), "v"(pk_v(b)));                                              \
        return pk_c(d);                                                                                \
    }
NAE_PK2(pk_add, "v_pk_add_f32 %0, %1, %2")                                                      // a + b
NAE_PK2(pk_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                            // a - b
NAE_PK2(pk_add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")         // a + (-i) b = (a.x + b.y, a.y - b.x)
NAE_PK2(pk_sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")         // a - (-i) b = (a.x - b.y, a.y + b.x)
NAE_PK2(pk_add_cj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                      // (a.x + b.x, a.y - b.y)
NAE_PK2(pk_sub_cj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                      // (a.x - b.x, a.y + b.y)
NAE_PK2(pk_swap_sub_add, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]")   // (a.y - b.x, a.x + b.y)
NAE_PK2(pk_mul, "v_pk_mul_f32 %0, %1, %2")                                                      // (a.x b.x, a.y b.y)
NAE_PK2(pk_mul_nh, "v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]")                                      // (a.x b.x, -(a.y b.y))
NAE_PK2(pk_mul_xx, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                                   // (a.x b.x, a.x b.y)
#undef NAE_PK2
// (-a.y b.y + c.x, a.y b.x + c.y)
__device__ __forceinline__ cf pk_fma_tw(cf a, cf b, cf c)
{
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(pk_v(a)), "v"(pk_v(b)), "v"(pk_v(c)));
    return pk_c(d);
}

#if NAE_PK
// v * w, one rounded product + one FMA per component
__device__ __forceinline__ cf cmul_tw(cf v, cf w) { return pk_fma_tw(v, w, pk_mul_xx(v, w)); }
#else
__device__ __forceinline__ cf cmul_tw(cf v, cf w)
{
    cf r;
    r.x = __builtin_fmaf(-v.y, w.y, v.x * w.x);
    r.y = __builtin_fmaf(v.y, w.x, v.x * w.y);
    return r;
}
#endif
#if NAE_PK == 1
__device__ __forceinline__ cf cadd(cf a, cf b) { return pk_add(a, b); }
__device__ __forceinline__ cf csub(cf a, cf b) { return pk_sub(a, b); }

// canonical forward 8-point DFT, in place, natural-order output: 28 packed instructions
__device__ __forceinline__ void dft8_fwd(cf (&a)[8])
{
    const cf c{NAE_SQRT1_2, NAE_SQRT1_2};
    const cf s0 = pk_add(a[0], a[4]), d0 = pk_sub(a[0], a[4]);
    const cf s1 = pk_add(a[1], a[5]), e1 = pk_sub(a[1], a[5]);
    const cf s2 = pk_add(a[2], a[6]), e2 = pk_sub(a[2], a[6]);
    const cf s3 = pk_add(a[3], a[7]), e3 = pk_sub(a[3], a[7]);
    const cf d1 = pk_mul(pk_add_mi(e1, e1), c);                   // ((e1.x + e1.y) c, (e1.y - e1.x) c)
    const cf d3 = pk_mul_nh(pk_swap_sub_add(e3, e3), c);          // ((e3.y - e3.x) c, -((e3.x + e3.y) c))
    const cf t0 = pk_add(s0, s2), t1 = pk_sub(s0, s2), t2 = pk_add(s1, s3), w3 = pk_sub(s1, s3);
    const cf u0 = pk_add_mi(d0, e2), u1 = pk_sub_mi(d0, e2), u2 = pk_add(d1, d3), x3 = pk_sub(d1, d3);
    a[0] = pk_add(t0, t2); a[4] = pk_sub(t0, t2); a[2] = pk_add_mi(t1, w3); a[6] = pk_sub_mi(t1, w3);
    a[1] = pk_add(u0, u2); a[5] = pk_sub(u0, u2); a[3] = pk_add_mi(u1, x3); a[7] = pk_sub_mi(u1, x3);
}
#else
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }

// canonical forward 8-point DFT, in place, natural-order output
__device__ __forceinline__ void dft8_fwd(cf (&a)[8])
{
    const float c = NAE_SQRT1_2;
    const cf s0 = cadd(a[0], a[4]), d0 = csub(a[0], a[4]);
    const cf s1 = cadd(a[1], a[5]), e1 = csub(a[1], a[5]);
    const cf s2 = cadd(a[2], a[6]), e2 = csub(a[2], a[6]);
    const cf s3 = cadd(a[3], a[7]), e3 = csub(a[3], a[7]);
    const cf d1 = cf{(e1.x + e1.y) * c, (e1.y - e1.x) * c};
    const cf d2 = mul_mi(e2);
    const cf d3 = cf{(e3.y - e3.x) * c, -((e3.x + e3.y) * c)};
    const cf t0 = cadd(s0, s2), t1 = csub(s0, s2), t2 = cadd(s1, s3), t3 = mul_mi(csub(s1, s3));
    const cf u0 = cadd(d0, d2), u1 = csub(d0, d2), u2 = cadd(d1, d3), u3 = mul_mi(csub(d1, d3));
    a[0] = cadd(t0, t2); a[4] = csub(t0, t2); a[2] = cadd(t1, t3); a[6] = csub(t1, t3);
    a[1] = cadd(u0, u2); a[5] = csub(u0, u2); a[3] = cadd(u1, u3); a[7] = csub(u1, u3);
}
#endif

// per-lane twiddles of the two twiddled passes, loop-invariant across frames
struct FftTw {
    cf a[7];        // W512^(lane*q), q = 1..7: 14 VGPRs, loop-invariant
    const cf* b;    // LDS table [m][p] = W512^(8 m p), 64 entries shared by the workgroup; this lane reads row lane&7
};

__device__ __forceinline__ void load_fft_tw(FftTw& tw, const cf* __restrict__ w512, const cf* w64_lds, int lane)
{
#pragma unroll
    for (int q = 1; q < 8; q++) tw.a[q - 1] = w512[lane * q];
    tw.b = w64_lds + 8 * (lane & 7);
}

// bin / packed-sample index held by (lane, register r) after the forward FFT:  k = lane + 64 r  (natural order)
__device__ __forceinline__ int kl_of_lane(int lane) { return lane; }

// canonical forward 512-point FFT.  in: v[j] = z[lane + 64 j];  out: v[r] = Z[lane + 64 r].
//
// The two transposes go through the wave's 512-entry LDS scratch with XOR-swizzled addresses chosen so that every
// ds_write_b64 (16-lane groups, 32 banks) and ds_read_b64 (32-lane groups, 64 banks) is conflict-free WITHOUT
// padding, and so that pass C leaves the result in natural order (lane = k mod 64):
//   T1  element u1[q][l]     at  (l ^ (q << 3)) + 64 q
//   T2  element u2[q][p][m]  at  (q ^ ((m & 3) << 1)) | ((p ^ (m >> 2)) << 3) | (m << 6)
// (address bits are an invertible GF(2) map of the index bits whose low 4 / 5 bits are a bijection of the
//  lane bits that vary inside one write / read group).
// kLaunder: the 32 swizzled LDS addresses below are loop-invariant, and hoisting them out of the frame loop pins 32
// VGPRs.  2 = recompute all of them per call (~75 instructions; the 128-VGPR vocoder builds), 1 = recompute only the 14
// that are one XOR each and keep the 16 two-to-three-instruction ones hoisted (the spectrum kernel: +16 VGPRs, -36
// instructions per FFT), 0 = keep everything hoisted (the 256-VGPR build).
template <int kLaunder = 2>
__device__ __forceinline__ void fft512_fwd(cf (&v)[8], cf* __restrict__ scratch, const FftTw& tw, int lane)
{
    int lane_x = lane;                                   // feeds the XOR-with-constant addresses
    if (kLaunder >= 1) asm volatile("" : "+v"(lane_x));
    if (kLaunder >= 2) asm volatile("" : "+v"(lane));
    const int m = lane & 7, qq = lane >> 3;
    // pass A
    dft8_fwd(v);
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], tw.a[q - 1]);
    // transpose 1: u1[q][l] -> lane (m, qq) register j = u1[qq][m + 8 j]
#pragma unroll
    for (int q = 0; q < 8; q++) scratch[(lane_x ^ (q << 3)) + 64 * q] = v[q];
    wave_lds_sync();
    {
        const int base = m + 64 * qq;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = scratch[base + ((j ^ qq) << 3)];
    }
    wave_lds_sync();
    // pass B
    dft8_fwd(v);
#pragma unroll
    for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], tw.b[p]);
    // transpose 2: u2[qq][p][m] -> lane (q'' = lane & 7, p'' = lane >> 3) register j = u2[q''][p''][j]
    {
        const int base = (qq ^ ((m & 3) << 1)) | (m << 6);
        const int ph = m >> 2;
#pragma unroll
        for (int p = 0; p < 8; p++) scratch[base | ((p ^ ph) << 3)] = v[p];
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = scratch[(lane_x ^ (((j & 3) << 1) | ((j >> 2) << 3))) + 64 * j];
    wave_lds_sync();
    // pass C
    dft8_fwd(v);
}

// ---------------------------------------------------------------------------------------------------------------
// Low-register form of the same FFT (identical arithmetic, so identical bits): every LDS access is a per-lane base
// register plus an IMMEDIATE offset — no swizzle arithmetic, no hoisted address registers — and the pass-A twiddles
// come from an LDS table instead of 14 VGPRs.  Conflict-freedom comes from padding instead of XOR swizzles:
//   T1  element u1[q][l]     at  72 q + l          (write: lane l, imm 72 q;  read: lane (m, qq) at 72 qq + m, imm 8 j)
//   T2  element u2[q][p][m]  at  q + 8 p + 66 m    (write: lane (m, qq) at qq + 66 m, imm 8 p;  read: lane, imm 66 j)
// ds_write_b64 is serviced in 16-lane groups over 32 banks, ds_read_b64 in 32-lane groups over 64 banks
// (MI355X_MICROARCH.md §LDS): 72 = 8 mod 32 and 66 = 2 mod 16 make each group hit distinct banks.
// Scratch: 576 complex (4608 B) per wave; natural order [lane + 64 r] (+ one entry at 512) uses the same area.
// What it buys: the FFT alone fits 45 VGPRs (the round-1 form with hoisted swizzle addresses and twiddles in registers
// needed ~100), so a kernel built on it can keep 6 waves per SIMD resident — one wave alone issues one vector instruction per 4.5-5
// cycles, two or more together one per 2.15 (profiles/r05_valu_wallclock.md), and the further waves hide LDS round trips.  Measured alone (profiles/r02_fftpad.md): 141-155 cycles of a CU per
// FFT at 6-8 waves per SIMD, LDS-bound (16 ds_write_b64 at 3.8 cycles + 30 ds_read_b64 at 1.3 cycles of the CU's LDS pipe).
constexpr int kPadScratchCf = 576;
constexpr int kTwaCf = 7 * 64;                   // LDS table W512^(lane q), q = 1..7, laid out [q-1][lane]

// 8-byte LDS accesses that hipcc must not pair up: its load/store optimizer turns two ds_read_b64 off one base register
// into one ds_read2_b64 / ds_read2st64_b64, which on gfx950 moves the same 16 bytes per lane in twice the LDS time
// (profiles/r02_valu_issue.md: 20.3 vs 2 x 5.15 cycles per SIMD at 8 waves).  A volatile access is left alone.
typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;
__device__ __forceinline__ cf lds_ld(const cf* p)
{
    const unsigned long long u = *(const lds_vu64*)(p);
    return cf{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))};
}
__device__ __forceinline__ void lds_st(cf* p, cf v)
{
    *(lds_vu64*)(p) = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
}

struct FftLds {
    cf* nat;         // scratch + lane                        natural [64 r], T1 write [72 q], T2 read [66 j]
    cf* t1r;         // scratch + 72 (lane >> 3) + (lane & 7) T1 read  [8 j]
    cf* t2w;         // scratch + (lane >> 3) + 66 (lane & 7) T2 write [8 p]
    cf* mir;         // scratch + 64 - lane                   mirror   [448 - 64 r]  == natural[512 - lane - 64 r]
    const cf* twa;   // twa table + lane                      [64 (q - 1)]
    const cf* twb;   // w64 table + 8 (lane & 7)              [p]
};

__device__ __forceinline__ FftLds make_fft_lds(cf* scratch, const cf* twa, const cf* w64, int lane)
{
    FftLds L;
    L.nat = scratch + lane;
    L.t1r = scratch + 72 * (lane >> 3) + (lane & 7);
    L.t2w = scratch + (lane >> 3) + 66 * (lane & 7);
    L.mir = scratch + 64 - lane;
    L.twa = twa + lane;
    L.twb = w64 + 8 * (lane & 7);
    return L;
}

// fills the pass-A twiddle table (one workgroup-wide call before the first barrier)
__device__ __forceinline__ void fill_twa(cf* twa, const cf* __restrict__ w512, int tid, int n_threads)
{
    for (int i = tid; i < kTwaCf; i += n_threads) twa[i] = w512[(i & 63) * ((i >> 6) + 1)];
}

// pass A (registers + the read-only twiddle table): may run while another wave still reads this wave's scratch
__device__ __forceinline__ void fft512_pad_a(cf (&v)[8], const FftLds& L)
{
    // the twiddle reads are issued in front of the butterflies that hide their latency (the accesses are volatile, so the
    // compiler keeps this order)
    cf w[7];
#pragma unroll
    for (int q = 0; q < 7; q++) w[q] = lds_ld(L.twa + 64 * q);
    dft8_fwd(v);
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], w[q - 1]);
}

// transposes, passes B and C
__device__ __forceinline__ void fft512_pad_bc(cf (&v)[8], const FftLds& L)
{
#pragma unroll
    for (int q = 0; q < 8; q++) lds_st(L.nat + 72 * q, v[q]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds_ld(L.t1r + 8 * j);
    wave_lds_sync();
    {
        cf w[7];
#pragma unroll
        for (int p = 1; p < 8; p++) w[p - 1] = lds_ld(L.twb + p);
        dft8_fwd(v);
#pragma unroll
        for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], w[p - 1]);
    }
#pragma unroll
    for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
    wave_lds_sync();
    dft8_fwd(v);
}

// pass A with the twiddles already requested by the caller (w[q-1] = twa[64 (q-1)], q = 1..7): lets a caller put these
// reads into one LDS round trip with its own
__device__ __forceinline__ void fft512_pad_a_tw(cf (&v)[8], const cf (&w)[7])
{
    dft8_fwd(v);
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], w[q - 1]);
}

// transposes, passes B and C with everything optional a caller may hold in registers: kTwReg: the pass-B twiddles wb[p-1]
// (else read from the LDS table), kExtra: 8 more values are requested behind the reads of the second transpose (extra[r] =
// xp[64 r]) and arrive behind pass C
template <bool kTwReg, bool kExtra>
__device__ __forceinline__ void fft512_pad_bc_g(cf (&v)[8], const FftLds& L, const cf (&wb)[7], const cf* xp, cf (&extra)[8])
{
#pragma unroll
    for (int q = 0; q < 8; q++) lds_st(L.nat + 72 * q, v[q]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds_ld(L.t1r + 8 * j);
    wave_lds_sync();
    if (kTwReg) {
        dft8_fwd(v);
#pragma unroll
        for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], wb[p - 1]);
    } else {
        cf w[7];
#pragma unroll
        for (int p = 1; p < 8; p++) w[p - 1] = lds_ld(L.twb + p);
        dft8_fwd(v);
#pragma unroll
        for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], w[p - 1]);
    }
#pragma unroll
    for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
    if (kExtra) {
#pragma unroll
        for (int r = 0; r < 8; r++) extra[r] = lds_ld(xp + 64 * r);
    }
    wave_lds_sync();
    dft8_fwd(v);
}

// (A register-only form of transpose 1 — v_permlane32_swap / v_permlane16_swap / DPP row_ror:8, 40 instructions for 8 writes + 8
// reads — lives in tools/ubench/xlane_t1.h: measured without gain in the FFT loop (profiles/r02_fftpad.md) and slower in the vocoder
// pipeline, so the product does not carry it.)
__device__ __forceinline__ void fft512_pad(cf (&v)[8], const FftLds& L)
{
    fft512_pad_a(v, L);
    fft512_pad_bc(v, L);
}

// canonical r2c split.  in: v[r] = Z[kl + 64 r].  out: v[r] = X[kl + 64 r]; returns X[512] (meaningful on lane 0).
// Leaves Z in natural order in scratch[0..511].
// kTwice: deliver 2 X instead of X (the two 1/2 factors are skipped; a factor 2 is exact in every operation below, so
// 2 X is bit-for-bit twice the canonical X).  For callers that only need phases, or that halve the result later.
template <bool kTwice = false>
__device__ __forceinline__ cf rfft_split(cf (&v)[8], cf* __restrict__ scratch, const cf* __restrict__ t1024,
                                         int lane)
{
    constexpr float h = kTwice ? 1.0f : 0.5f;
    const int kl = kl_of_lane(lane);
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[kl + 64 * r] = v[r];
    wave_lds_sync();
    cf nyq;
    {
        // k = 512: A = B = Z[0]
        const cf A = scratch[0];
        const cf E = kTwice ? cf{A.x + A.x, A.y - A.y} : cf{h * (A.x + A.x), h * (A.y - A.y)};
        const cf O = kTwice ? cf{A.x - A.x, A.y + A.y} : cf{h * (A.x - A.x), h * (A.y + A.y)};
        const cf P = cmul_tw(O, t1024[512]);
        nyq = cf{E.x + P.y, E.y - P.x};
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int k = kl + 64 * r;
        const cf A = v[r];
        const cf B = scratch[(512 - k) & 511];
        const cf E = kTwice ? cf{A.x + B.x, A.y - B.y} : cf{h * (A.x + B.x), h * (A.y - B.y)};
        const cf O = kTwice ? cf{A.x - B.x, A.y + B.y} : cf{h * (A.x - B.x), h * (A.y + B.y)};
        const cf P = cmul_tw(O, t1024[k]);
        v[r] = cf{E.x + P.y, E.y - P.x};
    }
    wave_lds_sync();
    return nyq;
}

// correctly rounded square root for x = 0, inf, NaN and every x >= 2^-96: v_sqrt_f32 (<= 1 ulp) plus the two-sided
// residual test of the compiler's own expansion, without its rescaling of tiny arguments (5 of its 15 instructions).
__device__ __forceinline__ float sqrt_rn(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}

// canonical atan2 in turns -> Q0.32 (include/nae_dsp_spec.h, revision 2: no division — an integer seed and three Newton
// steps give the reciprocal — and the octant fix-ups are integer reflections driven by the sign bits)
__device__ __forceinline__ uint32_t atan2_q32(float im, float re)
{
    const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), NAE_ATAN_TINY);
    // (min of three with +inf: one v_min3_f32 — a two-operand fminf is preceded by two canonicalising v_max)
    const float mn = __builtin_fminf(__builtin_fminf(ax, ay), __builtin_inff());
    float r = __uint_as_float(NAE_RCP_MAGIC - __float_as_uint(mx));
#ifndef NAE_ATAN_PROBE
#define NAE_ATAN_PROBE 0      // 1: a TIMING probe of a revision 3 (two Newton steps, degree-5 polynomial: 3 instructions less; not the specification's values)
#endif
#pragma unroll
    for (int it = 0; it < (NAE_ATAN_PROBE ? 2 : 3); it++) {
        const float e = __builtin_fmaf(-mx, r, 1.0f);
        r = __builtin_fmaf(r, e, r);
    }
    const float t = mn * r;
    const float s = t * t;
    float q = NAE_ATAN_C6 * NAE_ATAN_SCALE;
    if (!NAE_ATAN_PROBE) q = __builtin_fmaf(q, s, NAE_ATAN_C5 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C4 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C3 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C2 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C1 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C0 * NAE_ATAN_SCALE);
    uint32_t i = (uint32_t)__float2int_rn(q * t);
    if (ay > ax) i = 0x40000000u - i;
    const uint32_t m_re = (uint32_t)((int32_t)__float_as_uint(re) >> 31), m_im = (uint32_t)((int32_t)__float_as_uint(im) >> 31);
    i = (i ^ m_re) + (m_re & 0x80000001u);
    i = (i ^ m_im) - m_im;
    return ((ax + ay) < NAE_ATAN_HUGE) ? i : 0u;           // a bin of 2^100 or more — NaN, Inf, an overflowing sum included — has phase 0
}

// strided signal access: element (i) of one (stream, channel) at p[i * fs]
struct ChanView {
    const float* p;
    long long fs;   // frame stride in elements
    long long len;  // valid sample-frames: indices outside [0, len) read as zero
};

// load one frame in FFT input layout, un-windowed: v[j] = (x[s+2n], x[s+2n+1]), n = lane + 64 j.
// `s` and `in` are wave-uniform, so the interior/boundary choice is a scalar branch and the interior path
// addresses with a scalar base + 32-bit lane offset.  kUnit: frame stride 1 (planar source) -> 8-byte loads.
template <bool kUnit>
__device__ __forceinline__ void load_frame_raw(cf (&v)[8], const ChanView& in, long long s, int lane)
{
    const bool interior = (s >= 0) && (s + NAE_FFT_N <= in.len);
    if (interior) {
        if (kUnit) {
            const float* base = in.p + s;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const f2u t = *reinterpret_cast<const f2u*>(base + 2 * lane + 128 * j); // 4-byte aligned 8-byte access
                v[j] = cf{t.x, t.y};
            }
        } else {
            // strided source (interleaved stereo: fs = 2): ONE vector multiply for the lane offset; the per-j part
            // of the address is wave-uniform and stays in scalar registers (v_mul_lo_u32 is quarter rate)
            const int fs = (int)in.fs;
            const int lane_off = 2 * lane * fs;
            const float* base = in.p + s * in.fs;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float* bj = base + (long long)(128 * j) * fs;
                v[j] = cf{bj[lane_off], bj[lane_off + fs]};
            }
        }
    } else {
        // (rare path: keep its 64-bit per-lane indices from being hoisted out of the caller's frame loop)
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const long long i0 = s + 2 * (lane + 64 * j);
            v[j].x = (i0 >= 0 && i0 < in.len) ? in.p[i0 * in.fs] : 0.0f;
            v[j].y = (i0 + 1 >= 0 && i0 + 1 < in.len) ? in.p[(i0 + 1) * in.fs] : 0.0f;
        }
    }
}

__device__ __forceinline__ void apply_window(cf (&v)[8], const float* __restrict__ hann_lds, int lane)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float2 w = *reinterpret_cast<const float2*>(hann_lds + 2 * (lane + 64 * j));
        v[j] = cf{v[j].x * w.x, v[j].y * w.y};
    }
}

template <bool kUnit>
__device__ __forceinline__ void load_frame_windowed(cf (&v)[8], const ChanView& in, long long s,
                                                    const float* __restrict__ hann_lds, int lane)
{
    load_frame_raw<kUnit>(v, in, s, lane);
    apply_window(v, hann_lds, lane);
}

} // namespace nae
