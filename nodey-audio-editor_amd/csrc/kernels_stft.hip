// kernels_stft.hip — K8 FFT spectrum and K7 tempo/pitch (phase vocoder + rate transposer) for gfx950.
//
// Mapping (MI355X: 256 CUs x 4 SIMD, wave64, 160 KiB LDS/CU):
//   * one wavefront = one (stream, channel, tile of frames); a 512-thread workgroup is 8 independent waves
//     that only share the read-only Hann / split-twiddle tables in LDS (one __syncthreads after the table
//     fill, none afterwards).
//   * the phase accumulator is Q0.32 integer, so the time recurrence of the vocoder is an exact prefix sum:
//     pass 1 (pv_phase_kernel) reduces each tile's phase increments, pass 2 (pv_scan_kernel) scans tiles,
//     pass 3 (kernels_pvpipe.hip) recomputes the tile with the right starting phase and overlap-adds in registers.
//     HBM traffic stays at the algorithmic 4 B in + 4 B out per sample per channel (+ one re-read in pass 1).
//
// Replaces: SoundTouch behind /root/reference/src/processor/audio-velocity.cpp:369-428 (K7; algorithm differs,
// see DESIGN.md §3) and the FFTW-based spectrum the reference declares but never implements (K8).
#include "stft_common.h"
#include <stdlib.h>

namespace nae {

constexpr int kWaves = 8;                        // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr size_t kLdsTables = NAE_FFT_N * sizeof(float) + kT1024Pad * sizeof(cf) + 64 * sizeof(cf);
constexpr size_t kLdsPerWaveSpec = kScratchCf * sizeof(cf);

struct LdsLayout {
    float* hann;
    cf* t1024;
    cf* w64;       // [m][p] = W512^(8 m p)
    cf* scratch;   // this wave's
};

__device__ __forceinline__ LdsLayout lds_setup(const Tables& tb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsLayout L;
    L.hann = reinterpret_cast<float*>(smem);
    L.t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    L.w64 = L.t1024 + kT1024Pad;
    L.scratch = reinterpret_cast<cf*>(smem + kLdsTables + wave_id() * kLdsPerWaveSpec);
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreads) L.hann[i] = tb.hann[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kThreads) L.t1024[i] = tb.t1024[i];
    if (threadIdx.x < 64) L.w64[threadIdx.x] = tb.w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    __syncthreads();
    return L;
}

// ------------------------------------------------------------------------------------------------ K8
// one wave per (stream, frame); channels looped so an interleaved source is fetched by one wave
template <bool kUnit>
__global__ __launch_bounds__(kThreads, 4) void spectrum_kernel(SigViewD src, long long T, int ch, long long n_frames,
                                                           long long n_items, float* __restrict__ dst,
                                                           long long dst_ss, Tables tb)
{
    LdsLayout L = lds_setup(tb);
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * kWaves + wave_id();
    if (item >= n_items) return;
    const long long s = item / n_frames, f = item % n_frames;
    FftTw tw;
    load_fft_tw(tw, tb.w512, L.w64, lane);
    const int kl = kl_of_lane(lane);
    for (int c = 0; c < ch; c++) {
        ChanView in{src.base + s * src.ss + c * src.cs, src.fs, T};
        cf v[8];
        load_frame_windowed<kUnit>(v, in, f * NAE_HOP, L.hann, lane);
        fft512_fwd(v, L.scratch, tw, lane);
        const cf nyq = rfft_split(v, L.scratch, L.t1024, lane);
        float* o = dst + s * dst_ss + (f * ch + c) * NAE_FFT_BINS;
#pragma unroll
        for (int r = 0; r < 8; r++) o[kl + 64 * r] = __builtin_sqrtf(v[r].x * v[r].x + v[r].y * v[r].y);
        if (lane == 0) o[512] = __builtin_sqrtf(nyq.x * nyq.x + nyq.y * nyq.y);
    }
}

// interleaved-stereo fast path: one 16-byte load per lane and row fetches (L0 R0 L1 R1), so a frame is read once for both
// channels and the window is applied once; requires a 16-byte aligned stream base and an even stream stride.
// A wave walks kSpecChunk consecutive frames of one stream; both channels run FFT -> r2c split -> magnitude on the padded low-register FFT (stft_device.h).
// Loop order:  window(f) -> stores(f-1) -> loads(f+1) -> FFT / split / magnitudes of frame f.  Vector-memory operations of a
// wave retire in issue order and share one counter, so a load issued behind its own frame's 18 stores can only be waited
// for together with them — and a store takes microseconds to be acknowledged.  Here the wait in front of window(f) covers
// the loads of frame f and, older than them, only the stores of frame f-2.  The magnitudes of the previous frame ride along
// in 18 registers.  (tools/ubench/spec_abl.hip, profiles/r02_spectrum_ablation.md: 3.43 -> 3.16 ms on the C5 signal; stores
// that bypass L2 allocation — they are never read again by this kernel — another 0.1-0.3 ms.)
constexpr int kSpecChunk = 32;         // frames one wave walks when the launch has many rounds of waves (16 / 64 / 128 measured
                                       // within 1 %: profiles/r02_spectrum_ablation.md); small batches: spec_pick_chunk
constexpr int kSpecChunkLarge = 16;    // frames per chunk of a large (persistent, chunk-drawing) launch: 12-16 measured 1.8 % faster than 32 at C5, 8 5 % slower
                                       // (round 6, gpurun_out: 2.70-2.73 against 2.76-2.78 and 2.92 ms; a chunk's head re-reads 3/4 of a frame)
constexpr int kSpecChunkFine = 8;      // frames of the short chunks at the end of a large launch's work list
constexpr int kSpecStoreAux = 2;       // cache policy bits of the spectrum stores (2 = nt)
constexpr size_t kLdsTablesPad = NAE_FFT_N * sizeof(float) + (kT1024Pad + 64 + kTwaCf) * sizeof(cf);
constexpr size_t kLdsSpecStereo = kLdsTablesPad + kWaves * kPadScratchCf * sizeof(cf);

// kWide (dst 16-byte aligned, even stream stride): a frame's two spectra are 4104 CONTIGUOUS bytes of the output; the wave drops its
// magnitudes into its FFT scratch in output order (the scratch is idle between two frames) and writes them back with 16 bytes
// per lane on 16-byte boundaries — four whole-wave 1-KiB pieces per frame instead of eighteen 256-byte dword pieces at every
// 4-byte phase of a line (the shape tools/ubench/rw_mix.hip measures the chip's streaming rate with).  4104 = 8 mod 16: the 8
// bytes by which a frame overhangs its last piece are carried in a register and go out with the next frame's first piece.
// Work distribution.  The launch is PERSISTENT: at most two workgroups per CU (what fits), and every wave draws its next chunk of
// consecutive frames from a device counter until the list is empty.  Round 5 measured why (tools/experiments/r05_spec_stamps.*): with
// one chunk per wave and 14.75 rounds of workgroups per CU, the waves of a workgroup left its slot up to 20 % apart — 10 % of all
// wave-slot time idle behind waves that had finished, 6 % more between workgroups.  The list is GUIDED: chunks of `chunk_c` frames for
// the first `coarse_streams` streams, then chunks of `chunk_f` (short: a quarter of the re-read at a chunk's head, but the launch's tail is
// one SHORT chunk long) for the rest.  Which wave computes which chunk does not touch any result.
// A wave's FIRST item is the one of its position in the grid (no atomic: 4096 waves drawing from one address at the same moment cost a
// small batch 0.15 ms); item n_waves + counter++ comes next.  The host zeroes the counter on the launch's stream in front of every launch
// that draws (a 4-byte fill: ~3 us; an earlier form let the launch's last wave reset it — one launch that dies would have left every later
// one on the context with a stale count).
struct SpecWork {
    int chunk_c, chunk_f;
    unsigned cps_c, cps_f;         // chunks per stream, coarse / fine
    unsigned coarse_streams;
    unsigned n_coarse;             // coarse items = coarse_streams * cps_c
    unsigned n_items;
    unsigned n_waves;              // waves of the launch
    unsigned dynamic;              // 0: no more items than waves — every wave works on the item of its position and nobody touches the counters
    unsigned* counters;
};

template <bool kWide>
__global__ __launch_bounds__(kThreads, 4) void spectrum_stereo_kernel(const float* __restrict__ src, long long src_ss, long long n_frames, SpecWork work,
                                                                     float* __restrict__ dst, long long dst_ss, Tables tb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreads) hann[i] = tb.hann[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kThreads) t1024[i] = tb.t1024[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = tb.w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, tb.w512, threadIdx.x, kThreads);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    cf* scratch = reinterpret_cast<cf*>(smem + kLdsTablesPad) + wave_id() * kPadScratchCf;
    const FftLds L = make_fft_lds(scratch, twa, w64, lane);
    const cf* hw = reinterpret_cast<const cf*>(hann) + lane;
    const cf* tsp = t1024 + lane;
    const cf* tspm = t1024 + 512 - lane;                       // split twiddles of the mirrors, [-64 r]
    // the chunk this wave works on (wave-uniform; set per drawn item)
    int s = 0, f0 = 0, f1 = 0;
    const float* sbase = src;
    float* obase = dst;
    long long gbase = 0;
    unsigned drawn = 0;                                        // lane 0: what the counter handed out
    auto draw = [&]() { if (lane == 0) drawn = atomicAdd(&work.counters[0], 1u); };
    unsigned item = blockIdx.x * kWaves + (unsigned)wave_id();
    // magnitudes of one channel: [0..3] bins lane + 64 r, [4..7] their mirrors 512 - lane - 64 r, [8] bin 256 (lane 0)
    float ma[9], mb[9];
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    float carry = 0.0f;                  // kWide: mb[4] of the frame staged last = bin 512 (lane 0) / 511 (lane 1) of its second channel
    // LDS byte offset of the wave's scratch (wave-uniform) for ds_write_addtid_b32: address = M0 + offset + 4 * lane without an address
    // VGPR — half the cycles of ds_write_b32 on gfx950 (MI355X_MICROARCH.md, LDS).  The lane-ascending halves go that way.
    const unsigned scratch_off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)scratch);
    // stage_frame: behind a frame's second channel the scratch is idle — the magnitudes go in, in output order.
    // emit_frame: one iteration later, in the LDS round trip that fetches the window, they come back 16 bytes per lane and leave.
    auto stage_frame = [&](int fs) {
        const int phase = (int)((gbase + (long long)fs * (2 * NAE_FFT_BINS)) & 3);    // of the frame's first float, counted from dst: 0 or 2 (wave-uniform)
        float* st = reinterpret_cast<float*>(scratch) + phase;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"           // (an SALU write of M0 needs one wait state before an add-TID store reads it)
                     "ds_write_addtid_b32 %1\n\tds_write_addtid_b32 %2 offset:256\n\t"
                     "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\t"
                     "ds_write_addtid_b32 %5 offset:2052\n\tds_write_addtid_b32 %6 offset:2308\n\t"
                     "ds_write_addtid_b32 %7 offset:2564\n\tds_write_addtid_b32 %8 offset:2820"
                     :: "s"(scratch_off + 4u * (unsigned)phase), "v"(ma[0]), "v"(ma[1]), "v"(ma[2]), "v"(ma[3]),
                        "v"(mb[0]), "v"(mb[1]), "v"(mb[2]), "v"(mb[3]) : "m0", "memory");
        float* sm = st + 512 - lane;                                          // bin 512 - lane - 64 r at [-64 r]
#pragma unroll
        for (int r = 0; r < 4; r++) { sm[-64 * r] = ma[4 + r]; sm[NAE_FFT_BINS - 64 * r] = mb[4 + r]; }
        if (lane == 0) { st[256] = ma[8]; st[NAE_FFT_BINS + 256] = mb[8]; }
        if (phase != 0 && lane < 2) reinterpret_cast<float*>(scratch)[1 - lane] = carry;    // the previous frame's last 8 bytes
        carry = mb[4];
        wave_lds_sync();
    };
    auto emit_frame = [&](int fs, const u32x4 (&q)[5], bool first, bool last) {
        const int phase = (int)((gbase + (long long)fs * (2 * NAE_FFT_BINS)) & 3);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + (long long)fs * (2 * NAE_FFT_BINS) - phase, 0, -1, 0x00020000);
        if (phase != 0 && first) {
            // no frame in front of this one in the wave's chunk: its first piece is only the upper 8 bytes
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{q[0].z, q[0].w}, rs, 8, 0, kSpecStoreAux);
            else __builtin_amdgcn_raw_buffer_store_b128(q[0], rs, 16 * lane, 0, kSpecStoreAux);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(q[0], rs, 16 * lane, 0, kSpecStoreAux);
        }
#pragma unroll
        for (int i = 1; i < 4; i++) __builtin_amdgcn_raw_buffer_store_b128(q[i], rs, 16 * lane, 1024 * i, kSpecStoreAux);
        if (phase != 0) {
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(q[4], rs, 0, 4096, kSpecStoreAux);
        } else if (last) {
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{q[4].x, q[4].y}, rs, 0, 4096, kSpecStoreAux);
        }
    };
    const u32x4* stq = reinterpret_cast<const u32x4*>(scratch) + lane;   // staged pieces [64 i] per lane; floats 1024..1027 lie at piece 256 of the scratch (read with a lane-independent address)
    auto store_frame = [&](int fs) {
        // buffer stores: scalar descriptor of the frame's two spectra + one lane offset (no 64-bit per-lane addresses)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + ((long long)fs * 2) * NAE_FFT_BINS, 0, -1, 0x00020000);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float (&m)[9] = c == 0 ? ma : mb;
            const int co = c * NAE_FFT_BINS * 4;
#pragma unroll
            for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[r]), rs, 4 * lane, co + 256 * r, kSpecStoreAux);
#pragma unroll
            for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[4 + r]), rs, 1280 - 4 * lane, co + 768 - 256 * r, kSpecStoreAux);   // bin 512 - lane - 64 r
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[8]), rs, 1024, co, kSpecStoreAux);
        }
    };
    // one channel: FFT, r2c split delivering 2 X (no 1/2 factors: |2 X|^2 = 4 |X|^2 and sqrt(4 a) = 2 sqrt(a) are exact
    // scalings, so 0.5 * sqrt(.) is the canonical magnitude bit for bit, for |X| above ~1e-18), magnitudes.
    // Bins in mirror pairs: a lane computes X[k] and X[512 - k], k = lane + 64 r, r < 4, from A = Z[k] (its own register) and
    // B = Z[512 - k] (the upper half of Z, handed over through LDS: 4 + 1 writes and 4 reads instead of 8 + 1 and 8) — the
    // mirror's E and O are (Ex, -Ey) and (-Ox, Oy), exact negations and commuted sums of the canonical formula.
    auto channel = [&](cf (&v)[8], float (&mc)[9]) {
        fft512_pad(v, L);
#pragma unroll
        for (int r = 4; r < 8; r++) lds_st(L.nat + 64 * r, v[r]);
        if (lane == 0) scratch[512] = v[0];                  // the mirror of bin 0 is read like any other
        wave_lds_sync();
        cf B[4], tk[4], tm[4];
#pragma unroll
        for (int r = 0; r < 4; r++) B[r] = lds_ld(L.mir + 448 - 64 * r);
#pragma unroll
        for (int r = 0; r < 4; r++) { tk[r] = lds_ld(tsp + 64 * r); tm[r] = lds_ld(tspm - 64 * r); }
        {
            // bin 256 = its own mirror: A = B = Z[256], held by lane 0 in v[4]
            const cf z = v[4];
            const cf E = cf{z.x + z.x, z.y - z.y};
            const cf O = cf{z.x - z.x, z.y + z.y};
            const cf P = cmul_tw(O, t1024[256]);
            const cf X = cf{E.x + P.y, E.y - P.x};
            mc[8] = 0.5f * sqrt_rn(X.x * X.x + X.y * X.y);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const cf A = v[r];
            const cf E = cf{A.x + B[r].x, A.y - B[r].y};
            const cf O = cf{A.x - B[r].x, A.y + B[r].y};
            const cf P = cmul_tw(O, tk[r]);
            const cf X = cf{E.x + P.y, E.y - P.x};
            mc[r] = 0.5f * sqrt_rn(X.x * X.x + X.y * X.y);
            const cf Pm = cmul_tw(cf{-O.x, O.y}, tm[r]);
            const cf Xm = cf{E.x + Pm.y, -E.y - Pm.x};
            mc[4 + r] = 0.5f * sqrt_rn(Xm.x * Xm.x + Xm.y * Xm.y);
        }
        wave_lds_sync();
    };
    // consecutive frames overlap by 768 of 1024 sample-frames = 6 of the 8 rows of the FFT input layout (pair index
    // n = lane + 64 j, hop = 128 pairs = 2 rows): the raw samples stay in registers and a frame loads only its last 2 rows,
    // so every input byte is read once
#pragma unroll 1
    for (;;) {
    if (item >= work.n_items) break;
    {
        // (the divisions run on the vector ALU once per chunk: bring the wave-uniform results back to scalar registers)
        unsigned sv, kv;
        int ck;
        if (item < work.n_coarse) { sv = item / work.cps_c; kv = item - sv * work.cps_c; ck = work.chunk_c; }
        else { const unsigned j = item - work.n_coarse; const unsigned q = j / work.cps_f; sv = work.coarse_streams + q; kv = j - q * work.cps_f; ck = work.chunk_f; }
        s = __builtin_amdgcn_readfirstlane((int)sv);
        const int chunk = __builtin_amdgcn_readfirstlane(ck);
        f0 = __builtin_amdgcn_readfirstlane((int)kv) * chunk;
        f1 = f0 + chunk > (int)n_frames ? (int)n_frames : f0 + chunk;
        sbase = src + (long long)s * src_ss + 4 * lane;            // frames lie fully inside [0, T) by construction
        obase = dst + (long long)s * dst_ss;
        gbase = (long long)s * dst_ss;
    }
    float4 raw[8], pre[2];
    if (f0 < f1) {
        const float* base = sbase + 2 * ((long long)f0 * NAE_HOP);
#pragma unroll
        for (int j = 0; j < 8; j++) raw[j] = *reinterpret_cast<const float4*>(base + 256 * j);
    }
    if (work.dynamic) draw();                                      // the next item: its latency hides behind this chunk
#pragma unroll 1
    for (int f = f0; f < f1; f++) {
        cf v0[8], v1[8];
        u32x4 q[5];
        if (kWide && f > f0) {
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = stq[64 * i];
            q[4] = reinterpret_cast<const u32x4*>(scratch)[256];   // every lane reads the SAME 16 bytes (inside the wave's scratch); lane 0's copy is stored
        }
        if (kWide && f > f0) {
            emit_frame(f - 1, q, f - 1 == f0, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        cf w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = lds_ld(hw + 64 * j);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v0[j] = cf{raw[j].x * w[j].x, raw[j].z * w[j].y};
            v1[j] = cf{raw[j].y * w[j].x, raw[j].w * w[j].y};
        }
        if (!kWide && f > f0) store_frame(f - 1);
        if (f + 1 < f1) {
            const float* base = sbase + 2 * ((long long)(f + 1) * NAE_HOP);
            pre[0] = *reinterpret_cast<const float4*>(base + 256 * 6);
            pre[1] = *reinterpret_cast<const float4*>(base + 256 * 7);
        }
        channel(v0, ma);
        __builtin_amdgcn_sched_barrier(0);      // keep the two channels apart: interleaved, their live values exceed the register budget
        channel(v1, mb);
        __builtin_amdgcn_sched_barrier(0);
        if (kWide) stage_frame(f);
#pragma unroll
        for (int j = 0; j < 6; j++) raw[j] = raw[j + 2];
        raw[6] = pre[0];
        raw[7] = pre[1];
    }
    if (f1 > f0) {
        if (kWide) {
            u32x4 q[5];
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = stq[64 * i];
            q[4] = reinterpret_cast<const u32x4*>(scratch)[256];   // every lane reads the SAME 16 bytes (inside the wave's scratch); lane 0's copy is stored
            emit_frame(f1 - 1, q, f1 - 1 == f0, true);
            wave_lds_sync();
        }
        else store_frame(f1 - 1);
    }
    if (!work.dynamic) break;
    item = work.n_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)drawn);
    }   // next chunk
}

// ------------------------------------------------------------------------------------------------ K7
// pass 1: per-tile sum of phase increments.  sums[(sc * n_tiles + tile) * 520 + k]
// One wave per tile on the padded low-register FFT (stft_device.h; round 5 — rounds 1-4 ran the 100-VGPR swizzled FFT at four waves per SIMD): 80 VGPRs,
// three 8-wave workgroups per CU = six waves per SIMD, which is what the LDS allows (12 KB of tables + 8 x 4.5 KB of scratch per workgroup).
constexpr size_t kLdsPhase = kLdsTablesPad + kWaves * kPadScratchCf * sizeof(cf);
static_assert(3 * kLdsPhase <= 160 * 1024, "three workgroups per CU");
template <bool kUnit>
__global__ __launch_bounds__(kThreads, 6) void pv_phase_kernel(SigViewD src, PvParams p, long long n_items,
                                                              uint32_t* __restrict__ sums, Tables tb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreads) hann[i] = tb.hann[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kThreads) t1024[i] = tb.t1024[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = tb.w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, tb.w512, threadIdx.x, kThreads);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    cf* scratch = reinterpret_cast<cf*>(smem + kLdsTablesPad) + wave_id() * kPadScratchCf;
    const long long item = (long long)blockIdx.x * kWaves + wave_id();
    if (item >= n_items) return;
    const long long sc = item / p.n_tiles;
    const int tile = (int)(item % p.n_tiles);
    if (tile >= p.skip_from) return;                    // wave-uniform
    const long long s_idx = sc / p.ch;
    const int c = (int)(sc % p.ch);
    ChanView in{src.base + s_idx * src.ss + c * src.cs, src.fs, p.in_len};
    const int kl = kl_of_lane(lane);

    const long long f0 = p.f_origin + (long long)tile * p.tile;
    long long f1 = f0 + p.tile;
    if (f1 > p.f_stop) f1 = p.f_stop;

    uint32_t acc[9], qp[9], qa[9];
#pragma unroll
    for (int r = 0; r < 9; r++) { acc[r] = 0; qp[r] = 0; }
    cf v[8];
    long long s_prev = 0;
    // one analysis call site: frame f0-1 only primes qp (its increment belongs to the previous tile)
#pragma unroll 1
    for (long long f = (f0 > 0 ? f0 - 1 : 0); f < f1; f++) {
        const long long s = frame_start(p, f);
        load_frame_windowed<kUnit>(v, in, s, hann, lane);
        fft512_pad(v, make_fft_lds(scratch, twa, w64, lane));
        const cf nyq = rfft_split<true>(v, scratch, t1024, lane);   // 2 X: only phases are taken from it
        phases_of(v, nyq, qa);
        if (f < f0) {
            // priming frame: its increment belongs to the previous tile / call
        } else if (f == 0) {
#pragma unroll
            for (int r = 0; r < 9; r++) acc[r] = qa[r]; // the "increment" of frame 0 is its analysis phase
        } else {
            const unsigned d = (unsigned)(s - s_prev);
            const unsigned R = (d == (unsigned)p.d0) ? p.r_q24_0 : p.r_q24_1;
            phase_inc(qa, qp, acc, kl, d, R);
        }
#pragma unroll
        for (int r = 0; r < 9; r++) qp[r] = qa[r];
        s_prev = s;
    }
    uint32_t* o = sums + item * kT1024Pad;
#pragma unroll
    for (int r = 0; r < 8; r++) o[kl + 64 * r] = acc[r];
    if (lane == 0) o[512] = acc[8];
}

// pass 2: exclusive prefix over tiles, in place.  one thread per (stream-channel, bin).
// carry_in (optional): phase in front of tile 0, [n_sc][520]; carry_out (optional): phase behind the last tile.
__global__ void pv_scan_kernel(uint32_t* __restrict__ sums, long long n_sc, int n_tiles,
                               const uint32_t* __restrict__ carry_in, uint32_t* __restrict__ carry_out, int n_read)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long sc = t / kT1024Pad;
    const int k = (int)(t % kT1024Pad);
    if (sc >= n_sc || k >= NAE_FFT_BINS) return;
    uint32_t* p = sums + sc * n_tiles * (long long)kT1024Pad + k;
    uint32_t run = carry_in ? carry_in[sc * kT1024Pad + k] : 0u;
    int j = 0;
    // the loads do not depend on the running sum: fetch 8 tiles ahead, then prefix them
    for (; j + 8 <= n_read; j += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(long long)(j + u) * kT1024Pad];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            p[(long long)(j + u) * kT1024Pad] = run;
            run += v[u];
        }
    }
    for (; j < n_tiles; j++) {
        const uint32_t v = (j < n_read) ? p[(long long)j * kT1024Pad] : 0u;
        p[(long long)j * kT1024Pad] = run;
        run += v;
    }
    if (carry_out) carry_out[sc * kT1024Pad + k] = run;
}

// the same for many tiles per stream-channel (a long lone stream: thousands of tiles on a few stream-channels, where one thread per bin walks them
// one after the other): 16 threads per bin take a sixteenth of the tiles each — sum it, exchange the 16 sums through LDS, prefix the own part
// (modular integer sums: the split changes no bit).  One workgroup per (stream-channel, 64 bins).
constexpr int kScanChunks = 16;
__global__ __launch_bounds__(64 * kScanChunks) void pv_scan_chunked_kernel(uint32_t* __restrict__ sums, long long n_sc, int n_tiles,
                                                                          const uint32_t* __restrict__ carry_in, uint32_t* __restrict__ carry_out, int n_read)
{
    __shared__ uint32_t part[kScanChunks][64];
    const int kb = threadIdx.x & 63, ck = threadIdx.x >> 6;
    const long long sc = blockIdx.x / 9;
    const int k = (int)(blockIdx.x % 9) * 64 + kb;
    const bool valid = k < NAE_FFT_BINS;
    const int per = (n_tiles + kScanChunks - 1) / kScanChunks;
    const int j0 = ck * per, j1 = (j0 + per < n_tiles) ? j0 + per : n_tiles;
    const int r1 = j1 < n_read ? j1 : n_read;                       // tiles at or beyond n_read count as zero
    uint32_t* p = sums + sc * n_tiles * (long long)kT1024Pad + (valid ? k : 0);
    uint32_t sum = 0;
    if (valid) {
        int j = j0;
        for (; j + 8 <= r1; j += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[(long long)(j + u) * kT1024Pad];
#pragma unroll
            for (int u = 0; u < 8; u++) sum += v[u];
        }
        for (; j < r1; j++) sum += p[(long long)j * kT1024Pad];
    }
    part[ck][kb] = sum;
    __syncthreads();
    uint32_t run = (valid && carry_in) ? carry_in[sc * kT1024Pad + k] : 0u;
    for (int c2 = 0; c2 < ck; c2++) run += part[c2][kb];
    if (!valid) return;
    int j = j0;
    for (; j + 8 <= r1; j += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(long long)(j + u) * kT1024Pad];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            p[(long long)(j + u) * kT1024Pad] = run;
            run += v[u];
        }
    }
    for (; j < j1; j++) {
        const uint32_t v = (j < n_read) ? p[(long long)j * kT1024Pad] : 0u;
        p[(long long)j * kT1024Pad] = run;
        run += v;
    }
    if (carry_out && ck == kScanChunks - 1) carry_out[sc * kT1024Pad + k] = run;
}

// rate transposer: out[j] = sum_i tab(phase)[i] * v[idx - 7 + i],  pos = j * step (Q32.32)
struct RsParams { unsigned long long step_q32; long long src_len; long long out_len; int ch; long long j_begin; int rot; int tile_out; };

// LDS slot of coefficient row ph in the tiled kernels.  Neighbouring outputs advance the phase by a fixed amount (24.2
// rows at the default +3 semitones), and with rows stored in order every other lane of a 16-lane read group lands on the
// same banks (3-way conflicts on each of the 8 coefficient reads of an output).  Rotating the 7-bit row number right by
// `rot` makes the bank group of a row depend on bits rot..rot+3 of the phase; the launcher picks the rotation that
// spreads the lanes of this ratio best (rs_pick_rot).  Row NAE_RS_PHASES (read as "ph + 1" of the last phase) keeps its place.
__host__ __device__ __forceinline__ unsigned rs_slot(unsigned ph, int rot)
{
    return ph >= NAE_RS_PHASES ? ph : (ph >> rot) | ((ph & ((1u << rot) - 1u)) << (7 - rot));
}
static_assert(NAE_RS_PHASES == 128, "rs_slot rotates a 7-bit row number");

__global__ __launch_bounds__(256) void resample_kernel(SigViewD src, RsParams p, long long n_streams,
                                                      const float* __restrict__ tab, OutViewD out)
{
    __shared__ float stab[(NAE_RS_PHASES + 1) * NAE_RS_TAPS];
    for (int i = threadIdx.x; i < (NAE_RS_PHASES + 1) * NAE_RS_TAPS; i += blockDim.x) stab[i] = tab[i];
    __syncthreads();
    const long long j = p.j_begin + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = blockIdx.y;
    if (j >= p.out_len) return;
    const unsigned long long lo = (unsigned long long)j * p.step_q32;
    const unsigned long long hi = __umul64hi((unsigned long long)j, p.step_q32);
    const long long idx = (long long)((hi << 32) | (lo >> 32));
    const unsigned frac = (unsigned)lo;
    const unsigned ph = frac >> 25;
    const float alpha = (float)(frac & 0x1FFFFFFu) * (1.0f / 33554432.0f);
    const float* t0 = stab + ph * NAE_RS_TAPS;
    const float* t1 = t0 + NAE_RS_TAPS;
    float coef[NAE_RS_TAPS];
#pragma unroll
    for (int i = 0; i < NAE_RS_TAPS; i++) coef[i] = t0[i] + alpha * (t1[i] - t0[i]);
    for (int c = 0; c < p.ch; c++) {
        const float* v = src.base + s * src.ss + c * src.cs;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < NAE_RS_TAPS; i++) {
            const long long m = idx - (NAE_RS_TAPS / 2 - 1) + i;
            const float x = (m >= 0 && m < p.src_len) ? v[m * src.fs] : 0.0f;
            acc += coef[i] * x;
        }
        out.base[s * out.ss + c * out.cs + j * out.fs] = acc;
    }
}

// tiled rate transposer: a 256-thread workgroup produces kRsOut consecutive output frames of one stream; the
// source span it needs (kRsOut*rho + 16 samples per channel) is staged once into LDS with 16-byte loads, the
// 16 taps are read from LDS, and interleaved stereo output leaves as one 8-byte store per frame.
constexpr int kRsOut = 512;                      // output frames per workgroup when the wider tile does not fit.  Round 1: 256 -> 3.6 ms, 512 -> 3.07,
                                                 // 1024 -> 3.4 (C5 mix+transposer; LDS per workgroup sets the occupancy)
constexpr int kRsOutWide = 768;                  // round 4 (16-byte tap reads): 384 -> 2.58 ms, 512 -> 2.44, 768 -> 2.35, 1024 -> 2.70 on one box — three
                                                 // longer-lived workgroups per CU beat five; used while 4 streams of 768 rho + 28 frames fit the
                                                 // 1536-frame staging rows (rho <= 1.96), see rs_pick_tile
constexpr int kRsRow = 20;                       // LDS row stride of the coefficient table (16 taps + 4 pad): a 64-B
                                                 // stride maps every row to one of 4 bank slots (4-way conflicts on b128)
constexpr int kRsMaxSpan = 4096 + 32;            // staged source samples per channel (rho <= 4)

// coefficient table -> LDS (rows rotated by rs_slot): 16-byte loads, ALL of a thread's loads requested before the first LDS write.
// (The table is 8 KB per workgroup out of L2; one dword per thread and trip with a wait in every trip — what the first version
// did — put eight serial L2 round trips in front of every workgroup's staging loads.)
// Assumes (checked where it can be): workgroups of kRsThreads threads (every caller's __launch_bounds__), a 16-byte aligned
// table (hipMalloc), rows of a whole number of float4.
constexpr int kRsThreads = 256;
constexpr int kRsRowQuads = NAE_RS_TAPS / 4;                                 // float4 per coefficient row
static_assert(NAE_RS_TAPS % 4 == 0 && (kRsRowQuads & (kRsRowQuads - 1)) == 0, "a coefficient row is a power-of-two number of float4");
static_assert(kRsRow % 4 == 0 && kRsRow >= NAE_RS_TAPS, "LDS rows keep 16-byte alignment and hold a whole row");
__device__ __forceinline__ void rs_fill_table(float* stab, const float* __restrict__ tab, int rot)
{
    constexpr int kQuads = (NAE_RS_PHASES + 1) * kRsRowQuads;                // 516 float4
    constexpr int kTrips = (kQuads + kRsThreads - 1) / kRsThreads;
    const float4* t4 = reinterpret_cast<const float4*>(tab);
    float4 v[kTrips];
#pragma unroll
    for (int u = 0; u < kTrips; u++) {
        const int i = threadIdx.x + kRsThreads * u;
        v[u] = i < kQuads ? t4[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int u = 0; u < kTrips; u++) {
        const int i = threadIdx.x + kRsThreads * u;
        if (i < kQuads) *reinterpret_cast<float4*>(stab + rs_slot(i / kRsRowQuads, rot) * kRsRow + 4 * (i % kRsRowQuads)) = v[u];
    }
}


// Stereo taps.  The two channels are staged INTERLEAVED in LDS (one read feeds both accumulators), and NS streams share a workgroup:
// the 16 interpolated coefficients of an output depend on its position only, so they are built once (8 x 128-bit LDS reads, 48
// instructions) and applied to every stream.
// The taps are read 16 BYTES at a time (round 4).  With one ds_read_b64 per tap (rounds 1-3) the 32 lanes of a read group span 32 rho
// frames (38 at +3 semitones) = more words than the 64 banks: every tap read was a 2-way conflict, half of the kernel's LDS cycles
// (profiles/r03_v3_sq_stalls.md: bank-conflict share 0.49 at 86 % LDS-busy).  A ds_read_b128 is served 16 lanes at a time and moves
// 256 B per LDS cycle: 16 CONSECUTIVE outputs span 16 rho + 1 frames = at most 40 words at rho <= 1.2 — conflict-free — if
//   (i)  the 16 lanes of a hardware group hold consecutive outputs: the groups are not contiguous in the wave ({0-3,12-15,20-27},
//        {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md §LDS), so the output a lane works on is permuted (rs_slot_of_lane);
//   (ii) the reads are 16-byte aligned: a lane whose first tap frame o is odd reads from o - 1 and SHIFTS ITS COEFFICIENTS instead of
//        its data — cz[i'] = c[i' - (o & 1)], 17 positions, built once per output and used for all NS streams; the end positions
//        0 and 16 belong to one parity each and are added under a select, so a non-finite sample outside the window stays outside.
// The accumulation order per output is unchanged (tap 0 first, separate multiply and add): same bits.
__device__ __forceinline__ int rs_slot_of_lane(int lane)
{
    const int l5 = lane & 31;
    const bool b = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;        // second group of the half-wave
    const int pos = l5 < 4 ? l5 : l5 < 12 ? l5 - 4 : l5 < 16 ? l5 - 8 : l5 < 20 ? l5 - 8 : l5 < 28 ? l5 - 12 : l5 - 16;
    return (lane & 32) + (b ? 16 : 0) + pos;
}

template <int NS>
__device__ __forceinline__ void rs_apply_stereo(const float* stab, const float* stage, int span_alloc, const RsParams& p, long long j0,
                                                     long long j1, long long m_lo, const OutViewD& out, long long s0, long long n_streams)
{
    const bool out_pair = (out.cs == 1) && (out.fs == 2) && ((out.ss & 1) == 0) && ((reinterpret_cast<uintptr_t>(out.base) & 7) == 0);
    const int slot = (int)(threadIdx.x & ~63u) + rs_slot_of_lane((int)(threadIdx.x & 63u));
    for (long long jb = j0; jb < j1; jb += 256) {
        const long long j = jb + slot;
        const bool live = j < j1;
        const long long jj = live ? j : j1 - 1;                   // idle lanes repeat the last output (reads stay inside the staged span)
        const unsigned long long lo = (unsigned long long)jj * p.step_q32;
        const unsigned long long hi = __umul64hi((unsigned long long)jj, p.step_q32);
        const long long idx = (long long)((hi << 32) | (lo >> 32));
        const unsigned frac = (unsigned)lo;
        const unsigned ph = frac >> 25;
        const float alpha = (float)(frac & 0x1FFFFFFu) * (1.0f / 33554432.0f);
        const float4* t0 = reinterpret_cast<const float4*>(stab + rs_slot(ph, p.rot) * kRsRow);
        const float4* t1 = reinterpret_cast<const float4*>(stab + rs_slot(ph + 1, p.rot) * kRsRow);
        float coef[NAE_RS_TAPS];
#pragma unroll
        for (int q = 0; q < NAE_RS_TAPS / 4; q++) {
            const float4 a = t0[q], b = t1[q];
            coef[4 * q + 0] = a.x + alpha * (b.x - a.x);
            coef[4 * q + 1] = a.y + alpha * (b.y - a.y);
            coef[4 * q + 2] = a.z + alpha * (b.z - a.z);
            coef[4 * q + 3] = a.w + alpha * (b.w - a.w);
        }
        const int o = (int)(idx - (NAE_RS_TAPS / 2 - 1) - m_lo);
        const bool odd = (o & 1) != 0;
        float cz[NAE_RS_TAPS + 1];
        cz[0] = coef[0];
#pragma unroll
        for (int i = 1; i < NAE_RS_TAPS; i++) cz[i] = odd ? coef[i - 1] : coef[i];
        cz[NAE_RS_TAPS] = coef[NAE_RS_TAPS - 1];
        const int q0 = (o - (odd ? 1 : 0)) >> 1;                  // float4 index of the aligned pair of frames
#pragma unroll
        for (int k = 0; k < NS; k++) {
            if (s0 + k < n_streams) {
                const float4* st = reinterpret_cast<const float4*>(stage + (size_t)k * 2 * span_alloc) + q0;
                float4 x[NAE_RS_TAPS / 2];
#pragma unroll
                for (int t = 0; t < NAE_RS_TAPS / 2; t++) x[t] = st[t];
                const float2 xe = *reinterpret_cast<const float2*>(st + NAE_RS_TAPS / 2);      // frame 16 of the aligned window
                // position 0: even lanes only
                const float e0 = 0.0f + cz[0] * x[0].x, e1 = 0.0f + cz[0] * x[0].y;
                float a0 = odd ? 0.0f : e0, a1 = odd ? 0.0f : e1;
#pragma unroll
                for (int i = 1; i < NAE_RS_TAPS; i++) {
                    const float xl = (i & 1) ? x[i >> 1].z : x[i >> 1].x, xr = (i & 1) ? x[i >> 1].w : x[i >> 1].y;
                    a0 += cz[i] * xl;
                    a1 += cz[i] * xr;
                }
                // position 16: odd lanes only
                const float f0 = a0 + cz[NAE_RS_TAPS] * xe.x, f1 = a1 + cz[NAE_RS_TAPS] * xe.y;
                a0 = odd ? f0 : a0;
                a1 = odd ? f1 : a1;
                if (live) {
                    float* ob = out.base + (s0 + k) * out.ss;
                    if (out_pair) {
                        *reinterpret_cast<float2*>(ob + 2 * j) = float2{a0, a1};
                    } else {
                        ob[j * out.fs] = a0;
                        ob[out.cs + j * out.fs] = a1;
                    }
                }
            }
        }
    }
}

template <bool kStereo, int NS>
__global__ __launch_bounds__(kRsThreads) void resample_tile_kernel(SigViewD src, RsParams p, const float* __restrict__ tab,
                                                           OutViewD out, int span_alloc, long long n_streams)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    float* stab = reinterpret_cast<float*>(rs_smem);                           // (PHASES+1) x kRsRow
    float* stage = stab + (NAE_RS_PHASES + 1) * kRsRow;                        // stereo: [NS][span_alloc][2]; else [ch][span_alloc]
    rs_fill_table(stab, tab, p.rot);
    const long long s0 = (long long)blockIdx.y * NS;
    const long long j0 = p.j_begin + (long long)blockIdx.x * p.tile_out;
    long long j1 = j0 + p.tile_out;
    if (j1 > p.out_len) j1 = p.out_len;
    // source window [m_lo, m_hi) of this tile, m_lo rounded down to a multiple of 4 samples
    const unsigned long long lo0 = (unsigned long long)j0 * p.step_q32, hi0 = __umul64hi((unsigned long long)j0, p.step_q32);
    const unsigned long long lo1 = (unsigned long long)(j1 - 1) * p.step_q32, hi1 = __umul64hi((unsigned long long)(j1 - 1), p.step_q32);
    const long long idx_first = (long long)((hi0 << 32) | (lo0 >> 32));
    const long long idx_last = (long long)((hi1 << 32) | (lo1 >> 32));
    const long long m_lo = (idx_first - (NAE_RS_TAPS / 2 - 1)) & ~3ll;          // floor to 4 (arithmetic on negatives too)
    const long long m_hi = idx_last + NAE_RS_TAPS / 2 + 1;
    const int span = (int)(m_hi - m_lo);
    if (kStereo) {
#pragma unroll 1
        for (int k = 0; k < NS; k++) {
            if (s0 + k >= n_streams) break;
            const float* v0 = src.base + (s0 + k) * src.ss;
            const float* v1 = v0 + src.cs;
            float* stg = stage + (size_t)k * 2 * span_alloc;
            const bool vec = (src.fs == 1) && (((reinterpret_cast<uintptr_t>(v0) | reinterpret_cast<uintptr_t>(v1)) & 15) == 0);
            if (vec) {
                for (int i = 4 * threadIdx.x; i < span; i += 4 * 256) {
                    const long long m = m_lo + i;
                    float4 x, y;
                    if (m >= 0 && m + 4 <= p.src_len) {
                        x = *reinterpret_cast<const float4*>(v0 + m);
                        y = *reinterpret_cast<const float4*>(v1 + m);
                    } else {
                        float xe[4], ye[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const bool ok = (m + e >= 0) && (m + e < p.src_len);
                            xe[e] = ok ? v0[m + e] : 0.0f;
                            ye[e] = ok ? v1[m + e] : 0.0f;
                        }
                        x = float4{xe[0], xe[1], xe[2], xe[3]};
                        y = float4{ye[0], ye[1], ye[2], ye[3]};
                    }
                    *reinterpret_cast<float4*>(stg + 2 * i) = float4{x.x, y.x, x.y, y.y};
                    *reinterpret_cast<float4*>(stg + 2 * i + 4) = float4{x.z, y.z, x.w, y.w};
                }
            } else {
                for (int i = threadIdx.x; i < span; i += 256) {
                    const long long m = m_lo + i;
                    const bool ok = m >= 0 && m < p.src_len;
                    *reinterpret_cast<float2*>(stg + 2 * i) = float2{ok ? v0[m * src.fs] : 0.0f, ok ? v1[m * src.fs] : 0.0f};
                }
            }
        }
    } else {
        const float* v0 = src.base + s0 * src.ss;
        for (int c = 0; c < p.ch; c++) {
            const float* v = v0 + c * src.cs;
            float* st = stage + c * span_alloc;
            for (int i = threadIdx.x; i < span; i += 256) {
                const long long m = m_lo + i;
                st[i] = (m >= 0 && m < p.src_len) ? v[m * src.fs] : 0.0f;
            }
        }
    }
    __syncthreads();
    if (kStereo) {
        rs_apply_stereo<NS>(stab, stage, span_alloc, p, j0, j1, m_lo, out, s0, n_streams);
    } else {
        for (long long j = j0 + threadIdx.x; j < j1; j += 256) {
            const unsigned long long lo = (unsigned long long)j * p.step_q32;
            const unsigned long long hi = __umul64hi((unsigned long long)j, p.step_q32);
            const long long idx = (long long)((hi << 32) | (lo >> 32));
            const unsigned frac = (unsigned)lo;
            const unsigned ph = frac >> 25;
            const float alpha = (float)(frac & 0x1FFFFFFu) * (1.0f / 33554432.0f);
            const float* t0 = stab + rs_slot(ph, p.rot) * kRsRow;
            const float* t1 = stab + rs_slot(ph + 1, p.rot) * kRsRow;
            float coef[NAE_RS_TAPS];
#pragma unroll
            for (int i = 0; i < NAE_RS_TAPS; i++) coef[i] = t0[i] + alpha * (t1[i] - t0[i]);
            const int o = (int)(idx - (NAE_RS_TAPS / 2 - 1) - m_lo);
            for (int c = 0; c < p.ch; c++) {
                const float* st = stage + c * span_alloc + o;
                float a = 0.0f;
#pragma unroll
                for (int i = 0; i < NAE_RS_TAPS; i++) a += coef[i] * st[i];
                out.base[s0 * out.ss + c * out.cs + j * out.fs] = a;
            }
        }
    }
}

// mix(2) fused into the transposer's staging (graph: mix -> pitch with the transposer first).  The tile's source span
// is mixed on the fly from the two interleaved inputs — out = (0 + a*va) + b*vb, the reference's order
// (audio-amix.cpp:293-307) — written to the mix node's output for the frames this tile owns, and staged for the taps.
// Saves the mix kernel's launch and one read of the mix buffer.
struct MixFuseD {
    const float* a; long long a_ss;       // interleaved stereo, 16-byte aligned
    const float* b; long long b_ss;       // same; b_ss = 0: one buffer shared by every stream
    float va, vb;
    float* mix; long long mix_ss, mix_cs, mix_fs;
};

template <int NS>
__global__ __launch_bounds__(kRsThreads) void mix_resample_tile_kernel(MixFuseD f, RsParams p, const float* __restrict__ tab, OutViewD out,
                                                               int span_alloc, long long n_streams)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    float* stab = reinterpret_cast<float*>(rs_smem);
    float* stage = stab + (NAE_RS_PHASES + 1) * kRsRow;
    const long long s0 = (long long)blockIdx.y * NS;
    const long long j0 = (long long)blockIdx.x * p.tile_out;
    long long j1 = j0 + p.tile_out;
    const bool last = j1 >= p.out_len;
    if (last) j1 = p.out_len;
    auto idx_of = [&](long long j) {
        const unsigned long long lo = (unsigned long long)j * p.step_q32, hi = __umul64hi((unsigned long long)j, p.step_q32);
        return (long long)((hi << 32) | (lo >> 32));
    };
    const long long idx_first = idx_of(j0), idx_last = idx_of(j1 - 1);
    const long long m_lo = (idx_first - (NAE_RS_TAPS / 2 - 1)) & ~3ll;
    const long long m_hi = idx_last + NAE_RS_TAPS / 2 + 1;
    // frames of the mix output this tile writes: from its first output's position to the next tile's
    const long long own_lo = blockIdx.x == 0 ? 0 : idx_first;
    const long long own_hi = last ? p.src_len : idx_of(j1);
    const bool mix_planar = f.mix_fs == 1;
    // four frames per thread and trip (a 512-output tile spans ~640 source frames: one trip).  The loads of all NS
    // streams are issued before anything is stored — the staging is bound by round trips to HBM — and the mix leaves
    // as 16-byte stores per plane, like the stand-alone mix kernel's.
    const long long m_end = last ? (p.src_len > m_hi ? p.src_len : m_hi) : m_hi;
    auto load = [&](long long m, float4 (&xa)[NS][2], float4 (&xb)[NS][2]) {
        const bool inside = m >= 0 && m + 4 <= p.src_len;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            xa[k][0] = xa[k][1] = xb[k][0] = xb[k][1] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (s0 + k < n_streams) {
                const float* __restrict__ a = f.a + (s0 + k) * f.a_ss;
                const float* __restrict__ b = f.b + (s0 + k) * f.b_ss;
                if (inside) {
                    xa[k][0] = *reinterpret_cast<const float4*>(a + 2 * m);
                    xa[k][1] = *reinterpret_cast<const float4*>(a + 2 * m + 4);
                    xb[k][0] = *reinterpret_cast<const float4*>(b + 2 * m);
                    xb[k][1] = *reinterpret_cast<const float4*>(b + 2 * m + 4);
                } else {
                    float ta[8], tb[8];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const bool ok = m + e >= 0 && m + e < p.src_len;
                        ta[2 * e] = ok ? a[2 * (m + e)] : 0.0f;
                        ta[2 * e + 1] = ok ? a[2 * (m + e) + 1] : 0.0f;
                        tb[2 * e] = ok ? b[2 * (m + e)] : 0.0f;
                        tb[2 * e + 1] = ok ? b[2 * (m + e) + 1] : 0.0f;
                    }
                    xa[k][0] = float4{ta[0], ta[1], ta[2], ta[3]}; xa[k][1] = float4{ta[4], ta[5], ta[6], ta[7]};
                    xb[k][0] = float4{tb[0], tb[1], tb[2], tb[3]}; xb[k][1] = float4{tb[4], tb[5], tb[6], tb[7]};
                }
            }
        }
    };
    auto finish = [&](long long m, const float4 (&xa)[NS][2], const float4 (&xb)[NS][2]) {
        bool own[4], in[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            own[e] = m + e >= own_lo && m + e < own_hi;
            in[e] = m + e >= 0 && m + e < p.src_len;
        }
        const bool own_all = own[0] && own[3];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            if (s0 + k < n_streams) {
                float y[8];
                const float av[8] = {xa[k][0].x, xa[k][0].y, xa[k][0].z, xa[k][0].w, xa[k][1].x, xa[k][1].y, xa[k][1].z, xa[k][1].w};
                const float bv[8] = {xb[k][0].x, xb[k][0].y, xb[k][0].z, xb[k][0].w, xb[k][1].x, xb[k][1].y, xb[k][1].z, xb[k][1].w};
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = in[i >> 1] ? (0.0f + av[i] * f.va) + bv[i] * f.vb : 0.0f;
                float* stg = stage + (size_t)k * 2 * span_alloc + 2 * (m - m_lo);
                if (m + 3 < m_hi) {
                    *reinterpret_cast<float4*>(stg) = float4{y[0], y[1], y[2], y[3]};
                    *reinterpret_cast<float4*>(stg + 4) = float4{y[4], y[5], y[6], y[7]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (m + e < m_hi) *reinterpret_cast<float2*>(stg + 2 * e) = float2{y[2 * e], y[2 * e + 1]};
                }
                float* __restrict__ mx = f.mix + (s0 + k) * f.mix_ss;
                if (own_all && mix_planar) {
                    *reinterpret_cast<float4*>(mx + m) = float4{y[0], y[2], y[4], y[6]};
                    *reinterpret_cast<float4*>(mx + f.mix_cs + m) = float4{y[1], y[3], y[5], y[7]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (own[e]) { mx[(m + e) * f.mix_fs] = y[2 * e]; mx[f.mix_cs + (m + e) * f.mix_fs] = y[2 * e + 1]; }
                }
            }
        }
    };
    // first trip: its HBM loads are requested, THEN the coefficient table (L2) is fetched and laid out while they travel
    {
        float4 xa[NS][2], xb[NS][2];
        const long long m = m_lo + 4 * threadIdx.x;
        const bool has = m < m_end;
        if (has) load(m, xa, xb);
        rs_fill_table(stab, tab, p.rot);
        if (has) finish(m, xa, xb);
    }
    for (long long m = m_lo + 4 * threadIdx.x + 4 * 256; m < m_end; m += 4 * 256) {
        float4 xa[NS][2], xb[NS][2];
        load(m, xa, xb);
        finish(m, xa, xb);
    }
    __syncthreads();
    rs_apply_stereo<NS>(stab, stage, span_alloc, p, j0, j1, m_lo, out, s0, n_streams);
}

} // namespace nae

// ================================================================================================ host side
using namespace nae;

// rotation of the coefficient rows (rs_slot) with the fewest bank conflicts for this ratio: the 8 x 128-bit reads of an
// output are served 16 lanes at a time, a row's bank group is 5 * slot mod 16; count the worst multiplicity per group of
// lanes over the first 256 outputs
static int rs_pick_rot(unsigned long long step_q32)
{
    int best = 0;
    long best_cost = -1;
    for (int rot = 0; rot < 4; rot++) {
        long cost = 0;
        for (int g = 0; g < 16; g++) {
            int n0[16] = {0}, n1[16] = {0};
            unsigned seen0[16], seen1[16];          // distinct rows only: lanes reading the same row share the access
            for (int l = 0; l < 16; l++) {
                const unsigned long long pos = (unsigned long long)(16 * g + l) * step_q32;
                const unsigned ph = (unsigned)(pos & 0xFFFFFFFFull) >> 25;
                const unsigned s0 = rs_slot(ph, rot), s1 = rs_slot(ph + 1, rot);
                bool dup0 = false, dup1 = false;
                for (int k = 0; k < l; k++) { dup0 |= seen0[k] == s0; dup1 |= seen1[k] == s1; }
                seen0[l] = s0; seen1[l] = s1;
                if (!dup0) n0[(5 * s0) & 15]++;
                if (!dup1) n1[(5 * s1) & 15]++;
            }
            int m0 = 0, m1 = 0;
            for (int b = 0; b < 16; b++) { m0 = n0[b] > m0 ? n0[b] : m0; m1 = n1[b] > m1 ? n1[b] : m1; }
            cost += m0 + m1;
        }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rot; }
    }
    return best;
}

// frames per wave of the stereo spectrum kernel.  A CU holds 16 of its waves; with many rounds of waves the tail of the last
// round does not matter and 32 frames keep the waves short.  A small batch (an eighth of the C5 job is 1.8 rounds at 32) gets
// the chunk that fills a whole number of rounds: waves <= rounds x slots, the fewest frames per slot over the launch.
static int spec_pick_chunk(long long frames, long long n_streams, int n_cu)
{
    const long long slots = (long long)n_cu * 16;
    const long long waves32 = ((frames + kSpecChunk - 1) / kSpecChunk) * n_streams;
    if (waves32 >= 8 * slots || n_streams > slots) return kSpecChunk;
    long long best_chunk = kSpecChunk, best_cost = ((waves32 + slots - 1) / slots) * kSpecChunk;
    for (long long rounds = 1; rounds <= 8; rounds++) {
        const long long per_stream = rounds * slots / n_streams;            // chunks a stream may be cut into
        if (per_stream < 1) continue;
        long long chunk = (frames + per_stream - 1) / per_stream;
        if (chunk < 8) chunk = 8;
        if (chunk > 128) continue;                                           // (longer waves were not measured)
        const long long waves = ((frames + chunk - 1) / chunk) * n_streams;
        const long long cost = ((waves + slots - 1) / slots) * chunk;       // frames a slot walks over the launch
        if (cost < best_cost) { best_cost = cost; best_chunk = chunk; }
    }
    return (int)best_chunk;
}

// output frames per transposer workgroup and the staging row it needs (frames per channel, a multiple of 4)
static int rs_pick_tile(double rho, long long* span_need)
{
    auto need = [&](int tile) { return (long long)(tile * rho) + NAE_RS_TAPS + 8 + 4; };   // + 4: the aligned 16-byte tap reads look one pair of frames further
    const int tile = need(kRsOutWide) <= 1536 ? kRsOutWide : kRsOut;
    *span_need = need(tile);
    return tile;
}

static inline SigViewD to_view(const nae_sig* s)
{
    return SigViewD{static_cast<const float*>(s->base), (long long)s->stream_stride, (long long)s->chan_stride,
                    (long long)s->frame_stride};
}
static inline OutViewD to_out(const nae_sig* s)
{
    return OutViewD{static_cast<float*>(s->base), (long long)s->stream_stride, (long long)s->chan_stride,
                    (long long)s->frame_stride};
}

int nae_launch_spectrum(nae_ctx* ctx, const nae_sig* src, size_t T, int ch, size_t n_streams, float* dst,
                        size_t dst_stream_stride)
{
    const size_t F = nae_spectrum_frames(T);
    if (F == 0 || n_streams == 0) return NAE_OK;
    const long long items = (long long)(F * n_streams);
    const unsigned grid = (unsigned)((items + kWaves - 1) / kWaves);
    const size_t lds = kLdsTables + kWaves * kLdsPerWaveSpec;
    Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    const bool stereo_fast = ch == 2 && src->chan_stride == 1 && src->frame_stride == 2 && src->stream_stride % 4 == 0 &&
                             (reinterpret_cast<uintptr_t>(src->base) & 15) == 0 && !ctx->dbg_spec_generic;
    if (stereo_fast) {
        const long long slots = (long long)ctx->n_cu * 16;                   // waves a launch keeps resident (two workgroups per CU)
        SpecWork w{};
        w.counters = ctx->d_spec_ctr;
        const long long coarse_all = (((long long)F + kSpecChunk - 1) / kSpecChunk) * (long long)n_streams;
        if (coarse_all < 6 * slots) {
            // small batch: one list of equal chunks, sized so that the waves fill a whole number of rounds
            w.chunk_c = ctx->dbg_spec_chunk > 0 ? ctx->dbg_spec_chunk : spec_pick_chunk((long long)F, (long long)n_streams, ctx->n_cu);
            w.cps_c = (unsigned)(((long long)F + w.chunk_c - 1) / w.chunk_c);
            w.coarse_streams = (unsigned)n_streams;
            w.chunk_f = w.chunk_c;
            w.cps_f = w.cps_c;
        } else {
            // large batch: 16-frame chunks, and 8-frame chunks for the last streams — about four short chunks per resident wave, at most an
            // eighth of the job — so that the launch ends within one short chunk
            w.chunk_c = ctx->dbg_spec_chunk > 0 ? ctx->dbg_spec_chunk : kSpecChunkLarge;
            w.cps_c = (unsigned)(((long long)F + w.chunk_c - 1) / w.chunk_c);
            w.chunk_f = ctx->dbg_spec_fine > 0 ? ctx->dbg_spec_fine : kSpecChunkFine;
            w.cps_f = (unsigned)(((long long)F + w.chunk_f - 1) / w.chunk_f);
            long long fine_streams = ((ctx->dbg_spec_fine_rounds > 0 ? ctx->dbg_spec_fine_rounds : 4) * slots + w.cps_f - 1) / w.cps_f;
            if (fine_streams > (long long)n_streams / 8) fine_streams = (long long)n_streams / 8;
            w.coarse_streams = (unsigned)((long long)n_streams - fine_streams);
        }
        const long long n_coarse = (long long)w.coarse_streams * w.cps_c;
        const long long items = n_coarse + ((long long)n_streams - w.coarse_streams) * w.cps_f;
        if (items > 0x7fffffffll) return nae_fail(ctx, NAE_ERR_INVALID, "spectrum_stereo_kernel: too many chunks");
        w.n_coarse = (unsigned)n_coarse;
        w.n_items = (unsigned)items;
        long long groups = (items + kWaves - 1) / kWaves;
        if (groups > 2ll * ctx->n_cu) groups = 2ll * ctx->n_cu;
        w.n_waves = (unsigned)(groups * kWaves);
        w.dynamic = items > groups * kWaves ? 1u : 0u;
        if (w.dynamic) {
            (void)nae_use_device(ctx);
            const hipError_t e = hipMemsetAsync(ctx->d_spec_ctr, 0, sizeof(unsigned), ctx->stream);
            if (e != hipSuccess) return nae_check(ctx, e, "hipMemsetAsync(spectrum work counter)");
        }
        const bool wide = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && dst_stream_stride % 2 == 0 && !ctx->dbg_spec_narrow;
        if (wide)
            NAE_KLAUNCH(ctx, "spectrum_stereo_kernel", spectrum_stereo_kernel<true>, dim3((unsigned)groups), dim3(kThreads),
                        kLdsSpecStereo, ctx->stream, static_cast<const float*>(src->base), (long long)src->stream_stride,
                        (long long)F, w, dst, (long long)dst_stream_stride, tb);
        else
            NAE_KLAUNCH(ctx, "spectrum_stereo_kernel", spectrum_stereo_kernel<false>, dim3((unsigned)groups), dim3(kThreads),
                        kLdsSpecStereo, ctx->stream, static_cast<const float*>(src->base), (long long)src->stream_stride,
                        (long long)F, w, dst, (long long)dst_stream_stride, tb);
    }
    else if (src->frame_stride == 1)
        NAE_KLAUNCH(ctx, "spectrum_kernel", (spectrum_kernel<true>), dim3(grid), dim3(kThreads), lds, ctx->stream, to_view(src),
                    (long long)T, ch, (long long)F, items, dst, (long long)dst_stream_stride, tb);
    else
        NAE_KLAUNCH(ctx, "spectrum_kernel", (spectrum_kernel<false>), dim3(grid), dim3(kThreads), lds, ctx->stream, to_view(src),
                    (long long)T, ch, (long long)F, items, dst, (long long)dst_stream_stride, tb);
    return nae_check(ctx, hipGetLastError(), "spectrum_kernel");
}

static PvParams make_pv_params(const nae_stretch_plan& pl, size_t in_len, int ch, int tile, const nae_pv_segment* seg)
{
    PvParams p;
    p.ha_q24 = pl.ha_q24;
    p.in_len = (long long)in_len;
    p.frames = seg ? seg->f_limit : (long long)pl.frames;
    p.mid_len = seg ? seg->mid_limit : (long long)pl.mid_len;
    p.d0 = pl.d0;
    p.r_q24_0 = pl.r_q24[0];
    p.r_q24_1 = pl.r_q24[1];
    p.ch = ch;
    p.tile = tile;
    p.f_origin = seg ? seg->f_origin : 0;
    const long long cnt = seg ? seg->f_count : (long long)pl.frames;
    p.f_stop = p.f_origin + cnt;
    p.n_tiles = (int)((cnt + tile - 1) / tile);
    p.skip_from = p.n_tiles;
    p.phase_step = 1;
    p.phase_tiles = p.n_tiles;
    p.carry_out = nullptr;
    p.carry_frame = -1;
    p.base_zero = 0;
    return p;
}

size_t nae_pv_phase_workspace_bytes(size_t n_frames, int ch, size_t n_streams, int tile)
{
    const size_t n_tiles = (n_frames + tile - 1) / tile;
    return n_streams * ch * n_tiles * kT1024Pad * sizeof(uint32_t);
}

// pass 1 + 2: leaves the exclusive tile-prefix phases in `phase_ws` (one record per pass-1 tile).
// Pass 1 may use shorter tiles than pass 3 (`synth_tile` = a multiple of `tile`): its waves are independent, so short
// tiles keep the chip full on small batches, while pass 3 wants few long tiles (each re-analyses its frames).
// Only the sums in front of the last synthesis tile are needed, unless the phase behind the segment is carried on (a
// continued stream): nothing at all when the stream-channel is a single synthesis tile.
int nae_launch_pv_phase(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, int tile, int synth_tile, uint32_t* phase_ws, const nae_pv_segment* seg)
{
    if (tile <= 0 || synth_tile < tile || synth_tile % tile) return nae_fail(ctx, NAE_ERR_INVALID, "phase tile must divide the synthesis tile");
    PvParams p = make_pv_params(*pl, in_len, ch, tile, seg);
    const long long n_sc = (long long)n_streams * ch;
    if (n_sc * p.n_tiles == 0) return NAE_OK;
    const bool need_last = seg && seg->carry_out && !seg->carry_by_synth;
    const int step = synth_tile / tile;
    const int n_synth = (p.n_tiles + step - 1) / step;
    const int n_needed = need_last ? p.n_tiles : (n_synth - 1) * step;      // sums of tiles [0, n_needed) are used
    Tables tb{ctx->d_w512, ctx->d_t1024, ctx->d_hann};
    if (n_needed == 0) {
        // base phase of the only synthesis tile (record 0 of each stream-channel): the carried phase, or zero
        hipError_t e = hipSuccess;
        if (!(seg && seg->carry_in)) {
            // nothing carried in and a single synthesis tile: pass 3 starts from zero by itself (PvParams::base_zero) — no memset launch
            if (p.n_tiles != 1) e = hipMemsetAsync(phase_ws, 0, (size_t)n_sc * p.n_tiles * kT1024Pad * sizeof(uint32_t), ctx->stream);
        }
        else if (p.n_tiles == 1)
            e = hipMemcpyAsync(phase_ws, seg->carry_in, (size_t)n_sc * kT1024Pad * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
        else
            e = hipMemcpy2DAsync(phase_ws, (size_t)p.n_tiles * kT1024Pad * sizeof(uint32_t), seg->carry_in, kT1024Pad * sizeof(uint32_t),
                                 kT1024Pad * sizeof(uint32_t), (size_t)n_sc, hipMemcpyDeviceToDevice, ctx->stream);
        return nae_check(ctx, e, "phase base init");
    }
    {
        // items are (stream-channel, tile) with tile fastest; the kernel skips the tiles whose sums are not needed
        PvParams pp = p;
        pp.skip_from = n_needed;
        const long long items = n_sc * p.n_tiles;
        const unsigned grid = (unsigned)((items + kWaves - 1) / kWaves);
        const size_t lds = kLdsPhase;
        if (src->frame_stride == 1)
            NAE_KLAUNCH(ctx, "pv_phase_kernel", (pv_phase_kernel<true>), dim3(grid), dim3(kThreads), lds, ctx->stream,
                        to_view(src), pp, items, phase_ws, tb);
        else
            NAE_KLAUNCH(ctx, "pv_phase_kernel", (pv_phase_kernel<false>), dim3(grid), dim3(kThreads), lds, ctx->stream,
                        to_view(src), pp, items, phase_ws, tb);
        int rc = nae_check(ctx, hipGetLastError(), "pv_phase_kernel");
        if (rc) return rc;
    }
    {
        if (p.n_tiles >= 256 && n_sc * 9 <= 0x7fffffffll) {
            NAE_KLAUNCH(ctx, "pv_scan_kernel", pv_scan_chunked_kernel, dim3((unsigned)(n_sc * 9)), dim3(64 * kScanChunks), 0, ctx->stream, phase_ws, n_sc, p.n_tiles,
                        seg ? seg->carry_in : nullptr, seg ? seg->carry_out : nullptr, n_needed);
            return nae_check(ctx, hipGetLastError(), "pv_scan_kernel");
        }
        const long long threads = n_sc * kT1024Pad;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        NAE_KLAUNCH(ctx, "pv_scan_kernel", pv_scan_kernel, dim3(grid), dim3(256), 0, ctx->stream, phase_ws, n_sc, p.n_tiles,
                    seg ? seg->carry_in : nullptr, seg ? seg->carry_out : nullptr, n_needed);
        return nae_check(ctx, hipGetLastError(), "pv_scan_kernel");
    }
}

int nae_launch_pv_synth(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, int tile, int phase_tile, const uint32_t* phase_ws, const nae_sig* out,
                        const nae_pv_segment* seg, int frames_per_step)
{
    if (phase_tile <= 0 || tile < phase_tile || tile % phase_tile) return nae_fail(ctx, NAE_ERR_INVALID, "phase tile must divide the synthesis tile");
    PvParams p = make_pv_params(*pl, in_len, ch, tile, seg);
    const long long cnt = p.f_stop - p.f_origin;
    p.phase_step = tile / phase_tile;
    p.phase_tiles = (int)((cnt + phase_tile - 1) / phase_tile);
    p.base_zero = (p.n_tiles == 1 && p.phase_tiles == 1 && !(seg && seg->carry_in)) ? 1 : 0;
    if (seg && seg->carry_by_synth && seg->carry_out) {
        if (p.n_tiles != 1) return nae_fail(ctx, NAE_ERR_INVALID, "carry_by_synth needs a single synthesis tile");
        p.carry_out = seg->carry_out;
        p.carry_frame = p.f_stop - 1;
    }
    return nae_launch_pv_pipe(ctx, p, to_view(src), (long long)n_streams * ch, phase_ws, to_out(out), src->frame_stride == 1, frames_per_step);
}

// outputs [j_begin, j_end)
int nae_launch_resample(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t src_len, int ch,
                        size_t n_streams, const float* d_tab, const nae_sig* out, size_t j_begin, size_t j_end)
{
    if (j_end <= j_begin || n_streams == 0) return NAE_OK;
    const size_t count = j_end - j_begin;
    // tiled kernel while one tile's source span fits the staging buffer (rho <= 4), else the direct kernel
    const double rho = (double)pl->step_q32 / 4294967296.0;
    long long span_need = 0;
    const int tile_out = rs_pick_tile(rho, &span_need);
    RsParams p{pl->step_q32, (long long)src_len, (long long)j_end, ch, (long long)j_begin, rs_pick_rot(pl->step_q32), tile_out};
    const bool tiled = span_need <= kRsMaxSpan && !ctx->dbg_rs_direct;
    const int span_alloc = (int)((span_need + 3) & ~3ll);
    // stereo batches: 4 streams per workgroup share the per-output coefficients (if their staging fits LDS)
    const int group = (tiled && ch == 2 && n_streams >= 4 && span_alloc <= 1536 && !ctx->dbg_rs_single) ? 4 : 1;
    const size_t lds = ((NAE_RS_PHASES + 1) * kRsRow + (size_t)ch * span_alloc * group) * sizeof(float);
    const unsigned gx = tiled ? (unsigned)((count + tile_out - 1) / tile_out) : (unsigned)((count + 255) / 256);
    // blockIdx.y is limited to 65535
    const size_t per_launch = (size_t)65535 * group;
    for (size_t s0 = 0; s0 < n_streams; s0 += per_launch) {
        const size_t ns = (n_streams - s0 < per_launch) ? n_streams - s0 : per_launch;
        SigViewD sv = to_view(src);
        OutViewD ov = to_out(out);
        sv.base += (long long)s0 * sv.ss;
        ov.base += (long long)s0 * ov.ss;
        if (tiled && ch == 2 && group == 4)
            NAE_KLAUNCH(ctx, "resample_tile_kernel", (resample_tile_kernel<true, 4>), dim3(gx, (unsigned)((ns + 3) / 4)), dim3(256), lds,
                        ctx->stream, sv, p, d_tab, ov, span_alloc, (long long)ns);
        else if (tiled && ch == 2)
            NAE_KLAUNCH(ctx, "resample_tile_kernel", (resample_tile_kernel<true, 1>), dim3(gx, (unsigned)ns), dim3(256), lds, ctx->stream,
                        sv, p, d_tab, ov, span_alloc, (long long)ns);
        else if (tiled)
            NAE_KLAUNCH(ctx, "resample_tile_kernel", (resample_tile_kernel<false, 1>), dim3(gx, (unsigned)ns), dim3(256), lds, ctx->stream,
                        sv, p, d_tab, ov, span_alloc, (long long)ns);
        else
            NAE_KLAUNCH(ctx, "resample_kernel", resample_kernel, dim3(gx, (unsigned)ns), dim3(256), 0, ctx->stream, sv, p,
                        (long long)ns, d_tab, ov);
        int rc = nae_check(ctx, hipGetLastError(), "resample kernel");
        if (rc) return rc;
    }
    return NAE_OK;
}

// mix(2) + transposer in one launch (see mix_resample_tile_kernel).  Returns 1 when the shapes do not fit the fused
// kernel (the caller then runs the two nodes separately), 0 on success, < 0 on error.
int nae_launch_mix_resample(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* a, const nae_sig* b, float va, float vb,
                            const nae_sig* mix_out, size_t S, size_t n_streams, const float* d_tab, const nae_sig* out)
{
    const double rho = (double)pl->step_q32 / 4294967296.0;
    long long span_need = 0;
    const int tile_out = rs_pick_tile(rho, &span_need);
    const int span_alloc = (int)((span_need + 3) & ~3ll);
    auto inter16 = [](const nae_sig* v) {
        return v->chan_stride == 1 && v->frame_stride == 2 && (reinterpret_cast<uintptr_t>(v->base) & 15) == 0 && (v->stream_stride & 3) == 0;
    };
    // planar mix output: 16-byte stores per plane; interleaved (or any other) layout: scalar stores
    const bool mix_ok = mix_out->frame_stride != 1 ||
                        ((reinterpret_cast<uintptr_t>(mix_out->base) & 15) == 0 && (mix_out->stream_stride & 3) == 0 &&
                         (mix_out->chan_stride & 3) == 0);
    if (ctx->dbg_no_mix_fuse || !inter16(a) || !inter16(b) || !mix_ok || span_need > kRsMaxSpan || span_alloc > 1536 || n_streams == 0 ||
        pl->mid_len == 0 || S == 0)
        return 1;
    RsParams p{pl->step_q32, (long long)S, (long long)pl->mid_len, 2, 0, rs_pick_rot(pl->step_q32), tile_out};
    const size_t lds = ((NAE_RS_PHASES + 1) * kRsRow + (size_t)2 * span_alloc * 4) * sizeof(float);
    const unsigned gx = (unsigned)((pl->mid_len + tile_out - 1) / tile_out);
    const size_t per_launch = (size_t)65535 * 4;
    for (size_t s0 = 0; s0 < n_streams; s0 += per_launch) {
        const size_t ns = (n_streams - s0 < per_launch) ? n_streams - s0 : per_launch;
        MixFuseD f;
        f.a = static_cast<const float*>(a->base) + s0 * a->stream_stride; f.a_ss = (long long)a->stream_stride;
        f.b = static_cast<const float*>(b->base) + s0 * b->stream_stride; f.b_ss = (long long)b->stream_stride;
        f.va = va; f.vb = vb;
        f.mix = static_cast<float*>(mix_out->base) + s0 * mix_out->stream_stride;
        f.mix_ss = (long long)mix_out->stream_stride; f.mix_cs = (long long)mix_out->chan_stride; f.mix_fs = (long long)mix_out->frame_stride;
        OutViewD ov = to_out(out);
        ov.base += (long long)s0 * ov.ss;
        NAE_KLAUNCH(ctx, "mix_resample_tile_kernel", (mix_resample_tile_kernel<4>), dim3(gx, (unsigned)((ns + 3) / 4)), dim3(256), lds,
                    ctx->stream, f, p, d_tab, ov, span_alloc, (long long)ns);
        int rc = nae_check(ctx, hipGetLastError(), "mix_resample_tile_kernel");
        if (rc) return rc;
    }
    return NAE_OK;
}

