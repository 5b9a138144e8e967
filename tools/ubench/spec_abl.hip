// Micro-benchmark (not part of the product): the stereo spectrum kernel (kernels_stft.hip: spectrum_stereo2_kernel) with
// one stage removed at a time, on the C5-sized signal, to see what its time is made of.  Results are wrong by
// construction in every mode but 0.
//   mode 0 full | 1 no global stores | 2 no global loads | 3 no split / magnitude (stores FFT output parts)
//   mode 4 no FFT (split + magnitude of the windowed samples) | 5 loads + stores only
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o spec_abl spec_abl.hip && ./spec_abl
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace nae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChunk = 32, kWavesWg = 12, kThreadsWg = 64 * kWavesWg, kT1024Pad = 520;
constexpr size_t kLdsTables = NAE_FFT_N * sizeof(float) + (kT1024Pad + 64 + kTwaCf) * sizeof(cf);
constexpr size_t kLds = kLdsTables + kWavesWg * kPadScratchCf * sizeof(cf);

template <int kAbl>
__global__ __launch_bounds__(kThreadsWg, 6) void spec_kernel(const float* __restrict__ src, long long src_ss, long long n_frames, long long chunks_per_stream,
                                                            long long n_items, float* __restrict__ dst, long long dst_ss, const cf* w512, const cf* t1024g,
                                                            const float* hanng)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreadsWg) hann[i] = hanng[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kThreadsWg) t1024[i] = t1024g[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, w512, threadIdx.x, kThreadsWg);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long item = (long long)blockIdx.x * kWavesWg + wave;
    if (item >= n_items) return;
    cf* scratch = reinterpret_cast<cf*>(smem + kLdsTables) + wave * kPadScratchCf;
    const FftLds L = make_fft_lds(scratch, twa, w64, lane);
    const float2* hw = reinterpret_cast<const float2*>(hann) + lane;
    const cf* tsp = t1024 + lane;
    const long long s = item / chunks_per_stream;
    const long long f0 = (item % chunks_per_stream) * kChunk;
    long long f1 = f0 + kChunk;
    if (f1 > n_frames) f1 = n_frames;
    const float* sbase = src + s * src_ss + 4 * lane;
    float* obase = dst + s * dst_ss;
    float acc = 0.0f;
#pragma unroll 1
    for (long long f = f0; f < f1; f++) {
        const float* base = sbase + 2 * (f * NAE_HOP);
        cf v0[8], v1[8];
        {
            float4 raw[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (kAbl == 2 || kAbl == 6 || kAbl == 8 || kAbl == 12 || (kAbl == 11 && !(wave & 1))) raw[j] = float4{(float)(lane + j) * 1e-3f + acc, (float)f * 1e-6f, (float)j * 0.01f, (float)lane * 2e-3f};
                else raw[j] = *reinterpret_cast<const float4*>(base + 256 * j);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float2 w = hw[64 * j];
                v0[j] = cf{raw[j].x * w.x, raw[j].z * w.y};
                v1[j] = cf{raw[j].y * w.x, raw[j].w * w.y};
            }
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + (f * 2) * NAE_FFT_BINS, 0, -1, 0x00020000);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) v0[j] = v1[j];
            }
            const int cofs = c * NAE_FFT_BINS * 4;
            constexpr int kAux = (kAbl >= 9) ? 2 : 0;   // nt
            if (kAbl == 7 || (kAbl == 11 && (wave & 1))) { acc += v0[0].x + v0[3].y + v0[7].x; continue; }
            if (kAbl == 8) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                // the same 2052 bytes per channel-frame as two 16-byte-per-lane stores + tail
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v0[0].x), __float_as_uint(v0[1].x), __float_as_uint(v0[2].x), __float_as_uint(v0[3].x)}, rs, 16 * lane, cofs, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v0[4].x), __float_as_uint(v0[5].x), __float_as_uint(v0[6].x), __float_as_uint(v0[7].x)}, rs, 1024 + 16 * lane, cofs, 0);
                if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[0].y), rs, 2048, cofs, 0);
                continue;
            }
            if (kAbl == 5 || kAbl == 6 || kAbl == 9 || kAbl == 11) {
#pragma unroll
                for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[r].x + v0[r].y), rs, 4 * lane + 256 * r, cofs, kAux);
                if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[0].x), rs, 2048, cofs, kAux);
                continue;
            }
            if (kAbl != 4) fft512_pad(v0, L);
            if (kAbl == 3) {
#pragma unroll
                for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[r].x + v0[r].y), rs, 4 * lane + 256 * r, cofs, 0);
                if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[0].x), rs, 2048, cofs, 0);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, v0[r]);
            if (lane == 0) scratch[512] = v0[0];
            wave_lds_sync();
            const cf z0 = scratch[0];
            cf nyq;
            {
                const cf E = cf{z0.x + z0.x, z0.y - z0.y};
                const cf O = cf{z0.x - z0.x, z0.y + z0.y};
                const cf P = cmul_tw(O, t1024[512]);
                nyq = cf{E.x + P.y, E.y - P.x};
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const cf A = v0[r], B = lds_ld(L.mir + 448 - 64 * r);
                const cf E = cf{A.x + B.x, A.y - B.y};
                const cf O = cf{A.x - B.x, A.y + B.y};
                const cf P = cmul_tw(O, lds_ld(tsp + 64 * r));
                const cf X = cf{E.x + P.y, E.y - P.x};
                const float m = 0.5f * sqrt_rn(X.x * X.x + X.y * X.y);
                if (kAbl == 1 || kAbl == 12) acc += m;
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rs, 4 * lane + 256 * r, cofs, kAux);
            }
            const float mn = 0.5f * sqrt_rn(nyq.x * nyq.x + nyq.y * nyq.y);
            if (kAbl == 1 || kAbl == 12) acc += mn;
            else if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mn), rs, 2048, cofs, kAux);
            wave_lds_sync();
        }
    }
    if (kAbl == 1 || kAbl == 2 || kAbl == 7 || kAbl == 12 || (kAbl == 11 && (wave & 1))) obase[(f0 * 2) * NAE_FFT_BINS + lane] = acc;
}

// mode 13: the full kernel with the loop re-ordered so that no load is ever issued behind its own frame's stores:
//   window(f) -> stores(f-1) -> loads(f+1) -> FFT / split / magnitudes of frame f   (128 VGPRs, 4 waves per SIMD)
constexpr int kWavesWgR = 8, kThreadsWgR = 64 * kWavesWgR;
constexpr size_t kLdsR = kLdsTables + kWavesWgR * kPadScratchCf * sizeof(cf);
__global__ __launch_bounds__(kThreadsWgR, 4) void spec_kernel_reordered(const float* __restrict__ src, long long src_ss, long long n_frames,
                                                                        long long chunks_per_stream, long long n_items, float* __restrict__ dst,
                                                                        long long dst_ss, const cf* w512, const cf* t1024g, const float* hanng)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreadsWgR) hann[i] = hanng[i];
    for (int i = threadIdx.x; i < NAE_FFT_BINS; i += kThreadsWgR) t1024[i] = t1024g[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, w512, threadIdx.x, kThreadsWgR);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long item = (long long)blockIdx.x * kWavesWgR + wave;
    if (item >= n_items) return;
    cf* scratch = reinterpret_cast<cf*>(smem + kLdsTables) + wave * kPadScratchCf;
    const FftLds L = make_fft_lds(scratch, twa, w64, lane);
    const cf* hw = reinterpret_cast<const cf*>(hann) + lane;
    const cf* tsp = t1024 + lane;
    const int s = __builtin_amdgcn_readfirstlane((int)(item / chunks_per_stream));
    const int f0 = __builtin_amdgcn_readfirstlane((int)(item % chunks_per_stream)) * kChunk;
    const int f1 = f0 + kChunk > (int)n_frames ? (int)n_frames : f0 + kChunk;
    const float* sbase = src + (long long)s * src_ss + 4 * lane;
    float* obase = dst + (long long)s * dst_ss;
    float ma[9], mb[9];
    auto store_frame = [&](int fs) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + ((long long)fs * 2) * NAE_FFT_BINS, 0, -1, 0x00020000);
#pragma unroll
        for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ma[r]), rs, 4 * lane, 256 * r, 0);
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ma[8]), rs, 2048, 0, 0);
#pragma unroll
        for (int r = 0; r < 8; r++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mb[r]), rs, 4 * lane, NAE_FFT_BINS * 4 + 256 * r, 0);
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mb[8]), rs, 2048, NAE_FFT_BINS * 4, 0);
    };
    auto channel = [&](cf (&v)[8], float (&mc)[9]) {
        fft512_pad(v, L);
#pragma unroll
        for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, v[r]);
        if (lane == 0) scratch[512] = v[0];
        wave_lds_sync();
        const cf z0 = scratch[0];
        {
            const cf E = cf{z0.x + z0.x, z0.y - z0.y};
            const cf O = cf{z0.x - z0.x, z0.y + z0.y};
            const cf P = cmul_tw(O, t1024[512]);
            const cf nyq = cf{E.x + P.y, E.y - P.x};
            mc[8] = 0.5f * sqrt_rn(nyq.x * nyq.x + nyq.y * nyq.y);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const cf A = v[r], B = lds_ld(L.mir + 448 - 64 * r);
            const cf E = cf{A.x + B.x, A.y - B.y};
            const cf O = cf{A.x - B.x, A.y + B.y};
            const cf P = cmul_tw(O, lds_ld(tsp + 64 * r));
            const cf X = cf{E.x + P.y, E.y - P.x};
            mc[r] = 0.5f * sqrt_rn(X.x * X.x + X.y * X.y);
        }
        wave_lds_sync();
    };
    float4 raw[8];
    if (f0 < f1) {
        const float* base = sbase + 2 * ((long long)f0 * NAE_HOP);
#pragma unroll
        for (int j = 0; j < 8; j++) raw[j] = *reinterpret_cast<const float4*>(base + 256 * j);
    }
#pragma unroll 1
    for (int f = f0; f < f1; f++) {
        cf v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const cf w = lds_ld(hw + 64 * j);
            v0[j] = cf{raw[j].x * w.x, raw[j].z * w.y};
            v1[j] = cf{raw[j].y * w.x, raw[j].w * w.y};
        }
        if (f > f0) store_frame(f - 1);
        if (f + 1 < f1) {
            const float* base = sbase + 2 * ((long long)(f + 1) * NAE_HOP);
#pragma unroll
            for (int j = 0; j < 8; j++) raw[j] = *reinterpret_cast<const float4*>(base + 256 * j);
        }
        channel(v0, ma);
        __builtin_amdgcn_sched_barrier(0);
        channel(v1, mb);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (f1 > f0) store_frame(f1 - 1);
}

static float run_reordered(const float* src, long long S, long long n_streams, float* dst, const cf* w, const cf* t, const float* h)
{
    const long long F = (S - 1024) / 256 + 1, chunks = (F + kChunk - 1) / kChunk, items = chunks * n_streams;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(spec_kernel_reordered, dim3((unsigned)((items + kWavesWgR - 1) / kWavesWgR)), dim3(kThreadsWgR), kLdsR, 0, src, 2 * S, F, chunks, items,
                           dst, F * 2 * 513, w, t, h);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    return best;
}

template <int kAbl>
static float run(const float* src, long long S, long long n_streams, float* dst, const cf* w, const cf* t, const float* h)
{
    const long long F = (S - 1024) / 256 + 1, chunks = (F + kChunk - 1) / kChunk, items = chunks * n_streams;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(spec_kernel<kAbl>, dim3((unsigned)((items + kWavesWg - 1) / kWavesWg)), dim3(kThreadsWg), kLds, 0, src, 2 * S, F, chunks, items, dst,
                           F * 2 * 513, w, t, h);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const long long S = 403635, n_streams = 1024, F = (S - 1024) / 256 + 1;
    std::vector<cf> w(512), t(520);
    std::vector<float> h(1024);
    for (int k = 0; k < 512; k++) w[k] = cf{(float)cos(6.283185307179586 * k / 512.0), (float)(-sin(6.283185307179586 * k / 512.0))};
    for (int k = 0; k <= 512; k++) t[k] = cf{(float)cos(6.283185307179586 * k / 1024.0), (float)(-sin(6.283185307179586 * k / 1024.0))};
    for (int n = 0; n < 1024; n++) h[n] = (float)(0.5 - 0.5 * cos(6.283185307179586 * n / 1024.0));
    cf *dw, *dt; float *dh, *src, *dst;
    CK(hipMalloc(&dw, 512 * 8)); CK(hipMalloc(&dt, 520 * 8)); CK(hipMalloc(&dh, 4096));
    CK(hipMemcpy(dw, w.data(), 512 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, t.data(), 520 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, h.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMalloc(&src, n_streams * S * 2 * 4)); CK(hipMalloc(&dst, n_streams * F * 2 * 513 * 4));
    std::vector<float> host(1 << 20);
    for (size_t i = 0; i < host.size(); i++) host[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
    for (long long off = 0; off < n_streams * S * 2; off += (long long)host.size())
        CK(hipMemcpy(src + off, host.data(), std::min<long long>(host.size(), n_streams * S * 2 - off) * 4, hipMemcpyHostToDevice));
    const char* names[13] = {"full", "no global stores", "no global loads", "no split / magnitude", "no FFT", "loads + window + stores only", "stores only (dword per lane, as shipped)", "loads + window only", "stores only, 16 B per lane", "loads + window + nt stores", "full with nt stores", "odd waves: loads + window only; even waves: stores only (half of each)", "compute only (no global loads, no global stores)"};
    printf("# stereo spectrum kernel, C5-sized signal (1024 streams x %lld frames x 2 channels), one stage removed at a time\n\n| mode | ms |\n|---|---|\n", F);
    float r[13] = {run<0>(src, S, n_streams, dst, dw, dt, dh), run<1>(src, S, n_streams, dst, dw, dt, dh), run<2>(src, S, n_streams, dst, dw, dt, dh),
                  run<3>(src, S, n_streams, dst, dw, dt, dh), run<4>(src, S, n_streams, dst, dw, dt, dh), run<5>(src, S, n_streams, dst, dw, dt, dh), run<6>(src, S, n_streams, dst, dw, dt, dh), run<7>(src, S, n_streams, dst, dw, dt, dh), run<8>(src, S, n_streams, dst, dw, dt, dh), run<9>(src, S, n_streams, dst, dw, dt, dh), run<10>(src, S, n_streams, dst, dw, dt, dh), run<11>(src, S, n_streams, dst, dw, dt, dh), run<12>(src, S, n_streams, dst, dw, dt, dh)};
    for (int i = 0; i < 13; i++) printf("| %s | %.3f |\n", names[i], r[i]);
    printf("| full, loop order window(f) -> stores(f-1) -> loads(f+1) -> compute(f), 4 waves per SIMD | %.3f |\n", run_reordered(src, S, n_streams, dst, dw, dt, dh));
    return 0;
}
