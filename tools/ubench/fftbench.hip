// Micro-benchmark of the wave-level STFT building blocks (not part of the product): cycles per stage per wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fftbench fftbench.hip && ./fftbench
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace nae;

template <int kStage, int kWavesPerBlock, int kOcc>
__global__ __launch_bounds__(kWavesPerBlock * 64, kOcc) void bench_kernel(const cf* w512, const cf* t1024g, const float* hanng,
                                                                         float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + 4096);
    cf* w64 = t1024 + 520;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cf* scratch = reinterpret_cast<cf*>(smem + 4096 + 520 * 8 + 512) + wave * kScratchCf;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) hann[i] = hanng[i];
    for (int i = threadIdx.x; i < 513; i += blockDim.x) t1024[i] = t1024g[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    FftTw tw;
    load_fft_tw(tw, w512, w64, lane);
    cf v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cf{(float)(lane + j) * 0.001f, (float)(lane - j) * 0.002f};
    uint32_t acc = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (kStage >= 1) fft512_fwd(v, scratch, tw, lane);
        cf nyq{0, 0};
        if (kStage >= 2) nyq = rfft_split(v, scratch, t1024, lane);
        if (kStage >= 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc += atan2_q32(v[r].y, v[r].x);
            acc += atan2_q32(nyq.y, nyq.x);
        }
        if (kStage >= 4) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float ph = (float)(int32_t)(acc + r) * (1.0f / 4294967296.0f);
                const float mag = __builtin_amdgcn_sqrtf(__builtin_fmaf(v[r].x, v[r].x, v[r].y * v[r].y));
                v[r] = cf{mag * __builtin_amdgcn_cosf(ph), mag * __builtin_amdgcn_sinf(ph)};
            }
        }
        if (kStage == 10) {   // phase_inc alone (9 bins)
#pragma unroll
            for (int r = 0; r < 9; r++) {
                const unsigned k = (r < 8) ? (unsigned)(lane + 64 * r) : 512u;
                const unsigned d = 215u + (it & 1), R = 19976592u - (it & 1) * 92484u;
                const uint32_t qa = __float_as_uint(v[r & 7].x), qp = __float_as_uint(v[r & 7].y);
                const uint32_t e = ((k * d) & 1023u) << 22;
                const int32_t dw = (int32_t)(qa - qp - e);
                const uint32_t adv = ((k * 256u) & 1023u) << 22;
                const long long scaled = ((long long)dw * (long long)R + (1ll << 23)) >> 24;
                acc += adv + (uint32_t)scaled;
            }
        }
        if (kStage == 11) {   // atan2 alone (9 bins)
#pragma unroll
            for (int r = 0; r < 8; r++) acc += atan2_q32(v[r].y, v[r].x);
            acc += atan2_q32(v[0].x, v[1].y);
        }
        if (kStage == 12) {   // rotate by phase difference (9 bins): cvt, sin, cos, 4 flops
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float ph = (float)(int32_t)(acc + r * 7919u) * (1.0f / 4294967296.0f);
                const float c = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
                v[r] = cf{__builtin_fmaf(v[r].x, c, -(v[r].y * sn)), __builtin_fmaf(v[r].x, sn, v[r].y * c)};
            }
            acc += __float_as_uint(v[3].x);
        }
        if (kStage == 13) {   // c2r pre-twiddle: natural write, natural + mirrored read, table read
#pragma unroll
            for (int r = 0; r < 8; r++) scratch[lane + 64 * r] = v[r];
            if (lane == 0) scratch[512] = v[0];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = lane + 64 * r;
                const cf Xk = scratch[k], Xm = scratch[512 - k];
                const cf T = t1024[k];
                const cf E{0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y)};
                const cf D{0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y)};
                const cf Q{__builtin_fmaf(T.x, D.x, T.y * D.y), __builtin_fmaf(T.x, D.y, -(T.y * D.x))};
                v[r] = cf{E.x - Q.y, -(E.y + Q.x)};
            }
            wave_lds_sync();
        }
        if (kStage == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = cf{v[r].x * 1.0001f + 0.5f, v[r].y * 0.9999f - 0.5f};
        }
        // keep magnitudes bounded
#pragma unroll
        for (int r = 0; r < 8; r++) { v[r].x *= 0.04f; v[r].y *= 0.04f; }
    }
    float s = (float)acc;
#pragma unroll
    for (int r = 0; r < 8; r++) s += v[r].x + v[r].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kStage, int kWavesPerBlock, int kOcc>
static void run(const char* name, const cf* w512, const cf* t1024, const float* hann, float* out, int blocks_per_cu)
{
    const int iters = 2000;
    const size_t lds = 4096 + 520 * 8 + 512 + kWavesPerBlock * kScratchCf * 8;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((bench_kernel<kStage, kWavesPerBlock, kOcc>), dim3(grid), dim3(kWavesPerBlock * 64), lds, 0, w512, t1024, hann, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((bench_kernel<kStage, kWavesPerBlock, kOcc>), dim3(grid), dim3(kWavesPerBlock * 64), lds, 0, w512, t1024, hann, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double waves_per_cu = (double)kWavesPerBlock * blocks_per_cu;
    const double us_per_iter_per_wave = ms * 1e3 / iters;          // every wave runs `iters` iterations concurrently
    const double iters_per_s_chip = 256.0 * waves_per_cu * iters / (ms * 1e-3);
    printf("%-28s waves/CU %4.0f  %8.3f ms  %7.3f us/iter/wave  (%6.0f cyc @2.4GHz)  chip %.3e iter/s  per-CU cycles/iter %.0f\n", name,
           waves_per_cu, ms, us_per_iter_per_wave, us_per_iter_per_wave * 2400, iters_per_s_chip, 2.4e9 * 256 / iters_per_s_chip);
}

int main()
{
    std::vector<cf> w512(512), t1024(520);
    std::vector<float> hann(1024);
    const double two_pi = 6.283185307179586;
    for (int k = 0; k < 512; k++) w512[k] = cf{(float)cos(two_pi * k / 512), (float)-sin(two_pi * k / 512)};
    for (int k = 0; k <= 512; k++) t1024[k] = cf{(float)cos(two_pi * k / 1024), (float)-sin(two_pi * k / 1024)};
    for (int n = 0; n < 1024; n++) hann[n] = (float)(0.5 - 0.5 * cos(two_pi * n / 1024));
    cf *dw, *dt; float *dh, *dout;
    hipMalloc(&dw, 512 * 8); hipMalloc(&dt, 520 * 8); hipMalloc(&dh, 4096); hipMalloc(&dout, 256 * 8 * 512 * 4);
    hipMemcpy(dw, w512.data(), 512 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dt, t1024.data(), 520 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dh, hann.data(), 4096, hipMemcpyHostToDevice);
    run<0, 8, 2>("trivial loop, 8 w/CU", dw, dt, dh, dout, 1);
    run<10, 8, 2>("phase_inc x9, 8 w/CU", dw, dt, dh, dout, 1);
    run<11, 8, 2>("atan2 x9, 8 w/CU", dw, dt, dh, dout, 1);
    run<12, 8, 2>("rotate x8, 8 w/CU", dw, dt, dh, dout, 1);
    run<13, 8, 2>("c2r pretwiddle, 8 w/CU", dw, dt, dh, dout, 1);
    run<1, 8, 2>("fft only, 8 w/CU", dw, dt, dh, dout, 1);
    run<1, 8, 4>("fft only, 16 w/CU", dw, dt, dh, dout, 2);
    run<1, 4, 1>("fft only, 4 w/CU", dw, dt, dh, dout, 1);
    run<2, 8, 2>("fft+split, 8 w/CU", dw, dt, dh, dout, 1);
    run<2, 8, 4>("fft+split, 16 w/CU", dw, dt, dh, dout, 2);
    run<3, 8, 2>("fft+split+atan2, 8 w/CU", dw, dt, dh, dout, 1);
    run<3, 8, 4>("fft+split+atan2, 16 w/CU", dw, dt, dh, dout, 2);
    run<4, 8, 2>("..+mag/sincos, 8 w/CU", dw, dt, dh, dout, 1);
    run<4, 8, 4>("..+mag/sincos, 16 w/CU", dw, dt, dh, dout, 2);
    return 0;
}
