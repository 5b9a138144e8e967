// Micro-benchmark (not part of the product): the 512-point FFT of stft_device.h with plain f32 instructions (-DNAE_PK=0) against
// hand-placed packed VOP3P ones (-DNAE_PK=1) at FULL-CHIP load, with the shader clock held and the board power beside the time.
// Build both and run one after the other on the same box (tools/experiments/r06_pk.sh):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -DNAE_PK=0 -o fftpk_bench_s fftpk_bench.hip
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -DNAE_PK=1 -o fftpk_bench_p fftpk_bench.hip
// Shapes: 768-thread workgroups, two per CU (6 waves per SIMD; the vector rate does not change beyond 2: profiles/r05_valu_wallclock.md); every wave runs `iters` frames of
//   mode 0  "alu":  three register passes (dft8 + twiddles) without the LDS transposes — the arithmetic alone
//   mode 1  "fft":  window, fft512_pad (passes + both padded LDS transposes + LDS twiddle reads), natural-order store — what R1 / R3 do per frame
// The launch repeats for `seconds` of wall clock so that the power management settles; reported: ms per launch (median of the second half),
// clock (s_memtime / s_memrealtime of the last launch), rocm-smi average power over the second half, and a checksum of one small launch
// (equal checksums of the two builds = equal bits).
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
using namespace nae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cyc0, cyc1, real0, real1; };

template <int kMode>
__global__ __launch_bounds__(768) void fft_kernel(const cf* __restrict__ w512, const float* __restrict__ hanng, const cf* __restrict__ seed,
                                                      cf* __restrict__ out, Stamp* stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* w64 = reinterpret_cast<cf*>(smem + 4096);
    cf* twa = w64 + 64;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cf* scratch = twa + kTwaCf + wave * kPadScratchCf;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) hann[i] = hanng[i];
    if (threadIdx.x < 64) w64[threadIdx.x] = w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, w512, threadIdx.x, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const FftLds L = make_fft_lds(scratch, twa, w64, lane);
    // every iteration starts from the same frame (values of a real signal's scale: the power drawn depends on the bits that toggle) and folds its
    // result into an integer checksum
    cf v0[8], v[8];
    cf* frame = reinterpret_cast<cf*>(smem + 4096 + (64 + kTwaCf) * 8 + 12 * kPadScratchCf * 8) + (wave & 1) * 512;
#pragma unroll
    for (int j = 0; j < 8; j++) v0[j] = seed[(blockIdx.x * 12 + wave) % 64 * 512 + lane + 64 * j];
    if (wave < 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) frame[lane + 64 * j] = v0[j];
    }
    __syncthreads();
    uint32_t acc = 0;
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (kMode == 0) {
            cf w[7];
#pragma unroll
            for (int q = 0; q < 7; q++) w[q] = cf{1.0f - 0.001f * q, 0.0447f * q};     // (|w| ~ 1)
#pragma unroll
            for (int j = 0; j < 8; j++) { v[j] = v0[j]; asm volatile("" : "+v"(v[j].x), "+v"(v[j].y)); }
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                dft8_fwd(v);
                if (pass < 2) {
#pragma unroll
                    for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], w[q - 1]);
                }
            }
        } else {
            cf w[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { v[r] = lds_ld(frame + lane + 64 * r); w[r] = lds_ld(reinterpret_cast<const cf*>(hann) + lane + 64 * r); }
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = cf{v[r].x * w[r].x, v[r].y * w[r].y};
            fft512_pad(v, L);
#pragma unroll
            for (int r = 0; r < 8; r++) lds_st(L.nat + 64 * r, v[r]);
            wave_lds_sync();
        }
#pragma unroll
        for (int r = 0; r < 8; r++) acc = (acc ^ __float_as_uint(v[r].x)) + __float_as_uint(v[r].y);
    }
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{c0, __builtin_readcyclecounter(), r0, __builtin_amdgcn_s_memrealtime()};
    if (blockIdx.x < 4) {
#pragma unroll
        for (int r = 0; r < 8; r++) out[(blockIdx.x * 12 + wave) * 512 + lane + 64 * r] = r == 0 ? cf{__uint_as_float(acc), v[0].y} : v[r];
    }
}

static std::atomic<bool> g_stop{false};
static std::vector<std::pair<double, double>> g_power;          // (seconds since start, watts)
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void watch(double t0)
{
    while (!g_stop) {
        FILE* f = popen("rocm-smi -d 0 --showpower --json 2>/dev/null", "r");
        if (!f) return;
        char buf[4096]; std::string s;
        while (fgets(buf, sizeof buf, f)) s += buf;
        pclose(f);
        const size_t k = s.find("Power (W)");
        if (k != std::string::npos) {
            const size_t c = s.find(':', k), q = s.find('"', c);
            if (q != std::string::npos) g_power.push_back({now_s() - t0, atof(s.c_str() + q + 1)});
        }
    }
}

template <int kMode>
static void run(const char* name, const cf* dw, const float* dh, const cf* dseed, cf* dout, Stamp* dst, int n_cu, double seconds, int iters)
{
    const size_t lds = 4096 + (64 + kTwaCf) * 8 + 12 * kPadScratchCf * 8 + 2 * 512 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_kernel<kMode>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 2 * n_cu;
    // checksum launch
    hipLaunchKernelGGL((fft_kernel<kMode>), dim3(grid), dim3(768), lds, 0, dw, dh, dseed, dout, dst, 3);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(4 * 12 * 512 * 2);
    CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 1469598103934665603ull;
    for (uint32_t x : h) sum = (sum ^ x) * 1099511628211ull;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    g_power.clear(); g_stop = false;
    const double t0 = now_s();
    std::thread th(watch, t0);
    std::vector<std::pair<double, double>> ms_at;
    while (now_s() - t0 < seconds) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fft_kernel<kMode>), dim3(grid), dim3(768), lds, 0, dw, dh, dseed, dout, dst, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms_at.push_back({now_s() - t0, ms});
    }
    g_stop = true; th.join();
    std::vector<Stamp> st(grid);
    CK(hipMemcpy(st.data(), dst, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
    std::vector<double> clk, late;
    for (const Stamp& s : st) if (s.real1 > s.real0) clk.push_back((double)(s.cyc1 - s.cyc0) / (double)(s.real1 - s.real0) * 0.1);
    std::sort(clk.begin(), clk.end());
    for (auto& m : ms_at) if (m.first > seconds / 2) late.push_back(m.second);
    std::sort(late.begin(), late.end());
    double pw = 0; int np = 0;
    for (auto& p : g_power) if (p.first > seconds / 2) { pw += p.second; np++; }
    const double ms = late.empty() ? 0 : late[late.size() / 2];
    const double ghz = clk.empty() ? 0 : clk[clk.size() / 2];
    const double frames = (double)grid * 12 * iters;
    printf("| %s | %s | %.3f | %.2f | %.0f | %.0f | %.1f | %.3f | %016llx |\n", NAE_PK ? "packed" : "plain", name, ms, ghz, np ? pw / np : 0.0,
           ms * 1e-3 * ghz * 1e9 * n_cu / frames, frames / (ms * 1e-3) / 1e9, np ? (pw / np) * ms * 1e-3 / frames * 1e9 : 0.0, sum);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<cf> w512(512), seed(64 * 512);
    std::vector<float> hann(1024);
    const double two_pi = 6.283185307179586;
    for (int k = 0; k < 512; k++) w512[k] = cf{(float)cos(two_pi * k / 512), (float)-sin(two_pi * k / 512)};
    for (int n = 0; n < 1024; n++) hann[n] = (float)(0.5 - 0.5 * cos(two_pi * n / 1024));
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (auto& s : seed) {
        float f[2];
        for (int i = 0; i < 2; i++) {
            x += 0x9E3779B97F4A7C15ull;
            unsigned long long z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            f[i] = (float)((unsigned)(z >> 40)) * (1.0f / 8388608.0f) - 1.0f;
        }
        s = cf{f[0], f[1]};
    }
    // a few exact zeros, signed zeros and tiny values: the two builds must agree there too
    seed[3] = cf{0.0f, -0.0f}; seed[70] = cf{-0.0f, 1e-41f}; seed[600] = cf{1e-39f, -1e-38f}; seed[1030] = cf{0.0f, 0.0f};
    cf *dw, *dseed, *dout; float* dh; Stamp* dst;
    CK(hipMalloc(&dw, 512 * 8)); CK(hipMalloc(&dh, 4096)); CK(hipMalloc(&dseed, seed.size() * 8)); CK(hipMalloc(&dout, 4 * 16 * 512 * 8));
    CK(hipMalloc(&dst, sizeof(Stamp) * 2 * n_cu));
    CK(hipMemcpy(dw, w512.data(), 512 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, hann.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemcpy(dseed, seed.data(), seed.size() * 8, hipMemcpyHostToDevice));
    if (argc > 2 && !strcmp(argv[2], "head"))
        printf("| build | loop | ms per launch | clock GHz | board W | CU-cycles per frame | Gframes/s | nJ per frame | checksum (3 frames) |\n|---|---|---|---|---|---|---|---|---|\n");
    run<0>("alu", dw, dh, dseed, dout, dst, n_cu, seconds, 6000);
    run<1>("fft", dw, dh, dseed, dout, dst, n_cu, seconds, 3000);
    return 0;
}
