// Micro-benchmark (not part of the product): where the time of the padded 512-point FFT (stft_device.h: fft512_pad) goes
// as a function of the waves resident per SIMD — vector instructions, LDS traffic, and whether the two overlap.
//   mode 0  fft512_pad as shipped (twiddles read from LDS tables)
//   mode 1  the same with both twiddle sets held in registers (28 VGPRs more)
//   mode 2  vector work only: the butterflies and twiddle products, no LDS access (transposes replaced by register moves)
//   mode 3  LDS traffic only: the two transposes and the table reads, no arithmetic
// Per wave: kIters FFTs between two s_memtime stamps; reported per FFT and per CU (cycles until a CU has finished one
// FFT on each of its waves, divided by the number of waves).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o fftpad_bench fftpad_bench.hip && ./fftpad_bench
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include "xlane_t1.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace nae;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kMode>
__global__ __launch_bounds__(1024) void bench_kernel(const cf* __restrict__ w512, unsigned long long* out, float* sink, int iters, size_t lds_pad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* w64 = reinterpret_cast<cf*>(smem);
    cf* twa = w64 + 64;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cf* scratch = twa + kTwaCf + wave * kPadScratchCf;
    if (threadIdx.x < 64) w64[threadIdx.x] = w512[8 * (threadIdx.x >> 3) * (threadIdx.x & 7)];
    fill_twa(twa, w512, threadIdx.x, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const FftLds L = make_fft_lds(scratch, twa, w64, lane);
    cf v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cf{(float)(lane + j) * 0.001f, (float)(lane - j) * 0.002f};
    cf ta[7], tb[7];
#pragma unroll
    for (int q = 0; q < 7; q++) { ta[q] = L.twa[64 * q]; tb[q] = L.twb[q + 1]; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (kMode == 0) fft512_pad(v, L);
        if (kMode == 4) {
            fft512_pad_a(v, L);
            t1_xlane(v);
            dft8_fwd(v);
#pragma unroll
            for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], lds_ld(L.twb + p));
#pragma unroll
            for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
            wave_lds_sync();
            dft8_fwd(v);
        }
        if (kMode == 5) {
            dft8_fwd(v);
#pragma unroll
            for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], ta[q - 1]);
            t1_xlane(v);
            dft8_fwd(v);
#pragma unroll
            for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], tb[p - 1]);
#pragma unroll
            for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
            wave_lds_sync();
            dft8_fwd(v);
        }
        if (kMode == 1 || kMode == 2 || (kMode == 6 && !(wave & 1))) {
            dft8_fwd(v);
#pragma unroll
            for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], ta[q - 1]);
            if (kMode == 1) {
#pragma unroll
                for (int q = 0; q < 8; q++) lds_st(L.nat + 72 * q, v[q]);
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = lds_ld(L.t1r + 8 * j);
                wave_lds_sync();
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));
            }
            dft8_fwd(v);
#pragma unroll
            for (int p = 1; p < 8; p++) v[p] = cmul_tw(v[p], tb[p - 1]);
            if (kMode == 1) {
#pragma unroll
                for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
                wave_lds_sync();
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));
            }
            dft8_fwd(v);
        }
        if (kMode == 3 || (kMode == 6 && (wave & 1))) {
            cf t[7];
#pragma unroll
            for (int q = 0; q < 7; q++) t[q] = lds_ld(L.twa + 64 * q);
#pragma unroll
            for (int q = 0; q < 7; q++) v[q + 1].x += t[q].x;
#pragma unroll
            for (int q = 0; q < 8; q++) lds_st(L.nat + 72 * q, v[q]);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lds_ld(L.t1r + 8 * j);
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < 7; q++) t[q] = lds_ld(L.twb + q + 1);
#pragma unroll
            for (int q = 0; q < 7; q++) v[q + 1].y += t[q].y;
#pragma unroll
            for (int p = 0; p < 8; p++) lds_st(L.t2w + 8 * p, v[p]);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lds_ld(L.nat + 66 * j);
            wave_lds_sync();
        }
        // keep magnitudes bounded so no instruction runs on infinities / denormals
#pragma unroll
        for (int j = 0; j < 8; j++) { v[j].x *= 0.0441941738f; v[j].y *= 0.0441941738f; }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) r += v[j].x + v[j].y;
#pragma unroll
    for (int q = 0; q < 7; q++) r += ta[q].x + tb[q].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = c1 - c0;
}

struct Cfg { int waves_per_simd, threads, blocks_per_cu; };

template <int kMode>
static double run(int n_cu, const Cfg& cfg, int iters, const cf* d_w, unsigned long long* d_out, float* d_sink)
{
    const int waves = cfg.threads / 64;
    const size_t need = (64 + kTwaCf + (size_t)waves * kPadScratchCf) * sizeof(cf);
    // pad the request so that exactly blocks_per_cu workgroups fit a CU (160 KiB)
    size_t lds = cfg.blocks_per_cu == 1 ? 96 * 1024 : 64 * 1024;
    if (lds < need) lds = need;
    CK(hipFuncSetAttribute((const void*)bench_kernel<kMode>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * cfg.blocks_per_cu;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(bench_kernel<kMode>, dim3(grid), dim3(cfg.threads), lds, 0, d_w, d_out, d_sink, iters, lds);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)grid * waves);
    CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    if (kMode == 6) printf("| (mode 6 detail: quartiles of cycles per iteration per wave) | %d | %.0f / %.0f / %.0f | |\n", cfg.waves_per_simd, (double)h[h.size() / 4] / iters, (double)h[h.size() / 2] / iters, (double)h[3 * h.size() / 4] / iters);
    return (double)h[h.size() / 2] / iters;          // cycles per FFT per wave
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<cf> w(512);
    for (int k = 0; k < 512; k++) w[k] = cf{(float)cos(6.283185307179586 * k / 512.0), (float)(-sin(6.283185307179586 * k / 512.0))};
    cf* d_w;
    unsigned long long* d_out;
    float* d_sink;
    CK(hipMalloc(&d_w, 512 * sizeof(cf)));
    CK(hipMemcpy(d_w, w.data(), 512 * sizeof(cf), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, sizeof(unsigned long long) * n_cu * 2 * 16));
    CK(hipMalloc(&d_sink, sizeof(float) * n_cu * 2 * 1024));
    const Cfg cfgs[] = {{1, 256, 1}, {2, 512, 1}, {3, 768, 1}, {4, 1024, 1}, {6, 768, 2}, {8, 1024, 2}};
    const char* names[7] = {"fft512_pad, both transposes through LDS (twiddles from LDS)", "same, twiddles in registers", "vector work only (no LDS access)", "LDS traffic only (no arithmetic)", "transpose 1 across lanes (permlane / DPP), twiddles from LDS", "transpose 1 across lanes, twiddles in registers", "even waves: vector work only; odd waves: LDS traffic only (median over ALL waves)"};
    printf("# padded 512-point FFT: cycles per FFT per wave / per CU-FFT-slot, by waves per SIMD (tools/ubench/fftpad_bench.hip)\n\n");
    printf("per FFT: ~250 vector instructions, 16 ds_write_b64, 16 ds_read_b64 (+ 14 table reads in mode 0 and 3)\n\n");
    printf("| mode | waves/SIMD | cycles per FFT per wave | cycles per FFT per CU (= per wave / waves per CU) |\n|---|---|---|---|\n");
    for (int mode = 0; mode < 7; mode++)
        for (const Cfg& cfg : cfgs) {
            double c = 0;
            if (mode == 0) c = run<0>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 1) c = run<1>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 2) c = run<2>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 3) c = run<3>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 4) c = run<4>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 5) c = run<5>(n_cu, cfg, iters, d_w, d_out, d_sink);
            if (mode == 6) c = run<6>(n_cu, cfg, iters, d_w, d_out, d_sink);
            printf("| %s | %d | %.0f | %.1f |\n", names[mode], cfg.waves_per_simd, c, c / (4.0 * cfg.waves_per_simd));
        }
    return 0;
}
