// Micro-benchmark (not part of the product): one radix-8 layer of the 512-point FFT — 64 DFT-8s, one per lane in the
// product's layout — as butterflies on the vector ALU versus as a 16x16 real matrix product on the f32 MFMA pipe
// (v_mfma_f32_16x16x4_f32: Y[16 x 64] = W[16 x 16] X[16 x 64], 4 column tiles x 4 k-steps = 16 instructions), with
// and without independent vector instructions issued beside it.  Operands are assumed to arrive in MFMA layout (in
// the FFT they would be read that way from the LDS transposes that separate the layers).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o mfma_dft8 mfma_dft8.hip && ./mfma_dft8
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include <cstdio>
#include <vector>
using namespace nae;
typedef float floatx4 __attribute__((ext_vector_type(4)));

// kMode 0: VALU dft8_fwd + 7 twiddle products (what a layer costs today); 1: 16 MFMAs; kFill: independent FMAs per layer
template <int kMode, int kFill>
__global__ __launch_bounds__(512, 2) void layer_kernel(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    cf v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cf{(float)(lane + j) * 0.001f, (float)(lane - j) * 0.002f};
    cf tw[7];
#pragma unroll
    for (int j = 0; j < 7; j++) tw[j] = cf{0.999f - 0.01f * j, 0.01f * j};
    float a[4], b[16];
    floatx4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; s++) a[s] = 0.01f * (lane % 16) - 0.02f * s;
#pragma unroll
    for (int i = 0; i < 16; i++) b[i] = 0.001f * lane + 0.01f * i;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = 1.0f + 0.001f * i + 0.0001f * lane;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (kMode == 0) {
            dft8_fwd(v);
#pragma unroll
            for (int q = 1; q < 8; q++) v[q] = cmul_tw(v[q], tw[q - 1]);
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[4 * t + s], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {   // feed the results back so nothing is hoisted
                b[4 * t + 0] = acc[t].x; b[4 * t + 1] = acc[t].y; b[4 * t + 2] = acc[t].z; b[4 * t + 3] = acc[t].w;
            }
        }
#pragma unroll
        for (int i = 0; i < kFill; i++) f[i & 7] = __builtin_fmaf(f[i & 7], 1.0001f, 0.0001f);
    }
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) r += v[j].x + v[j].y + f[j];
#pragma unroll
    for (int i = 0; i < 16; i++) r += b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int kMode, int kFill>
static void run(const char* what, float* d_out)
{
    const int iters = 2000, blocks = 256;   // one 512-thread workgroup per CU: 2 waves per SIMD, as the vocoder kernel runs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((layer_kernel<kMode, kFill>), dim3(blocks), dim3(512), 0, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((layer_kernel<kMode, kFill>), dim3(blocks), dim3(512), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 2 waves per SIMD: time per layer per SIMD covers 2 wave-layers
    printf("%-44s %8.1f ns per layer and wave pair  (%6.0f cycles at 2.4 GHz)\n", what, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, 256 * 512 * sizeof(float));
    run<0, 0>("VALU butterflies + twiddles", d_out);
    run<0, 96>("VALU butterflies + twiddles + 96 FMA", d_out);
    run<1, 0>("16 MFMA 16x16x4 f32", d_out);
    run<1, 48>("16 MFMA + 48 FMA", d_out);
    run<1, 96>("16 MFMA + 96 FMA", d_out);
    run<1, 144>("16 MFMA + 144 FMA", d_out);
    run<1, 192>("16 MFMA + 192 FMA", d_out);
    return 0;
}
