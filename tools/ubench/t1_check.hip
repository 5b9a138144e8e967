// Check (not part of the product): the cross-lane transpose t1_xlane against the LDS transpose it replaces.
//   hipcc --offload-arch=gfx950 -O3 -o t1_check t1_check.hip && ./t1_check
#include "../../nodey-audio-editor_amd/csrc/stft_device.h"
#include "xlane_t1.h"
#include <cstdio>
using namespace nae;
__global__ void k(float* out)
{
    __shared__ cf scratch[kPadScratchCf];
    const int lane = threadIdx.x;
    cf a[8], b[8];
    for (int q = 0; q < 8; q++) a[q] = b[q] = cf{(float)(1000 * q + lane), (float)(-(1000 * q + lane))};
    // LDS form: u1[q][l] at 72 q + l, read lane (m, qq) register j = u1[qq][m + 8 j]
    for (int q = 0; q < 8; q++) scratch[72 * q + lane] = a[q];
    __syncthreads();
    for (int j = 0; j < 8; j++) a[j] = scratch[72 * (lane >> 3) + (lane & 7) + 8 * j];
    t1_xlane(b);
    for (int j = 0; j < 8; j++) { out[(lane * 8 + j) * 4 + 0] = a[j].x; out[(lane * 8 + j) * 4 + 1] = a[j].y; out[(lane * 8 + j) * 4 + 2] = b[j].x; out[(lane * 8 + j) * 4 + 3] = b[j].y; }
}
int main()
{
    float* d; hipMalloc(&d, 64 * 8 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 8 * 4]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64 * 8; i++) if (h[4 * i] != h[4 * i + 2] || h[4 * i + 1] != h[4 * i + 3]) { if (bad < 12) printf("lane %d reg %d: lds %g xlane %g\n", i / 8, i % 8, h[4 * i], h[4 * i + 2]); bad++; }
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "t1_xlane == LDS transpose", bad);
    return bad != 0;
}
