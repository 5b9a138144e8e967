// Micro-benchmark (not part of the product): what the memory system delivers for streaming kernels with different read : write ratios —
// the mixes of the graph's two memory-side kernels (mix + transposer reads 5.2 GB and writes 7.3 GB; spectrum reads 4.3 GB and writes
// 8.0 GB).  Plain grid-stride kernels, 16 bytes per lane, buffers far larger than the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip && ./rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// R reads and W writes of n float4 each (R, W in 0..3); kNt: non-temporal stores
template <int R, int W, bool kNt>
__global__ __launch_bounds__(256) void rw_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                 float4* __restrict__ x, float4* __restrict__ y, float4* __restrict__ z, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = float4{1.0f, 2.0f, 3.0f, (float)i};
        if (R > 0) { const float4 t = a[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (R > 1) { const float4 t = b[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (R > 2) { const float4 t = c[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (W == 0) { if (v.x == 12345.678f) x[i] = v; }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        auto st = [&](float4* p) {
            if (kNt) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p) + i);
            else p[i] = v;
        };
        if (W > 0) st(x);
        if (W > 1) st(y);
        if (W > 2) st(z);
    }
}

static int g_grid = 256 * 16;
template <int R, int W, bool kNt>
static void run(const char* what, float4** buf, size_t n)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = g_grid;
    for (int rep = 0; rep < 3; rep++) rw_kernel<R, W, kNt><<<grid, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], n);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; rep++) rw_kernel<R, W, kNt><<<grid, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)(R + W) * n * 16;
    printf("| %-34s | %d : %d | %6.2f GB | %7.3f ms | %6.2f TB/s |\n", what, R, W, bytes / 1e9, ms, bytes / ms / 1e9);
}

int main(int argc, char** argv)
{
    if (argc > 1) g_grid = atoi(argv[1]);          // workgroups of 256 threads (default 4096 = 16 per CU)
    printf("grid %d workgroups\n", g_grid);
    const size_t n = (size_t)1 << 28 >> 2;          // 2^26 float4 = 1 GiB per buffer
    float4* buf[6];
    for (auto& p : buf) { CK(hipMalloc(&p, n * 16)); CK(hipMemset(p, 0, n * 16)); }
    CK(hipDeviceSynchronize());
    printf("| kernel | reads : writes | bytes per launch | time | rate |\n|---|---|---|---|---|\n");
    run<1, 0, false>("read only", buf, n);
    run<3, 0, false>("read only, three streams", buf, n);
    run<0, 1, false>("write only", buf, n);
    run<0, 3, false>("write only, three streams", buf, n);
    run<0, 3, true>("write only, three streams, nt", buf, n);
    run<1, 1, false>("copy", buf, n);
    run<1, 1, true>("copy, nt stores", buf, n);
    run<2, 1, false>("2 reads : 1 write", buf, n);
    run<1, 2, false>("1 read : 2 writes (spectrum's mix)", buf, n);
    run<1, 2, true>("1 read : 2 writes, nt stores", buf, n);
    run<2, 3, false>("2 reads : 3 writes (mix kernel's)", buf, n);
    run<2, 3, true>("2 reads : 3 writes, nt stores", buf, n);
    run<1, 3, false>("1 read : 3 writes", buf, n);
    return 0;
}
