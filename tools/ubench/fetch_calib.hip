// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of this library (not part of the product).
// MI355X_MICROARCH.md §HBM: the counter reports half the bytes of a 16-byte-per-lane streaming read and is uncalibrated for
// other widths.  Each kernel below reads exactly kBytes (1 GiB, past the 256 MiB Infinity Cache) once:
//   read16_aligned      16 B per lane, 16-byte aligned           (mix / spectrum loads)
//   read8_aligned        8 B per lane, 8-byte loads, 128-byte aligned rows   (vocoder frame loads)
//   read8_shifted        8 B per lane, every row starts 8 bytes off a 128-byte line    (WSOLA copies: sequence starts are arbitrary frames)
//   read8_shifted_trips  the same in the WSOLA copy's shape: a 256-thread workgroup walks its region 4 x 256 frames per trip
//   read4_planar_trips   two planes, 4 B per lane each, rows 4 bytes off a line, the copy's trip shape   (WSOLA on a planar input)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc --output-format csv -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t kBytes = 1ull << 30;

__global__ __launch_bounds__(256) void read16_aligned(const float4* __restrict__ src, float* sink, size_t n)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read8_aligned(const float2* __restrict__ src, float* sink, size_t n)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// one workgroup per region of `per` frames (8 bytes each); region r starts at frame r * stride + 1 (8 bytes into a line)
__global__ __launch_bounds__(256) void read8_shifted(const float2* __restrict__ src, float* sink, int per, size_t stride)
{
    const float2* p = src + (size_t)blockIdx.x * stride + 1;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < per; i += 256) {
        const float2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read8_shifted_trips(const float2* __restrict__ src, float* sink, int per, size_t stride)
{
    const float2* p = src + (size_t)blockIdx.x * stride + 1;
    float acc = 0.0f;
    for (int i0 = 0; i0 < per; i0 += 4 * 256) {
        float2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * 256 + threadIdx.x;
            v[j] = i < per ? p[i] : float2{0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += v[j].x + v[j].y;
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read4_planar_trips(const float* __restrict__ src, float* sink, int per, size_t stride, size_t plane)
{
    const float* p = src + (size_t)blockIdx.x * stride + 1;
    float acc = 0.0f;
    for (int i0 = 0; i0 < per; i0 += 4 * 256) {
        float v[4][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * 256 + threadIdx.x;
            v[j][0] = i < per ? p[i] : 0.0f;
            v[j][1] = i < per ? p[plane + i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += v[j][0] + v[j][1];
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    void* src;
    float* sink;
    CK(hipMalloc(&src, kBytes + 4096));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 0, kBytes + 4096));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read16_aligned, dim3(8192), dim3(256), 0, 0, (const float4*)src, sink, kBytes / 16);
        hipLaunchKernelGGL(read8_aligned, dim3(8192), dim3(256), 0, 0, (const float2*)src, sink, kBytes / 8);
        // regions of 3408 frames (the C5 WSOLA copy: body + overlap), stride 3408 + 16 frames so that every start stays 8 bytes into a line
        const int per = 3408;
        const size_t stride = 3408 + 16;
        const unsigned regions = (unsigned)(kBytes / 8 / stride);
        hipLaunchKernelGGL(read8_shifted, dim3(regions), dim3(256), 0, 0, (const float2*)src, sink, per, stride);
        hipLaunchKernelGGL(read8_shifted_trips, dim3(regions), dim3(256), 0, 0, (const float2*)src, sink, per, stride);
        // two planes of half the buffer each, regions of 3408 samples per plane
        const size_t plane = kBytes / 8;
        const unsigned pregions = (unsigned)(plane / stride);
        hipLaunchKernelGGL(read4_planar_trips, dim3(pregions), dim3(256), 0, 0, (const float*)src, sink, per, stride, plane);
        CK(hipDeviceSynchronize());
        if (rep == 0) printf("read4_planar_trips %zu bytes\n", (size_t)pregions * per * 8);
        if (rep == 0) printf("bytes read: read16_aligned %zu, read8_aligned %zu, read8_shifted(_trips) %zu (regions %u x %d frames)\n", kBytes, kBytes, (size_t)regions * per * 8, regions, per);
    }
    return 0;
}
