// ds_write_addtid_b32 semantics on gfx950: which LDS dword does lane l of wave w write for a given M0 and offset?
// (an SALU write of M0 needs one wait state before an add-TID LDS instruction reads it: without the s_nop the first store
// of a wave used the stale M0.)
// build: hipcc --offload-arch=gfx950 -O2 -o addtid_check addtid_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, unsigned base_bytes)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned v = 1000u * (threadIdx.x >> 6) + (threadIdx.x & 63);
    const unsigned m0v = __builtin_amdgcn_readfirstlane(base_bytes + 1024u * (threadIdx.x >> 6));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1\n\tds_write_addtid_b32 %1 offset:256" :: "s"(m0v), "v"(v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main()
{
    unsigned* d; hipMalloc(&d, 4096);
    for (unsigned base : {0u, 8u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, base);
        unsigned h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        printf("base %u:\n", base);
        for (int i = 0; i < 1024; i++) if (h[i] != 0xdeadbeefu && (i % 64 < 4 || i % 64 > 61)) printf("  lds[%d] = %u\n", i, h[i]);
        int n = 0; for (int i = 0; i < 1024; i++) n += h[i] != 0xdeadbeefu; printf("  %d dwords written\n", n);
    }
    return 0;
}
