// Micro-benchmark (not part of the product): do the half-rate vector instructions of gfx950 (profiles/r05_valu_ops.md: v_max / v_cmp / v_cndmask / v_cvt / integer
// multiplies ..., 4.1 cycles per wave64 instruction per SIMD) share the issue slots of the full-rate ones (f32 add / mul / fma, 2.15 cycles), or do they run BESIDE them?
// 512-thread workgroups: waves 0-3 (one per SIMD) stream opcode A, waves 4-7 (the same SIMDs) opcode B; 4 workgroups per CU = 8 waves per SIMD, half of them on each
// opcode.  If the two share one issue path the launch takes (tA + tB) / 2 per instruction pair; if they run on separate paths, max(tA, tB) / 2.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_pipes valu_pipes.hip && ./valu_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA, ADD, MAXF, CNDMASK, CVT, MUL_LO, SIN, PK_ADD, PK_MUL, PK_FMA, ADD_DPP, MUL_DPP, MUL_SGPR, ADD_SDWA, FMA_SGPR, N_OPS };
static const char* kName[N_OPS] = {"v_fma_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32", "v_cvt_f32_i32", "v_mul_lo_u32", "v_sin_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32_dpp (identity quad_perm)", "v_mul_f32_dpp (identity quad_perm)", "v_mul_f32 (SGPR source)", "v_add_f32_sdwa (dword selects)", "v_fma_f32 (SGPR source)"};

template <int OP>
__device__ __forceinline__ void stream(float (&a)[16], unsigned (&u)[16], float b, float c, unsigned ub, unsigned long long smask, int iters)
{
    const float sb = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(b)));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "s"(smask));
                if (OP == CVT) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(u[i]));
                if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                if (OP == ADD_DPP) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
                if (OP == MUL_DPP) asm volatile("v_mul_f32_dpp %0, %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (OP == MUL_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                if (OP == ADD_SDWA) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(c));
                if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
                if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<float2*>(&a[2 * (i & 7)])) : "v"(*reinterpret_cast<float2*>(&a[2 * ((i + 4) & 7)])));
                if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<float2*>(&a[2 * (i & 7)])) : "v"(*reinterpret_cast<float2*>(&a[2 * ((i + 4) & 7)])));
                if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<float2*>(&a[2 * (i & 7)])) : "v"(*reinterpret_cast<float2*>(&a[2 * ((i + 4) & 7)])));
            }
        }
    }
}

template <int A, int B, int MULA = 1>
__global__ __launch_bounds__(512) void pair_kernel(float* sink, int iters, float b, float c)
{
    extern __shared__ float dyn_lds[];
    float a[16];
    unsigned u[16];
    if (iters < 0) dyn_lds[threadIdx.x] = b;
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = b * (float)(i + 1) + (float)threadIdx.x * 1e-3f; u[i] = (unsigned)(i * 2654435761u) ^ threadIdx.x; }
    const unsigned ub = __float_as_uint(b);
    const unsigned long long smask = __builtin_amdgcn_readfirstlane((int)ub) | 0x5555555500000000ull;
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) == 0) stream<A>(a, u, b, c, ub, smask, iters * MULA);
    else stream<B>(a, u, b, c, ub, smask, iters);
    float acc = 0;
    unsigned xs = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc += a[i]; xs += u[i]; }
    if (acc == 12345.678f || xs == 0x12345678u) sink[threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, int, float, float);
struct Case { const char* a; const char* b; kern_t k; };

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* sink; CK(hipMalloc(&sink, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Case cases[] = {
        {kName[FMA], kName[FMA], pair_kernel<FMA, FMA>}, {kName[ADD], kName[ADD], pair_kernel<ADD, ADD>}, {kName[MAXF], kName[MAXF], pair_kernel<MAXF, MAXF>},
        {kName[CNDMASK], kName[CNDMASK], pair_kernel<CNDMASK, CNDMASK>}, {kName[CVT], kName[CVT], pair_kernel<CVT, CVT>},
        {kName[MUL_LO], kName[MUL_LO], pair_kernel<MUL_LO, MUL_LO>}, {kName[SIN], kName[SIN], pair_kernel<SIN, SIN>},
        {kName[ADD], kName[MAXF], pair_kernel<ADD, MAXF>}, {kName[ADD], kName[CNDMASK], pair_kernel<ADD, CNDMASK>}, {kName[ADD], kName[CVT], pair_kernel<ADD, CVT>},
        {kName[ADD], kName[MUL_LO], pair_kernel<ADD, MUL_LO>}, {kName[ADD], kName[SIN], pair_kernel<ADD, SIN>}, {kName[FMA], kName[MAXF], pair_kernel<FMA, MAXF>},
        {kName[FMA], kName[SIN], pair_kernel<FMA, SIN>}, {kName[MAXF], kName[SIN], pair_kernel<MAXF, SIN>}, {kName[MAXF], kName[MUL_LO], pair_kernel<MAXF, MUL_LO>},
        {kName[PK_ADD], kName[PK_ADD], pair_kernel<PK_ADD, PK_ADD>}, {kName[PK_MUL], kName[PK_MUL], pair_kernel<PK_MUL, PK_MUL>}, {kName[PK_FMA], kName[PK_FMA], pair_kernel<PK_FMA, PK_FMA>},
        {kName[ADD], kName[PK_ADD], pair_kernel<ADD, PK_ADD>}, {kName[ADD], kName[PK_MUL], pair_kernel<ADD, PK_MUL>}, {kName[FMA], kName[PK_FMA], pair_kernel<FMA, PK_FMA>},
        {kName[MAXF], kName[PK_ADD], pair_kernel<MAXF, PK_ADD>},
        {kName[ADD_DPP], kName[ADD_DPP], pair_kernel<ADD_DPP, ADD_DPP>}, {kName[MUL_DPP], kName[MUL_DPP], pair_kernel<MUL_DPP, MUL_DPP>}, {kName[MUL_SGPR], kName[MUL_SGPR], pair_kernel<MUL_SGPR, MUL_SGPR>},
        {kName[ADD_SDWA], kName[ADD_SDWA], pair_kernel<ADD_SDWA, ADD_SDWA>}, {kName[FMA_SGPR], kName[FMA_SGPR], pair_kernel<FMA_SGPR, FMA_SGPR>},
        {kName[ADD], kName[ADD_DPP], pair_kernel<ADD, ADD_DPP>}, {kName[ADD], kName[MUL_DPP], pair_kernel<ADD, MUL_DPP>}, {kName[ADD], kName[MUL_SGPR], pair_kernel<ADD, MUL_SGPR>},
        {kName[ADD], kName[ADD_SDWA], pair_kernel<ADD, ADD_SDWA>},
        {"2 x v_add_f32", kName[ADD_DPP], pair_kernel<ADD, ADD_DPP, 2>}, {"2 x v_add_f32", kName[MUL_SGPR], pair_kernel<ADD, MUL_SGPR, 2>}, {"2 x v_add_f32", kName[MAXF], pair_kernel<ADD, MAXF, 2>},
        {"2 x v_fma_f32", kName[FMA_SGPR], pair_kernel<FMA, FMA_SGPR, 2>}, {kName[ADD], kName[FMA_SGPR], pair_kernel<ADD, FMA_SGPR>}, {kName[MAXF], kName[ADD_DPP], pair_kernel<MAXF, ADD_DPP>},
    };
    printf("# do half-rate and full-rate vector instructions share an issue path? (`tools/ubench/valu_pipes.hip`), %s, %d CUs\n\n", prop.gcnArchName, n_cu);
    printf("512-thread workgroups, 4 per CU (8 waves per SIMD): waves 0-3 of a workgroup stream opcode A, waves 4-7 opcode B (every SIMD holds four waves of each);\n"
           "each wave runs 512 x 256 instructions.  ms = hipEvent time of the launch (median of 4).  `alone` = the launch with BOTH halves on that opcode, so a half's own\n"
           "work is alone / 2: if the two opcodes take turns on one issue path the mixed launch needs (A alone + B alone) / 2, if they run beside each other max(A, B alone) / 2.\n\n");
    printf("| A | B | ms | A alone | B alone | taking turns: (A + B) / 2 | beside: max(A, B) / 2 |\n|---|---|---|---|---|---|---|\n");
    double alone[N_OPS] = {0};
    const int iters = 512, blocks = n_cu * 4 * 2;
    int idx = 0;
    for (const Case& cs : cases) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(cs.k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        std::vector<double> ms_all;
        for (int l = 0; l < 5; l++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(cs.k, dim3(blocks), dim3(512), 33 * 1024, 0, sink, iters, 1.0000001f, 1e-9f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (l > 0) ms_all.push_back(ms);
        }
        std::sort(ms_all.begin(), ms_all.end());
        const double ms = ms_all[ms_all.size() / 2];
        int ia = -1, ib = -1;
        for (int o = 0; o < N_OPS; o++) { if (cs.a == kName[o]) ia = o; if (cs.b == kName[o]) ib = o; }
        if (ia < 0) { printf("| %s | %s | **%.3f** | (the A half runs twice the instructions: 1.5 x the work of a launch of A alone) | | | |\n", cs.a, cs.b, ms); continue; }
        if (ia == ib) { alone[ia] = ms; printf("| %s | %s | %.3f | | | | |\n", cs.a, cs.b, ms); }
        else printf("| %s | %s | **%.3f** | %.3f | %.3f | %.3f | %.3f |\n", cs.a, cs.b, ms, alone[ia], alone[ib], (alone[ia] + alone[ib]) / 2, std::max(alone[ia], alone[ib]) / 2);
        fflush(stdout);
        idx++;
    }
    return 0;
}
