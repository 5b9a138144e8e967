// Micro-benchmark (not part of the product): vector-instruction issue interval on gfx950 as a function of the waves
// resident per SIMD, measured in SHADER CYCLES with s_memtime (no clock assumption), plus the clock the chip holds
// while the stream runs (delta s_memtime / delta s_memrealtime x 100 MHz).
//
// Every wave runs the same stream: kIters x 64 instructions of ONE opcode on 16 independent registers (each register
// is reused every 16 instructions, far beyond the dependent-issue latency), bracketed by two s_memtime stamps.
// Waves per SIMD are set by the workgroup size and a dynamic-LDS request that admits exactly one or two workgroups
// per CU; the placement is verified from HW_REG_HW_ID (distinct (xcc, se, cu, simd) tuples and waves per tuple).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue > profiles/rNN_valu_issue.md
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, real; unsigned hw_id, xcc_id; };

enum Op { FMA, MUL, ADD, SIN, RCP, SQRT, CVT_I32, MUL_LO_U32, MAD_U64_U32, MAD_I64_I32, AND_B32, CNDMASK, PK_FMA, FMA_DEP, FMA_SIN_9_1, DS_READ_B64, DS_WRITE_B64, DS_READ2_B64, DS_READ2ST64_B64, DS_WRITE2_B64, DS_WRITE2ST64_B64, DS_READ_B128, DS_WRITE_B128, MUL_SGPR, MUL_LIT, CMP_CNDMASK, CNDMASK_S64, MAX_F32, MOV_B32, FMAC_F32, FMAAK_F32, DIV_SCALE, DIV_FMAS, DIV_FIXUP, ADD_2SRC, FMA_3SRC, ADD_2SRC_LONG, FMA_3SRC_LONG, FMAC_LONG, MIX_FFT, ADD15_DSW1, ADD15_DSR1, ADD14_DSW1_DSR1, ADD15_GLD1, ADD15_GST1, ADD_DPP_QUAD, ADD_DPP_ROR8, MOV_DPP_ROR8_BANK, PERMLANE32_SWAP, PERMLANE16_SWAP, N_OPS };
static const char* kOpName[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sin_f32", "v_rcp_f32", "v_sqrt_f32", "v_cvt_i32_f32", "v_mul_lo_u32",
                                     "v_mad_u64_u32", "v_mad_i64_i32", "v_and_b32", "v_cndmask_b32", "v_pk_fma_f32 (2 lanes-ops each)",
                                     "v_fma_f32, ONE dependent chain", "15 v_fma_f32 : 1 v_sin_f32", "ds_read_b64 (no wait inside)", "ds_write_b64 (no wait inside)",
                                     "ds_read2_b64 offset1:1 (16 B/lane)", "ds_read2st64_b64 offset1:1 (16 B/lane)", "ds_write2_b64 offset1:9 (16 B/lane)",
                                     "ds_write2st64_b64 offset1:1 (16 B/lane)", "ds_read_b128 (16 B/lane)", "ds_write_b128 (16 B/lane)",
                                     "v_mul_f32 with an SGPR operand", "v_mul_f32 with a 32-bit literal", "v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (2 instr)",
                                     "v_cndmask_b32_e64 with an SGPR-pair mask", "v_max_f32", "v_mov_b32", "v_fmac_f32 (VOP2)", "v_fmaak_f32 (literal)",
                                     "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
                                     "v_add_f32 d, s0, s1: three different VGPRs, rotating", "v_fma_f32 d, s0, s1, s2: four different VGPRs, rotating",
                                     "same v_add_f32 stream, 4096-instruction loop body (16 KiB of code)", "same v_fma_f32 stream, 4096-instruction loop body (32 KiB)",
                                     "v_fmac_f32 d += s0*s1 (VOP2), rotating, 4096-instruction body", "add/sub/mul/fma mix of a radix-8 butterfly, rotating registers",
                                     "15 v_add_f32 : 1 ds_write_b64 (per 16 instructions)", "15 v_add_f32 : 1 ds_read_b64", "14 v_add_f32 : 1 ds_write_b64 : 1 ds_read_b64",
                                     "15 v_add_f32 : 1 global_load_dwordx2 (L2-resident, 512 B per wave)", "15 v_add_f32 : 1 global_store_dwordx2",
                                     "v_add_f32_dpp quad_perm:[1,0,3,2] (rotating registers)", "v_add_f32_dpp row_ror:8", "v_mov_b32_dpp row_ror:8 bank_mask:0x3",
                                     "v_permlane32_swap_b32", "v_permlane16_swap_b32"};

template <int OP>
__device__ __forceinline__ void block16(float (&a)[16], float2 (&p)[8], unsigned long long (&w)[8], f32x4 (&q4)[4], float b, float c, unsigned lds_addr,
                                        unsigned lds_addr16, unsigned long long mask, float* gptr)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned& u = reinterpret_cast<unsigned&>(a[i]);
        if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        if (OP == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(b));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(b));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(b) : );
        if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i & 7]) : "v"(b), "v"(c) : "vcc");
        if (OP == MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i & 7]) : "v"(b), "v"(c) : "vcc");
        if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(p[(i + 4) & 7]));
        if (OP == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        if (OP == FMA_SIN_9_1) {
            if (i == 7) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        if (OP == DS_READ_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i & 7]) : "v"(lds_addr));
        if (OP == DS_WRITE_B64) asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr), "v"(p[i & 7]));
        // two 8-byte accesses per lane: adjacent 512-B rows (lane-contiguous within a row, conflict-free)
        if (OP == DS_READ2_B64) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(q4[i & 3]) : "v"(lds_addr));
        if (OP == DS_READ2ST64_B64) asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1" : "=v"(q4[i & 3]) : "v"(lds_addr));
        if (OP == DS_WRITE2_B64) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:72" :: "v"(lds_addr), "v"(p[i & 7]), "v"(p[(i + 1) & 7]));
        if (OP == DS_WRITE2ST64_B64) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"(lds_addr), "v"(p[i & 7]), "v"(p[(i + 1) & 7]));
        if (OP == MUL_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b));
        if (OP == MUL_LIT) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(a[i]));
        if (OP == CMP_CNDMASK) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        if (OP == CNDMASK_S64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u) : "v"(b), "s"(mask));
        if (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
        if (OP == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == FMAAK_F32) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(a[i]) : "v"(b));
        if (OP == DIV_SCALE) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
        if (OP == DIV_FMAS) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
        if (OP == DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == ADD_2SRC || OP == ADD_2SRC_LONG) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
        if (OP == FMA_3SRC || OP == FMA_3SRC_LONG) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]), "v"(a[(i + 13) & 15]));
        if (OP == FMAC_LONG) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
        if (OP == MIX_FFT) {
            if ((i & 3) == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
            if ((i & 3) == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
            if ((i & 3) == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
            if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]), "v"(a[(i + 13) & 15]));
        }
        if (OP >= ADD15_DSW1 && OP <= ADD15_GST1) {
            const bool w_slot = (i == 7), r_slot = (OP == ADD14_DSW1_DSR1) ? (i == 15) : (i == 7);
            if (w_slot && (OP == ADD15_DSW1 || OP == ADD14_DSW1_DSR1)) asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr), "v"(p[0]));
            else if (r_slot && (OP == ADD15_DSR1 || OP == ADD14_DSW1_DSR1)) asm volatile("ds_read_b64 %0, %1" : "=v"(p[1]) : "v"(lds_addr));
            else if (w_slot && OP == ADD15_GLD1) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(p[2]) : "v"(lds_addr16 & 0x3f8u), "s"(gptr));
            else if (w_slot && OP == ADD15_GST1) asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(lds_addr16 & 0x3f8u), "v"(p[3]), "s"(gptr));
            else asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
        }
        if (OP == ADD_DPP_QUAD) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
        if (OP == ADD_DPP_ROR8) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
        if (OP == MOV_DPP_ROR8_BANK) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
        if (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
        if (OP == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
        if (OP == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[i & 3]) : "v"(lds_addr16));
        if (OP == DS_WRITE_B128) asm volatile("ds_write_b128 %0, %1" :: "v"(lds_addr16), "v"(q4[i & 3]));
    }
}

template <int OP>
__global__ __launch_bounds__(1024) void stream_kernel(Stamp* out, float* sink, int iters, float b, float c)
{
    extern __shared__ unsigned char smem[];
    float a[16];
    float2 p[8];
    unsigned long long w[8];
    f32x4 q4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q4[i] = f32x4{1.0f, 2.0f, 3.0f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; i++) { p[i] = float2{a[i], a[i + 8]}; w[i] = threadIdx.x * 977u + i; }
    const unsigned lds_addr = (unsigned)(size_t)smem + 8u * (threadIdx.x & 63) + 2048u * (threadIdx.x >> 6);    // conflict-free 8-byte slots, 2 KiB per wave
    const unsigned lds_addr16 = (unsigned)(size_t)smem + 16u * (threadIdx.x & 63) + 2048u * (threadIdx.x >> 6);
    float* gptr = sink + (size_t)(blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 1024;   // 4 KiB per wave
    const unsigned long long mask = __builtin_amdgcn_readfirstlane(iters) * 0x5555555555555555ull;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    constexpr bool kLong = (OP == ADD_2SRC_LONG || OP == FMA_3SRC_LONG || OP == FMAC_LONG);
    const int n_it = kLong ? iters / 64 : iters;        // the long body holds 64 x 64 instructions
#pragma unroll 1
    for (int it = 0; it < n_it; it++) {
#pragma unroll
        for (int u = 0; u < (kLong ? 256 : 4); u++) block16<OP>(a, p, w, q4, b, c, lds_addr, lds_addr16, mask, gptr);
        if ((OP >= DS_READ_B64 && OP <= DS_WRITE_B128) || OP >= ADD15_DSW1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y + (float)w[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += q4[i].x + q4[i].y + q4[i].z + q4[i].w;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        Stamp st;
        st.cyc = c1 - c0;
        st.real = r1 - r0;
        st.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
        st.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = st;
    }
}

struct Cfg { int waves_per_simd, threads, blocks_per_cu; };

template <int OP>
static void run(int n_cu, const Cfg& cfg, int iters, Stamp* d_out, float* d_sink, std::vector<Stamp>& h)
{
    const size_t lds = cfg.blocks_per_cu == 1 ? 96 * 1024 : 64 * 1024;      // admits exactly 1 or 2 workgroups per CU (160 KiB)
    CK(hipFuncSetAttribute((const void*)stream_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * cfg.blocks_per_cu;
    for (int rep = 0; rep < 3; rep++) {   // the third launch is the one reported (clock and caches settled)
        hipLaunchKernelGGL(stream_kernel<OP>, dim3(grid), dim3(cfg.threads), lds, 0, d_out, d_sink, iters, 1.0000001f, 1e-9f);
        CK(hipDeviceSynchronize());
    }
    h.resize((size_t)grid * (cfg.threads / 64));
    CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    Stamp* d_out;
    float* d_sink;
    CK(hipMalloc(&d_out, sizeof(Stamp) * n_cu * 2 * 16));
    CK(hipMalloc(&d_sink, sizeof(float) * n_cu * 2 * 16 * 1024 + 65536));
    const Cfg cfgs[] = {{1, 256, 1}, {2, 512, 1}, {3, 768, 1}, {4, 1024, 1}, {6, 768, 2}, {8, 1024, 2}};
    printf("# vector-instruction issue interval vs waves per SIMD (tools/ubench/valu_issue.hip)\n\n");
    printf("device: %s (%s), %d CUs; every wave: %d x 64 instructions of one opcode on 16 independent registers between two s_memtime stamps.\n", prop.name,
           prop.gcnArchName, n_cu, iters);
    printf("`cyc/instr/wave` = median over waves of (delta s_memtime) / instructions; `cyc/instr/SIMD` = that / waves per SIMD = the SIMD's issue\n"
           "interval in shader cycles; `GHz` = median delta s_memtime / delta s_memrealtime x 0.1 (the clock held during the stream);\n"
           "`placement` = distinct (xcc, se, cu, simd) tuples seen x waves on each (min..max), from HW_REG_HW_ID / HW_REG_XCC_ID.\n\n");
    printf("| opcode | waves/SIMD | cyc/instr/wave | cyc/instr/SIMD | GHz | wave-instr/s per SIMD (G) | placement |\n|---|---|---|---|---|---|---|\n");
    std::vector<Stamp> h;
    const int op_first = argc > 2 ? atoi(argv[2]) : 0;
    for (int op = op_first; op < N_OPS; op++) {
        for (const Cfg& cfg : cfgs) {
            switch (op) {
#define CASE(O) case O: run<O>(n_cu, cfg, iters, d_out, d_sink, h); break;
                CASE(FMA) CASE(MUL) CASE(ADD) CASE(SIN) CASE(RCP) CASE(SQRT) CASE(CVT_I32) CASE(MUL_LO_U32) CASE(MAD_U64_U32) CASE(MAD_I64_I32)
                CASE(AND_B32) CASE(CNDMASK) CASE(PK_FMA) CASE(FMA_DEP) CASE(FMA_SIN_9_1) CASE(DS_READ_B64) CASE(DS_WRITE_B64) CASE(DS_READ2_B64) CASE(DS_READ2ST64_B64) CASE(DS_WRITE2_B64)
                CASE(DS_WRITE2ST64_B64) CASE(DS_READ_B128) CASE(DS_WRITE_B128) CASE(MUL_SGPR) CASE(MUL_LIT) CASE(CMP_CNDMASK) CASE(CNDMASK_S64)
                CASE(MAX_F32) CASE(MOV_B32) CASE(FMAC_F32) CASE(FMAAK_F32) CASE(DIV_SCALE) CASE(DIV_FMAS) CASE(DIV_FIXUP)
                CASE(ADD_2SRC) CASE(FMA_3SRC) CASE(ADD_2SRC_LONG) CASE(FMA_3SRC_LONG) CASE(FMAC_LONG) CASE(MIX_FFT)
                CASE(ADD15_DSW1) CASE(ADD15_DSR1) CASE(ADD14_DSW1_DSR1) CASE(ADD15_GLD1) CASE(ADD15_GST1)
                CASE(ADD_DPP_QUAD) CASE(ADD_DPP_ROR8) CASE(MOV_DPP_ROR8_BANK) CASE(PERMLANE32_SWAP) CASE(PERMLANE16_SWAP)
#undef CASE
            }
            std::vector<double> cpi, ghz;
            std::map<unsigned long long, int> place;
            for (const Stamp& s : h) {
                cpi.push_back((double)s.cyc / ((double)iters * 64.0));
                ghz.push_back((double)s.cyc / (double)s.real * 0.1);
                // HW_ID (gfx9 layout): simd [5:4], cu [11:8], sh [12], se [15:13]
                const unsigned long long key = ((unsigned long long)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xff30u);
                place[key]++;
            }
            std::sort(cpi.begin(), cpi.end());
            std::sort(ghz.begin(), ghz.end());
            int mn = 1 << 30, mx = 0;
            for (auto& kv : place) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
            const double c = cpi[cpi.size() / 2], g = ghz[ghz.size() / 2];
            printf("| %s | %d | %.2f | %.2f | %.2f | %.3f | %zu x %d..%d |\n", kOpName[op], cfg.waves_per_simd, c, c / cfg.waves_per_simd, g,
                   g / (c / cfg.waves_per_simd), place.size(), mn, mx);
        }
    }
    return 0;
}
