// Micro-benchmark (not part of the product): wall-clock issue cost of the NON-f32-arithmetic vector instructions the vocoder's phase roles use (transcendentals,
// 32 x 32 -> 64-bit integer products, conversions, selects ...), in the frame of tools/ubench/valu_wallclock.hip: every wave runs iters x 256 instructions of one
// opcode on 16 rotating registers, the launch is timed with hipEvents, cycles per wave-instruction per SIMD = seconds x held clock x SIMDs / wave-instructions.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ops valu_ops.hip && ./valu_ops > profiles/r05_valu_ops.md
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA, SIN, COS, RCP, SQRT, MUL_LO_U32, MUL_HI_U32, MUL_HI_I32, MAD_U64_U32, MAD_I64_I32, LSHL_ADD_U64, ALIGNBIT, CVT_F32_I32, CVT_I32_F32, CNDMASK, ADD3_U32, MOV, AND,
          ADD_U32, MAD_U32_U24, LSHLREV, BFE, CNDMASK_S, XOR, OR, SUB_U32, MAX_F32, CMP_F32, BFI, SUBREV_F32, FMAC, MUL_LEGACY, LSHRREV, ASHRREV, LSHL_ADD_U32, ADD_LSHL_U32, AND_OR, MED3, MAX3, FMAAK, FMAMK, MUL_E64_NEG, ADD_E64_ABS, FMA_NEG, RNDNE, XAD, MIN_F32, MUL_SGPR, ADD_LIT, READLANE, MOV_B64, PERM, N_OPS };
static const char* kName[N_OPS] = {"v_fma_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_sqrt_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32",
                                   "v_lshl_add_u64", "v_alignbit_b32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_cndmask_b32", "v_add3_u32", "v_mov_b32", "v_and_b32", "v_add_u32",
                                   "v_mad_u32_u24", "v_lshlrev_b32", "v_bfe_u32", "v_cndmask_b32 (VOP3, SGPR pair)", "v_xor_b32", "v_or_b32", "v_sub_u32", "v_max_f32", "v_cmp_lt_f32 (writes vcc)", "v_bfi_b32", "v_subrev_f32", "v_fmac_f32", "v_mul_legacy_f32", "v_lshrrev_b32", "v_ashrrev_i32", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_med3_f32", "v_max3_f32", "v_fmaak_f32 (literal)", "v_fmamk_f32 (literal)", "v_mul_f32_e64 (neg modifier)", "v_add_f32_e64 (abs modifier)", "v_fma_f32 (neg modifier)", "v_rndne_f32", "v_xad_u32", "v_min_f32", "v_mul_f32 (SGPR operand)", "v_add_f32 (32-bit literal)", "v_readlane_b32", "v_mov_b64", "v_perm_b32"};

struct Stamp { unsigned long long cyc0, cyc1, real0, real1; };

template <int OP>
__global__ __launch_bounds__(256) void stream_kernel(float* sink, Stamp* stamps, int iters, float b, float c)
{
    extern __shared__ float dyn_lds[];
    float a[16];
    unsigned u[16];
    unsigned long long w[8];
    if (iters < 0) dyn_lds[threadIdx.x] = b;
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = b * (float)(i + 1) + (float)threadIdx.x * 1e-3f; u[i] = (unsigned)(i * 2654435761u) ^ threadIdx.x; }
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = ((unsigned long long)u[2 * i] << 32) | u[2 * i + 1];
    const unsigned ub = __float_as_uint(b), uc = __float_as_uint(c);
    const float sb = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(b)));
    unsigned sl = 0;
    const unsigned long long smask = __builtin_amdgcn_readfirstlane((int)ub) | 0x5555555500000000ull;       // a wave-uniform lane mask in an SGPR pair
    unsigned long long c0 = 0, r0 = 0;
    if ((threadIdx.x & 255) == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                if (OP == COS) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
                if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == MUL_HI_I32) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i & 7]) : "v"(ub), "v"(uc) : "vcc");
                if (OP == MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i & 7]) : "v"(ub), "v"(uc) : "vcc");
                if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i & 7]) : "v"(w[(i + 4) & 7]));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(u[i]) : "v"(ub));
                if (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(u[i]));
                if (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(u[i]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : );
                if (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i + 8) & 15]));
                if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
                if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
                if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
                if (OP == CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "s"(smask));
                if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == CMP_F32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
                if (OP == SUBREV_F32) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
                if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == MUL_LEGACY) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
                if (OP == ASHRREV) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(u[i]));
                if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(ub));
                if (OP == ADD_LSHL_U32) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(u[i]) : "v"(ub));
                if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
                if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == FMAAK) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800347" : "+v"(a[i]) : "v"(b));
                if (OP == FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a[i]) : "v"(c));
                if (OP == MUL_E64_NEG) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == ADD_E64_ABS) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(c));
                if (OP == FMA_NEG) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
                if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
                if (OP == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == MUL_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                if (OP == ADD_LIT) asm volatile("v_add_f32 %0, 0x2f800000, %0" : "+v"(a[i]));
                if (OP == READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sl) : "v"(u[i]));
                if (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "+v"(w[i & 7]) : "v"(w[(i + 4) & 7]));
                if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
            }
        }
    }
    if ((threadIdx.x & 255) == 0) {
        Stamp s{c0, __builtin_readcyclecounter(), r0, __builtin_amdgcn_s_memrealtime()};
        stamps[blockIdx.x] = s;
    }
    float acc = 0;
    unsigned long long xs = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc += a[i]; xs += u[i]; }
#pragma unroll
    for (int i = 0; i < 8; i++) xs ^= w[i];
    xs += sl;
    if (acc == 12345.678f || xs == 0x123456789abcull) sink[threadIdx.x] = acc;       // never true: keeps the streams alive
}

typedef void (*kern_t)(float*, Stamp*, int, float, float);
template <int... I> static void fill(kern_t* k, std::integer_sequence<int, I...>) { ((k[I] = stream_kernel<I>), ...); }

int main()
{
    kern_t kern[N_OPS];
    fill(kern, std::make_integer_sequence<int, N_OPS>{});
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* sink; CK(hipMalloc(&sink, 4096));
    const int max_blocks = n_cu * 8 * 8;
    Stamp* d_st; CK(hipMalloc(&d_st, sizeof(Stamp) * max_blocks));
    std::vector<Stamp> st(max_blocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# issue cost of the vocoder's other vector instructions by wall clock (`tools/ubench/valu_ops.hip`), %s, %d CUs\n\n", prop.gcnArchName, n_cu);
    printf("Frame of `r05_valu_wallclock.md`: every wave runs iters x 256 instructions of ONE opcode on 16 rotating registers (8 register pairs for the 64-bit ones);\n"
           "cycles per wave-instruction per SIMD = hipEvent seconds x held clock x (n_cu x 4) / wave-instructions, at exactly W waves per SIMD; `x fma` = against\n"
           "`v_fma_f32` at the same W.\n\n");
    printf("| opcode | W = 1: ms | clock GHz | cycles | W = 2: ms | clock GHz | cycles | W = 8: ms | clock GHz | cycles | x fma at W = 8 |\n|---|---|---|---|---|---|---|---|---|---|---|\n");
    double fma8 = 0;
    for (int op = 0; op < N_OPS; op++) {
        printf("| %s |", kName[op]);
        double cyc8 = 0;
        for (int W : {1, 2, 8}) {
            const int rounds = W == 8 ? 2 : W == 2 ? 8 : 16;
            const int iters = 512;
            const int blocks = n_cu * W * rounds;
            const size_t lds_bytes = W == 1 ? 81 * 1024 : W == 2 ? 54 * 1024 : 0;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern[op]), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            const double wave_instr = (double)blocks * 4 * iters * 256.0;
            std::vector<double> ms_all;
            for (int l = 0; l < 5; l++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern[op], dim3(blocks), dim3(256), lds_bytes, 0, sink, d_st, iters, 1.0000001f, 1e-9f);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (l > 0) ms_all.push_back(ms);
            }
            CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
            std::vector<double> clk;
            for (int i = 0; i < blocks; i++) {
                const double dc = (double)(st[i].cyc1 - st[i].cyc0), dr = (double)(st[i].real1 - st[i].real0);
                if (dr > 0) clk.push_back(dc / dr * 0.1);
            }
            std::sort(clk.begin(), clk.end());
            const double clock_ghz = clk.empty() ? 0 : clk[clk.size() / 2];
            std::sort(ms_all.begin(), ms_all.end());
            const double ms = ms_all[ms_all.size() / 2];
            const double cyc = ms * 1e-3 * clock_ghz * 1e9 * (n_cu * 4) / wave_instr;
            printf(" %.3f | %.2f | %.2f |", ms, clock_ghz, cyc);
            if (W == 8) cyc8 = cyc;
        }
        if (op == FMA) fma8 = cyc8;
        printf(" %.2f |\n", cyc8 / fma8);
        fflush(stdout);
    }
    return 0;
}
