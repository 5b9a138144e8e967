// Micro-benchmark (not part of the product): the chip's f32 vector rate by WALL CLOCK — no s_memtime anywhere in the figure.
// Every wave of the launch runs kIters x 256 instructions of one opcode on 16 rotating registers; the launch is timed with
// hipEvents; TFLOP/s = wave-instructions x 64 lanes x flops per lane-op / seconds.  Beside it (a second figure, not the first
// one's input): the shader clock the launch held, from s_memtime / s_memrealtime deltas of wave 0 of every workgroup.
// Shapes: 256-thread workgroups (one wave per SIMD); a dynamic-LDS request admits exactly W workgroups per CU (W = 1, 2, 4, 8
// waves per SIMD); the grid is n_cu x W x rounds workgroups, so every CU works through `rounds` of them per slot.
//   hipcc --offload-arch=gfx950 -O3 -o valu_wallclock valu_wallclock.hip && ./valu_wallclock > profiles/r05_valu_wallclock.md
// Under rocprofv3 (--pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU ...) pass "pmc" to run one launch per shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA, ADD, MUL, PK_FMA, MIX, N_OPS };
static const char* kName[N_OPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "butterfly mix (add/sub/mul/fma 5:5:3:3)"};
static const double kFlopsPerLaneOp[N_OPS] = {2, 1, 1, 4, 1.1875};   // mix: (10 add/sub + 3 mul + 3 fma x 2) / 16

struct Stamp { unsigned long long cyc0, cyc1, real0, real1; };

template <int OP>
__global__ __launch_bounds__(256) void stream_kernel(float* sink, Stamp* stamps, int iters, float b, float c)
{
    extern __shared__ float dyn_lds[];
    float a[16];
    float2 p[8];
    if (iters < 0) dyn_lds[threadIdx.x] = b;               // (never: keeps the dynamic LDS request attached to the kernel)
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = b * (float)(i + 1) + (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = float2{a[2 * i], a[2 * i + 1]};
    unsigned long long c0 = 0, r0 = 0;
    if ((threadIdx.x & 255) == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(p[(i + 4) & 7]));
                if (OP == MIX) {
                    if (i % 16 < 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                    else if (i % 16 < 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                    else if (i % 16 < 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 8) & 15]));
                }
            }
        }
    }
    if ((threadIdx.x & 255) == 0) {
        Stamp s{c0, __builtin_readcyclecounter(), r0, __builtin_amdgcn_s_memrealtime()};
        stamps[blockIdx.x] = s;
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) acc += p[i].x + p[i].y;
    if (acc == 12345.678f) sink[threadIdx.x] = acc;       // never true: keeps the stream alive
}

typedef void (*kern_t)(float*, Stamp*, int, float, float);
static kern_t kKern[N_OPS] = {stream_kernel<FMA>, stream_kernel<ADD>, stream_kernel<MUL>, stream_kernel<PK_FMA>, stream_kernel<MIX>};

int main(int argc, char** argv)
{
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* sink; CK(hipMalloc(&sink, 4096));
    const int max_blocks = n_cu * 8 * 8;
    Stamp* d_st; CK(hipMalloc(&d_st, sizeof(Stamp) * max_blocks));
    std::vector<Stamp> st(max_blocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# f32 vector rate by wall clock (`tools/ubench/valu_wallclock.hip`), %s, %d CUs\n\n", prop.gcnArchName, n_cu);
    printf("Each wave: iters x 256 instructions of one opcode on 16 rotating registers (an instruction's result is needed 16 instructions later).\n"
           "TFLOP/s = wave-instructions x 64 lanes x flops per lane-op / hipEvent seconds of the launch (median of 5 launches); `clock` = the shader\n"
           "clock held during the launch (s_memtime delta / s_memrealtime delta x 100 MHz, median over workgroups) — reported beside, not used for, the\n"
           "TFLOP/s.  `cycles per wave-instr per SIMD` = seconds x clock x (n_cu x 4 SIMDs) / wave-instructions: what a stamp-based measurement would read\n"
           "if every SIMD held exactly W waves for the whole launch.  Datasheet peak: 157.3 TFLOP/s f32 vector (256 CUs x 128 FMA lanes x 2 x 2.4 GHz).\n\n");
    printf("| opcode | waves per SIMD | rounds | ms per launch | TFLOP/s | clock GHz | cycles per wave-instr per SIMD | lanes retired per SIMD-cycle |\n|---|---|---|---|---|---|---|---|\n");
    for (int op = 0; op < N_OPS; op++) {
        for (int W : {1, 2, 4, 8}) {
            const int rounds = W == 8 ? 2 : W == 4 ? 4 : W == 2 ? 8 : 16;
            const int iters = pmc ? 512 : 2048;
            const int blocks = n_cu * W * rounds;
            // W workgroups fit a CU's 160 KiB, W + 1 do not (8: the wave slots limit)
            const size_t lds_bytes = W == 1 ? 81 * 1024 : W == 2 ? 54 * 1024 : W == 4 ? 33 * 1024 : 0;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kKern[op]), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            const double wave_instr = (double)blocks * 4 * iters * 256.0;
            std::vector<double> ms_all;
            double clock_ghz = 0;
            const int launches = pmc ? 1 : 6;
            for (int l = 0; l < launches; l++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kKern[op], dim3(blocks), dim3(256), lds_bytes, 0, sink, d_st, iters, 1.0000001f, 1e-9f);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (l > 0 || pmc) ms_all.push_back(ms);
            }
            CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
            std::vector<double> clk;
            for (int i = 0; i < blocks; i++) {
                const double dc = (double)(st[i].cyc1 - st[i].cyc0), dr = (double)(st[i].real1 - st[i].real0);
                if (dr > 0) clk.push_back(dc / dr * 0.1);
            }
            std::sort(clk.begin(), clk.end());
            clock_ghz = clk.empty() ? 0 : clk[clk.size() / 2];
            std::sort(ms_all.begin(), ms_all.end());
            const double ms = ms_all[ms_all.size() / 2];
            const double tflops = wave_instr * 64 * kFlopsPerLaneOp[op] / (ms * 1e-3) / 1e12;
            const double cyc = ms * 1e-3 * clock_ghz * 1e9 * (n_cu * 4) / wave_instr;
            printf("| %s | %d | %d | %.3f | %.1f | %.2f | %.2f | %.1f |\n", kName[op], W, rounds, ms, tflops, clock_ghz, cyc, 64.0 / cyc * (op == PK_FMA ? 2 : 1));
        }
    }
    return 0;
}
