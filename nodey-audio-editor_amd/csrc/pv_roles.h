// pv_roles.h — what the wave pipelines of the phase vocoder share (kernels_pvpipe.hip: two barriers per step, one buffer per hand-off;
// kernels_pvflow.hip: one barrier per step, two buffers per hand-off): the barrier, the opaque lane index, the exact phase increment and the
// rotation of a bin.  The arithmetic is the canonical one of DESIGN.md §3.
#pragma once
#include "stft_common.h"

namespace nae {

constexpr int kYCf = 520;                                   // Y[0..512] natural order
constexpr int kOlaQuarter = 256;                            // floats

// every LDS operation of this wave has completed, then the workgroup barrier (vector-memory operations stay in flight:
// the frame prefetch of R1 and the block stores of R3 must not be drained twice per step)
__device__ __forceinline__ void pipe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the lane index as a value the optimiser cannot see through: addresses derived from it are recomputed where they are used
// (one to four instructions each) instead of being hoisted out of the frame loop, where each would pin a VGPR of the 64.
// Only where that matters: R1 and R3 of the 64-VGPR build; the phase roles have registers to spare (their hoisted addresses
// stay below the kernel's maximum: -1.7 % kernel time), and the 128-VGPR builds hide nothing
template <bool kHide = true>
__device__ __forceinline__ int pipe_lane(int lane)
{
    if (kHide) asm volatile("" : "+v"(lane));
    return lane;
}

// exact phase increment of one hop for bin k: adv + round(dw * R / 2^24)
__device__ __forceinline__ uint32_t pipe_inc(uint32_t qa, uint32_t qp, unsigned k, unsigned d, unsigned R)
{
    const uint32_t e = ((k * d) & (NAE_FFT_N - 1)) << 22;
    const int32_t dw = (int32_t)(qa - qp - e);
    const uint32_t adv = ((k * NAE_HOP) & (NAE_FFT_N - 1)) << 22;
    const long long scaled = ((long long)dw * (long long)(int32_t)R + (1ll << (NAE_R_FRAC_BITS - 1))) >> NAE_R_FRAC_BITS;
    return adv + (uint32_t)scaled;
}

// synthesis bin X e^{i (qs - qa)} (tolerance path: v_sin / v_cos take turns)
__device__ __forceinline__ cf pipe_rotate(cf x, uint32_t qs, uint32_t qa)
{
    const float ph = (float)(int32_t)(qs - qa) * (1.0f / 4294967296.0f);
    const float cs = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
    return cf{__builtin_fmaf(x.x, cs, -(x.y * sn)), __builtin_fmaf(x.x, sn, x.y * cs)};
}

// issue priority of a wave when two workgroups share a CU (kernels_pvpipe.hip, "Issue priority"): the workgroups take turns at the higher level in
// time slices of 2^18 shader cycles, R3 sits one level above its workgroup's
constexpr int kPrioSliceBit = 18;
__device__ __forceinline__ void pipe_prio(int slot, int role, unsigned long long now)
{
    const int turn = (int)((now >> kPrioSliceBit) & 1);
    const int lvl = (((turn + slot) & 1) ? 2 : 0) + (role == 3 ? 1 : 0);                // wave-uniform
    if (lvl == 0) __builtin_amdgcn_s_setprio(0);
    else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// which of the two workgroups of its CU this one is (0 / 1, by arrival; thread 0 of the workgroup calls it): the parity of a counter per physical CU
__device__ __forceinline__ int pipe_cu_arrival(unsigned* arrivals)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_ID: CU 8-11, SE 13-14
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;     // XCC_ID
    const unsigned key = (xcc * 4 + ((hw >> 13) & 3u)) * 16 + ((hw >> 8) & 15u);
    return (int)(atomicAdd(&arrivals[key], 1u) & 1u);
}

} // namespace nae
