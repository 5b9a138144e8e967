// Cross-lane form of transpose 1 of the 512-point FFT (not part of the product): measured SLOWER than the LDS form
// (profiles/r02_fftpad.md: 147 vs 141 cycles per FFT per CU at 8 waves per SIMD), kept here as the evidence.
#pragma once
namespace nae {
// Transpose 1 without LDS.  T1 exchanges the register index q with lane bits [5:3] (lane (m, qq) register j <- lane
// (m, j) register qq), i.e. three butterfly stages with partners lane ^ 32, ^ 16, ^ 8:
//   ^ 32, ^ 16: v_permlane32_swap / v_permlane16_swap exchange the upper half (odd rows) of one register with the lower
//               half (even rows) of another in ONE instruction — exactly the butterfly of a register pair;
//   ^ 8:        row_ror:8 DPP moves (a rotation by 8 inside a 16-lane row is lane ^ 8) with a bank mask selecting which
//               half-row is written: 3 moves per register pair.
// 40 full-rate vector instructions replace 8 ds_write_b64 + 8 ds_read_b64 (about 41 cycles of the CU's LDS pipe = the
// issue time of ~140 vector instructions, profiles/r02_valu_issue.md) and two LDS round trips of latency.
__device__ __forceinline__ void xlane_swap32(float& lo_keeps, float& hi_keeps)
{
    // afterwards: lanes 0-31 of `hi_keeps` hold what lanes 32-63 of `lo_keeps` held, and vice versa
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_keeps), __float_as_uint(hi_keeps), false, false);
    lo_keeps = __uint_as_float(r.x);
    hi_keeps = __uint_as_float(r.y);
}
__device__ __forceinline__ void xlane_swap16(float& lo_keeps, float& hi_keeps)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo_keeps), __float_as_uint(hi_keeps), false, false);
    lo_keeps = __uint_as_float(r.x);
    hi_keeps = __uint_as_float(r.y);
}
__device__ __forceinline__ void xlane_swap8(float& lo_keeps, float& hi_keeps)
{
    // lanes with bit 3 clear: hi_keeps <- partner's lo_keeps;  lanes with bit 3 set: lo_keeps <- partner's hi_keeps
    const unsigned a = __float_as_uint(lo_keeps), b = __float_as_uint(hi_keeps);
    const unsigned t = __builtin_amdgcn_update_dpp(0u, b, 0x128, 0xf, 0xf, false);        // b of lane ^ 8
    const unsigned b2 = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);       // banks 0,1 (bit 3 clear) <- a of lane ^ 8
    const unsigned a2 = __builtin_amdgcn_update_dpp(a, t, 0xE4, 0xf, 0xC, false);        // banks 2,3 (bit 3 set) <- t
    lo_keeps = __uint_as_float(a2);
    hi_keeps = __uint_as_float(b2);
}

__device__ __forceinline__ void t1_xlane(cf (&v)[8])
{
    // register bit 2 <-> lane bit 5
#pragma unroll
    for (int r = 0; r < 4; r++) { xlane_swap32(v[r].x, v[r + 4].x); xlane_swap32(v[r].y, v[r + 4].y); }
    // register bit 1 <-> lane bit 4
#pragma unroll
    for (int r = 0; r < 8; r++)
        if (!(r & 2)) { xlane_swap16(v[r].x, v[r + 2].x); xlane_swap16(v[r].y, v[r + 2].y); }
    // register bit 0 <-> lane bit 3
#pragma unroll
    for (int r = 0; r < 8; r += 2) { xlane_swap8(v[r].x, v[r + 1].x); xlane_swap8(v[r].y, v[r + 1].y); }
}

} // namespace nae
