#!/bin/bash
# r05: how much of the vocoder's time are its HALF-RATE vector instructions (profiles/r05_valu_ops.md: v_cmp, v_cndmask, v_min / v_max, v_cvt, v_rndne, every integer
# multiply ... retire in 4.1 cycles per wave64 instruction per SIMD against 2.15 for f32 add / mul / fma)?  Builds a variant whose atan2 carries the SAME NUMBER of
# instructions with every half-rate one replaced by a full-rate stand-in (WRONG RESULTS: a probe, never shipped) and times it against the product build.
#   tools/experiments/r05_halfrate_probe.sh build      (here)       then on the GPU box:   tools/ab_libs.sh base atanprobe
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
V=/tmp/vsrc_atanprobe
rm -rf $V && mkdir -p $V
cp $R/nodey-audio-editor_amd/csrc/*.h $R/nodey-audio-editor_amd/csrc/kernels_pvpipe.hip $V/
python3 - "$V/stft_device.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
a = s.index("__device__ __forceinline__ uint32_t atan2_q32(float im, float re)")
b = s.index("// strided signal access")
probe = '''__device__ __forceinline__ uint32_t atan2_q32(float im, float re)
{
    // PROBE (wrong results): the canonical instruction count, every half-rate instruction replaced by ONE full-rate instruction
    const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
    float mx, mn;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(mx) : "v"(ax), "v"(ay));                 // for v_max3_f32
    asm volatile("v_mul_f32 %0, 0.5, %1" : "=v"(mn) : "v"(ax));                         // for v_min3_f32
    float r = __uint_as_float(NAE_RCP_MAGIC - __float_as_uint(mx));
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const float e = __builtin_fmaf(-mx, r, 1.0f);
        r = __builtin_fmaf(r, e, r);
    }
    const float t = mn * r;
    const float s = t * t;
    float q = NAE_ATAN_C6 * NAE_ATAN_SCALE;
    q = __builtin_fmaf(q, s, NAE_ATAN_C5 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C4 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C3 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C2 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C1 * NAE_ATAN_SCALE);
    q = __builtin_fmaf(q, s, NAE_ATAN_C0 * NAE_ATAN_SCALE);
    uint32_t i = __float_as_uint(q * t);
    asm volatile("v_xor_b32 %0, 0x1234, %0" : "+v"(i));                                 // for v_rndne_f32
    asm volatile("v_add_u32 %0, 77, %0" : "+v"(i));                                     // for v_cvt_i32_f32
    uint32_t j = 0x40000000u - i;
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(i) : "v"(__float_as_uint(ay)));          // for v_cmp_gt_f32
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i) : "v"(j));                            // for v_cndmask_b32
    const uint32_t m_re = (uint32_t)((int32_t)__float_as_uint(re) >> 31), m_im = (uint32_t)((int32_t)__float_as_uint(im) >> 31);
    i = (i ^ m_re) + (m_re & 0x80000001u);
    i = (i ^ m_im) - m_im;
    const float sum = ax + ay;
    asm volatile("v_and_b32 %0, %0, %1" : "+v"(i) : "v"(__float_as_uint(sum)));         // for v_cmp_lt_f32
    asm volatile("v_add_u32 %0, 1, %0" : "+v"(i));                                      // for v_cndmask_b32
    return i;
}

'''
s = s[:a] + probe + s[b:]
open(p, "w").write(s)
PY
SRC_PVPIPE=$V/kernels_pvpipe.hip bash $R/tools/mkvariant.sh atanprobe
