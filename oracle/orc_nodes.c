/* orc_nodes.c — CPU ORACLE (test infrastructure only; see nae_oracle.h).
 *
 * Literal restatements of the reference's per-frame inner loops K1..K6.  Compiled like the reference's
 * release build (/root/reference/xmake.lua:1 "mode.release" => -O3, x86-64 baseline, no fast-math) plus
 * -ffp-contract=off, so no multiply-add is ever fused — the reference's x86-64 baseline build cannot
 * fuse either (no FMA instructions without -march).
 *
 * PARITY PIN: no reference fixture exists; pinned by tests/golden vectors from the
 * independent numpy restatement tests/golden/gen_golden.py.
 */
#include "nae_oracle.h"
#include "../include/nae_dsp_spec.h"
#include <math.h>
#include <string.h>

/* ---------------------------------------------------------------- K1 gain
 * audio-vol.cpp:75-100: for each plane, std::copy(src, src+n, dst) then `dst[i] *= volume`.
 * For integer T the compound assignment is  dst[i] = T(float(dst[i]) * volume): int -> float (RNE),
 * one f32 multiply, then a TRUNCATING float -> T conversion with no clamp (apply_volume at :14-29,
 * which clamps, is never called).  Out-of-range behaviour is what x86-64 does: cvttss2si to int32
 * ("integer indefinite" 0x80000000 on overflow/NaN), then a modular narrowing to int16_t. */
void orc_change_volume_f32(float* const* dst, const float* const* src, int planes, int elems, float volume)
{
    for (int ch = 0; ch < planes; ch++) {
        memmove(dst[ch], src[ch], (size_t)elems * sizeof(float)); /* :95 std::copy */
        float* d = dst[ch];
        for (int i = 0; i < elems; i++) d[i] *= volume;           /* :98 */
    }
}

static inline int32_t x86_cvttss2si(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return NAE_X86_INT_INDEFINITE; /* incl. NaN */
    return (int32_t)f;
}

void orc_change_volume_s16(int16_t* const* dst, const int16_t* const* src, int planes, int elems, float volume)
{
    for (int ch = 0; ch < planes; ch++) {
        memmove(dst[ch], src[ch], (size_t)elems * sizeof(int16_t));
        int16_t* d = dst[ch];
        for (int i = 0; i < elems; i++) d[i] = (int16_t)(uint16_t)(uint32_t)x86_cvttss2si((float)d[i] * volume);
    }
}

void orc_change_volume_s32(int32_t* const* dst, const int32_t* const* src, int planes, int elems, float volume)
{
    for (int ch = 0; ch < planes; ch++) {
        memmove(dst[ch], src[ch], (size_t)elems * sizeof(int32_t));
        int32_t* d = dst[ch];
        for (int i = 0; i < elems; i++) d[i] = x86_cvttss2si((float)d[i] * volume);
    }
}

/* ---------------------------------------------------------------- K2 split / merge */
/* audio-velocity.cpp:169-180: channel-outer, sample-inner, strided store */
void orc_interleave_f32(const float* const* src_planes, float* dst, size_t S, int ch)
{
    for (int c = 0; c < ch; ++c) {
        float* p = dst + c;
        for (size_t i = 0; i < S; ++i) {
            *p = src_planes[c][i];
            p += ch;
        }
    }
}

/* swr_convert FLT -> FLTP at equal rate and layout (audio-amix.cpp:263-269, audio-bimix.cpp:259-265):
 * libswresample reduces to a per-channel strided copy. */
void orc_deinterleave_f32(const float* src, float* const* dst_planes, size_t S, int ch)
{
    for (int c = 0; c < ch; ++c)
        for (size_t i = 0; i < S; ++i) dst_planes[c][i] = src[i * (size_t)ch + c];
}

/* ---------------------------------------------------------------- K3 amix
 * audio-amix.cpp:296-307: sample-outer, input-inner, accumulate from 0.0f, multiply then add. */
void orc_amix_f32(const float* const* inL, const float* const* inR, const float* vol, int n, float* outL,
                  float* outR, size_t S)
{
    for (size_t j = 0; j < S; j++) {
        float temp_l = 0.0f;
        float temp_r = 0.0f;
        for (int i = 0; i < n; i++) {
            temp_l += inL[i][j] * vol[i];
            temp_r += inR[i][j] * vol[i];
        }
        outL[j] = temp_l;
        outR[j] = temp_r;
    }
}

/* audio-amix.cpp:379-387 (runs every GUI frame; the slider-edit branch :360-371 is GUI-only) */
void orc_amix_normalise_volumes(float* volumes, const uint8_t* locks, int n)
{
    float unlocked_volume_sum = 0.0f;
    for (int i = 0; i < n; i++) unlocked_volume_sum += locks[i] ? 0.0f : volumes[i];
    unlocked_volume_sum = unlocked_volume_sum > 0.001f ? unlocked_volume_sum : 0.001f; /* std::max */
    for (int i = 0; i < n; i++) {
        if (locks[i]) continue;
        volumes[i] /= unlocked_volume_sum;
    }
}

/* ---------------------------------------------------------------- K4 bimix v1
 * audio-bimix.cpp:310-317 */
void orc_bimix_f32(const float* ll, const float* lr, const float* rl, const float* rr, float bias, float* outL,
                   float* outR, size_t S)
{
    const float bias_minus = (1 - bias);
    const float bias_plus = (1 + bias);
    for (size_t i = 0; i < S; i++) {
        outL[i] = (ll[i] / 2 + lr[i] / 2) * bias_minus;
        outR[i] = (rl[i] / 2 + rr[i] / 2) * bias_plus;
    }
}

/* ---------------------------------------------------------------- K5 bimix v2
 * audio-bimix.cpp:624-627: dst = (left + right) * 0.5  — float add, then a DOUBLE multiply by 0.5,
 * then narrowing to float on assignment. */
void orc_bimix2_downmix_f32(const float* l, const float* r, float* mono, size_t S)
{
    for (size_t i = 0; i < S; i++) mono[i] = (float)((l[i] + r[i]) * 0.5);
}

/* audio-bimix.cpp:833-850 (and :797-803 with aligned == 0; the single-sided tails :736-742 / :759-765
 * are the same loop with later == NULL): the earlier stream fills channel `earlier_offset`, the other
 * channel is zero for the first `unaligned` samples and then carries `later`. */
void orc_bimix2_interleave_f32(float* dst, const float* earlier, const float* later, size_t unaligned,
                               size_t aligned, int earlier_offset)
{
    const int later_offset = 1 - earlier_offset;
    for (size_t i = 0; i < unaligned; i++) {
        dst[i * 2 + earlier_offset] = earlier[i];
        dst[i * 2 + later_offset] = 0;
    }
    for (size_t i = 0; i < aligned; i++) {
        dst[(i + unaligned) * 2 + earlier_offset] = earlier[i + unaligned];
        dst[(i + unaligned) * 2 + later_offset] = later[i];
    }
}

/* ---------------------------------------------------------------- K6 format -> interleaved f32
 * audio-velocity.cpp:160-229; the four integer divisors differ and are kept literally. */
int orc_to_f32_interleaved(int fmt, const void* const* planes, size_t S, int ch, float* dst)
{
    const size_t n = S * (size_t)ch;
    switch (fmt) {
    case ORC_FMT_FLT: /* :162-168 */
        memmove(dst, planes[0], n * sizeof(float));
        return 0;
    case ORC_FMT_FLTP: /* :169-180 */
        for (int c = 0; c < ch; ++c) {
            float* p = dst + c;
            const float* s = (const float*)planes[c];
            for (size_t i = 0; i < S; ++i, p += ch) *p = s[i];
        }
        return 0;
    case ORC_FMT_S16: { /* :181-188 */
        const int16_t* s = (const int16_t*)planes[0];
        for (size_t i = 0; i < n; ++i) dst[i] = (float)s[i] / 32768.0f;
        return 0;
    }
    case ORC_FMT_S16P: /* :189-201  (float)x / int(32767) */
        for (int c = 0; c < ch; ++c) {
            float* p = dst + c;
            const int16_t* s = (const int16_t*)planes[c];
            for (size_t i = 0; i < S; ++i, p += ch) *p = (float)s[i] / 32767;
        }
        return 0;
    case ORC_FMT_S32: { /* :202-209 */
        const int32_t* s = (const int32_t*)planes[0];
        for (size_t i = 0; i < n; ++i) dst[i] = (float)s[i] / 2147483648.0f;
        return 0;
    }
    case ORC_FMT_S32P: /* :210-222  (double)x / int(2147483647), narrowed on store */
        for (int c = 0; c < ch; ++c) {
            float* p = dst + c;
            const int32_t* s = (const int32_t*)planes[c];
            for (size_t i = 0; i < S; ++i, p += ch) *p = (float)((double)s[i] / 2147483647);
        }
        return 0;
    default: /* :223-228 throws Runtime_error("Unsupported sample format") */
        return -1;
    }
}

/* ---------------------------------------------------------------- sink clamp
 * audio-io.cpp:617-618  std::clamp<float>(val, -1.0, +1.0): (v < lo) ? lo : (hi < v) ? hi : v */
void orc_clamp_f32(float* data, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const float v = data[i];
        data[i] = (v < -1.0f) ? -1.0f : (1.0f < v) ? 1.0f : v;
    }
}

/* ---------------------------------------------------------------- synthetic input */
void orc_fill_uniform(float* dst, size_t n, uint64_t seed)
{
    uint64_t x = seed;
    for (size_t i = 0; i < n; i++) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const uint32_t u = (uint32_t)(z >> 32);
        dst[i] = (float)(u >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
}
