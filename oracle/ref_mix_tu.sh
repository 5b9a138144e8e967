#!/bin/sh
# oracle/ref_mix_tu.sh — writes to stdout the translation unit of oracle/_ref/libref_mix*.so (target `ref` of oracle/Makefile).
# Test infrastructure.  The ARITHMETIC STATEMENTS of three more reference loops are streamed, by line number, from where they lie
# under /root/reference into wrapper functions written here; nothing of the reference is copied into the repo or left on disk.
#   K3  audio-amix.cpp:298-306    the body of the sample loop of the N-input mixer (accumulators, input loop, two stores)
#   K4  audio-bimix.cpp:310-311   bias_minus / bias_plus;  :315-316  the two output statements of the channel mixer
#   K5  audio-bimix.cpp:627       the mono downmix statement of the channel mixer v2
#   K5  audio-bimix.cpp:799-803 and :836-850   the interleave loops of the channel mixer v2 WITH their headers (they are bounded by
#       plain size_t counts): the single-sided frame and the unaligned + aligned frame
# What is NOT the reference's: the loop headers (the reference bounds them by AVFrame::nb_samples / a C++23 views::zip, neither of
# which exists in this image — the statements themselves use only float pointers, std::vector and scalars) and the declarations of
# the variables the statements name, which repeat the reference's own types (audio-amix.cpp:244 `std::vector<uint8_t**> datas`,
# include/processor/audio-amix.hpp:37 `std::vector<float> volumes`, audio-bimix.cpp:303-308 float pointers, :515,533 std::vector<float>).
# No stand-in for any FFmpeg / Boost / JsonCpp header is written.  So these pins cover the arithmetic and its order of
# operations as the reference's compiler sees them, not the frame plumbing around them.
#   K3  audio-amix.cpp:379-387    the renormalisation of the unlocked mixer weights (loops with their headers: plain ints and vectors)
#   K6  audio-velocity.cpp:186 and :207   the two conversion LAMBDAS of extract_samples_interleaved (packed S16 and packed S32: each
#       is a whole lambda expression on one line); the planar branches (:196-197, :217-218) read AVFrame fields inside the same
#       statement and are not compiled
REF=${1:-/root/reference}
AMIX=$REF/src/processor/audio-amix.cpp
BIMIX=$REF/src/processor/audio-bimix.cpp
VELO=$REF/src/processor/audio-velocity.cpp
set -e
# the cited lines are still where SURVEY.md says
sed -n '298p' $AMIX | grep -q 'float temp_l = 0.0f;'
sed -n '306p' $AMIX | grep -q 'out_right\[j\] = temp_r;'
sed -n '310p' $BIMIX | grep -q 'const float bias_minus = (1 - bias);'
sed -n '315p' $BIMIX | grep -q 'out_left\[i\] = (float_data_ll\[i\] / 2 + float_data_lr\[i\] / 2) \* bias_minus;'
sed -n '627p' $BIMIX | grep -q 'dst = (left + right) \* 0.5;'
sed -n '799p' $BIMIX | grep -q 'for (size_t i = 0; i < eariler_stream.front().samples.size(); i++)'
sed -n '837p' $BIMIX | grep -q 'for (size_t i = 0; i < unaligned_samples; i++)'
sed -n '850p' $BIMIX | grep -q '^[[:space:]]*}$'
sed -n '379p' $AMIX | grep -q 'float unlocked_volume_sum = 0.0f;'
sed -n '386p' $AMIX | grep -q 'volumes\[i\] /= unlocked_volume_sum;'
sed -n '186p' $VELO | grep -q '\[\](int16_t sample) { return static_cast<float>(sample) / 32768.0f; }'
sed -n '207p' $VELO | grep -q '\[\](int32_t sample) { return static_cast<float>(sample) / 2147483648.0f; }'
cat <<'PRE'
#include <cstddef>
#include <cstdint>
#include <vector>
extern "C" void ref_amix_f32(uint8_t** const* datas_in, const float* volumes_in, int input_num, float* out_left, float* out_right, int n)
{
    std::vector<uint8_t**> datas(datas_in, datas_in + input_num);      // audio-amix.cpp:244
    std::vector<float> volumes(volumes_in, volumes_in + input_num);     // include/processor/audio-amix.hpp:37
    for (int j = 0; j < n; j++)
    {
PRE
sed -n '298,306p' $AMIX
cat <<'MID1'
    }
}
extern "C" void ref_bimix_f32(const float* float_data_ll, const float* float_data_lr, const float* float_data_rl, const float* float_data_rr,
                              float bias, float* out_left, float* out_right, size_t n)
{
MID1
sed -n '310,311p' $BIMIX
cat <<'MID2'
    for (size_t i = 0; i < n; i++)
    {
MID2
sed -n '315,316p' $BIMIX
cat <<'MID3'
    }
}
extern "C" void ref_bimix2_downmix_f32(const float* l, const float* r, float* mono, size_t n)
{
    std::vector<float> samples(n), shared_buffer[2] = {std::vector<float>(l, l + n), std::vector<float>(r, r + n)};   // audio-bimix.cpp:515,533
    for (size_t k = 0; k < n; k++)
    {
        float& dst = samples[k];
        float& left = shared_buffer[0][k];
        float& right = shared_buffer[1][k];
MID3
sed -n '627p' $BIMIX
cat <<'POST'
    }
    for (size_t k = 0; k < n; k++) mono[k] = samples[k];
}
POST
cat <<'MID4'
#include <list>
namespace
{
    struct Frame { std::vector<float> samples; };      // audio-bimix.cpp:513-516 (the loops touch `samples` only)
}
// later == nullptr: the single-sided frame (:797-803); else the unaligned + aligned frame (:833-850).  dst holds 2 * (unaligned + aligned) floats
extern "C" void ref_bimix2_interleave_f32(float* dst, const float* earlier, size_t n_earlier, const float* later, size_t n_later,
                                          size_t unaligned_samples, size_t aligned_samples, int earlier_channel)
{
    std::list<Frame> eariler_stream(1), later_stream(1);                 // :528 std::list<Frame> frames_l, frames_r
    eariler_stream.front().samples.assign(earlier, earlier + n_earlier);
    if (later) later_stream.front().samples.assign(later, later + n_later);
    const size_t eariler_offset = earlier_channel ? 1 : 0;               // :783-784
    const size_t later_offset = earlier_channel ? 0 : 1;
    std::vector<float> frame_samples;                                    // :534
    if (!later)
    {
        frame_samples.resize(eariler_stream.front().samples.size() * 2);
MID4
sed -n '799,803p' $BIMIX
cat <<'MID5'
    }
    else
    {
        frame_samples.resize((unaligned_samples + aligned_samples) * 2);
MID5
sed -n '836,850p' $BIMIX
cat <<'POST2'
    }
    for (size_t k = 0; k < frame_samples.size(); k++) dst[k] = frame_samples[k];
}
extern "C" void ref_k6_s16_packed(const int16_t* in, float* out, size_t n)
{
    auto convert =
POST2
sed -n '186p' $VELO
cat <<'MID6'
    ;
    for (size_t k = 0; k < n; k++) out[k] = convert(in[k]);
}
extern "C" void ref_k6_s32_packed(const int32_t* in, float* out, size_t n)
{
    auto convert =
MID6
sed -n '207p' $VELO
cat <<'POST3'
    ;
    for (size_t k = 0; k < n; k++) out[k] = convert(in[k]);
}
#include <algorithm>
extern "C" void ref_amix_normalise(float* volumes_io, const unsigned char* locks_in, int input_num)
{
    std::vector<float> volumes(volumes_io, volumes_io + input_num);      // include/processor/audio-amix.hpp:37-38
    std::vector<bool> locks(locks_in, locks_in + input_num);
POST3
sed -n '379,387p' $AMIX
cat <<'POST4'
    for (int k = 0; k < input_num; k++) volumes_io[k] = volumes[k];
}
POST4
