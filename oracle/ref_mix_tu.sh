#!/bin/sh
# oracle/ref_mix_tu.sh — writes to stdout the translation unit of oracle/_ref/libref_mix*.so (target `ref` of oracle/Makefile).
# Test infrastructure.  The ARITHMETIC STATEMENTS of three more reference loops are streamed, by line number, from where they lie
# under /root/reference into wrapper functions written here; nothing of the reference is copied into the repo or left on disk.
#   K3  audio-amix.cpp:298-306    the body of the sample loop of the N-input mixer (accumulators, input loop, two stores)
#   K4  audio-bimix.cpp:310-311   bias_minus / bias_plus;  :315-316  the two output statements of the channel mixer
#   K5  audio-bimix.cpp:627       the mono downmix statement of the channel mixer v2
# What is NOT the reference's: the loop headers (the reference bounds them by AVFrame::nb_samples / a C++23 views::zip, neither of
# which exists in this image — the statements themselves use only float pointers, std::vector and scalars) and the declarations of
# the variables the statements name, which repeat the reference's own types (audio-amix.cpp:244 `std::vector<uint8_t**> datas`,
# include/processor/audio-amix.hpp:37 `std::vector<float> volumes`, audio-bimix.cpp:303-308 float pointers, :515,533 std::vector<float>).
# No stand-in for any FFmpeg / Boost / JsonCpp header is written.  So these pins cover the arithmetic and its order of
# operations as the reference's compiler sees them, not the frame plumbing around them.
REF=${1:-/root/reference}
AMIX=$REF/src/processor/audio-amix.cpp
BIMIX=$REF/src/processor/audio-bimix.cpp
set -e
# the cited lines are still where SURVEY.md says
sed -n '298p' $AMIX | grep -q 'float temp_l = 0.0f;'
sed -n '306p' $AMIX | grep -q 'out_right\[j\] = temp_r;'
sed -n '310p' $BIMIX | grep -q 'const float bias_minus = (1 - bias);'
sed -n '315p' $BIMIX | grep -q 'out_left\[i\] = (float_data_ll\[i\] / 2 + float_data_lr\[i\] / 2) \* bias_minus;'
sed -n '627p' $BIMIX | grep -q 'dst = (left + right) \* 0.5;'
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
