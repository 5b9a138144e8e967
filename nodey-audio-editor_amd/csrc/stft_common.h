// stft_common.h — kernel-side types shared by the STFT translation units (kernels_stft.hip, kernels_pvpipe.hip).
#pragma once
#include "nae_internal.h"
#include "stft_device.h"

namespace nae {

constexpr int kT1024Pad = kPhasePad;             // 513 split twiddles / phases, padded to 520

struct Tables { const cf* w512; const cf* t1024; const float* hann; };
struct SigViewD { const float* base; long long ss, cs, fs; };
struct OutViewD { float* base; long long ss, cs, fs; };

// wave index as a SCALAR: hipcc cannot prove threadIdx.x >> 6 wave-uniform, and everything derived from it
// (stream / tile / frame addresses) would otherwise be carried in VGPRs with 64-bit vector address math
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

struct PvParams {
    long long ha_q24;
    long long in_len;     // valid input sample-frames per stream
    long long frames;     // F
    long long mid_len;    // PV-stage output samples wanted
    int d0;
    unsigned r_q24_0, r_q24_1;
    int ch;
    int tile;             // frames (== output hop blocks) per tile
    int n_tiles;
    long long f_origin;   // first frame / output block of tile 0 (0 in block mode; > 0 when a stream is continued)
    long long f_stop;     // one past the last frame / block this launch is responsible for
    int skip_from;        // pass 1 only: tiles >= skip_from are not analysed (their sums are not needed)
    int phase_step;       // pass 3 only: pass 1 ran on tiles `phase_step` times shorter (more waves for the same frames);
    int phase_tiles;      //              the base phase of tile t is record t * phase_step of `phase_tiles` per stream-channel
    int base_zero;        // pass 3: the base phase of every tile is zero (one tile per stream-channel, nothing carried in): no workspace read
    uint32_t* carry_out;  // pass 3, optional: receives the synthesis phase behind frame `carry_frame`, [stream-channel][520] (a continued
    long long carry_frame; //             stream whose segment is ONE tile: no pass 1 is needed just to carry the phase on)
};

__device__ __forceinline__ long long frame_start(const PvParams& p, long long f)
{
    return (((f - 1) * p.ha_q24 + (1ll << (NAE_HA_FRAC_BITS - 1))) >> NAE_HA_FRAC_BITS) - NAE_FFT_N / 2;
}

// canonical phases of this lane's 9 bins (k = lane + 64 r, and 512)
__device__ __forceinline__ void phases_of(const cf (&v)[8], cf nyq, uint32_t (&qa)[9])
{
#pragma unroll
    for (int r = 0; r < 8; r++) qa[r] = atan2_q32(v[r].y, v[r].x);
    qa[8] = (nyq.x < 0.0f) ? 0x80000000u : 0u;     // bin N/2 of a real signal is real
}

// phase increment of one hop for this lane's 9 bins (integer, exact)
__device__ __forceinline__ void phase_inc(const uint32_t (&qa)[9], const uint32_t (&qp)[9], uint32_t (&acc)[9],
                                          int kl, unsigned d, unsigned R)
{
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const unsigned k = (r < 8) ? (unsigned)(kl + 64 * r) : 512u;
        const uint32_t e = ((k * d) & (NAE_FFT_N - 1)) << 22;
        const int32_t dw = (int32_t)(qa[r] - qp[r] - e);
        const uint32_t adv = ((k * NAE_HOP) & (NAE_FFT_N - 1)) << 22;
        // R <= 2^26 (d >= 64), so it is a positive int32: one signed 32x32->64 multiply-add (v_mad_i64_i32)
        const long long scaled = ((long long)dw * (long long)(int32_t)R + (1ll << (NAE_R_FRAC_BITS - 1))) >> NAE_R_FRAC_BITS;
        acc[r] += adv + (uint32_t)scaled;
    }
}

} // namespace nae

// kernels_pvpipe.hip
int nae_launch_pv_pipe(nae_ctx* ctx, const nae::PvParams& p, const nae::SigViewD& src, long long n_sc, const uint32_t* phase_ws,
                       const nae::OutViewD& out, bool unit_stride, int frames_per_step);
