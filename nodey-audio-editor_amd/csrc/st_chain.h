// st_chain.h — host-side bookkeeping of the SoundTouch-shaped time-domain chain (K7 option A).
//
// The chain has three stages — TD (WSOLA stretcher), AA (64-tap anti-alias FIR), CU (cubic transposer) — ordered
// TD->AA->CU for rate > 1, AA->CU->TD for rate == 1, CU->AA->TD for rate < 1 (SoundTouch 2.3.2 as driven by
// /root/reference/src/processor/audio-velocity.cpp:369-428; the library itself is absent).
// Every COUNT in the chain (how many sequences, how many filter outputs, where each cubic output reads) is
// independent of the audio data, so it is computed here on the host with the library's own scalar arithmetic;
// the kernels of kernels_wsola.hip then do the data work for whole index ranges at once.
#pragma once
#include <stdint.h>
#include <vector>
#include "nae_internal.h"

namespace nae {

constexpr int kAaLen = 64;

struct StCfg {
    int sr = 0, ch = 0;
    double rate = 1.0, tempo = 1.0;   // effective: rate = pitch * rate_in, tempo = 1 / pitch
    int order = 0;                    // 0: TD,AA,CU   1: AA,CU,TD   2: CU,AA,TD
    int ovl = 0, swl = 0, seekl = 0, sample_req = 0, body = 0, first_skip = 0;
    double nominal_skip = 0.0;
    float aa[kAaLen];
};

// absolute counters since the start of the stream
struct StState {
    long long td_in = 0, td_ip = 0, td_nseq = 0, td_out = 0;
    double td_skip = 0.0;
    bool td_begin = true;
    long long aa_in = 0, aa_out = 0;                  // aa_out = outputs made = inputs consumed
    long long cu_in = 0, cu_pos = 0, cu_out = 0;      // cu_pos = inputs consumed
    double cu_fract = 0.0;
    double expected = 0.0;
};

// where each cubic output reads: input frame index (absolute) and the fraction behind it
struct CuTable {
    long long origin = 0;             // output index of element 0
    std::vector<long long> pos;
    std::vector<float> fract;
    void drop_before(long long n)
    {
        if (n <= origin) return;
        const size_t k = (size_t)(n - origin) < pos.size() ? (size_t)(n - origin) : pos.size();
        pos.erase(pos.begin(), pos.begin() + (long)k);
        fract.erase(fract.begin(), fract.begin() + (long)k);
        origin = n;
    }
};

int st_cfg_make(StCfg& c, int sample_rate, int channels, double rate, double pitch);
// feed n more input frames (SoundTouch::putSamples); appends the new cubic read positions to `tab` if given
void st_sim_put(const StCfg& c, StState& s, long long n, CuTable* tab);
long long st_final_out(const StCfg& c, const StState& s);   // frames the last stage has produced

// ---- kernels_wsola.hip
struct StView {              // read side: frame a (absolute), channel k of stream s = base[s*ss + k*cs + (a-origin)*fs];
    const float* base;       // frames at or beyond valid_end read as zero (flush padding)
    long long ss, cs, fs, origin, valid_end;
};
struct StOut {
    float* base;
    long long ss, cs, fs, origin;
};
struct TdRange {             // nseq sequences starting from this state; output frames at or beyond out_limit are dropped
    long long ip0, op0, nseq;
    double skip0;
    int begin0;
    long long out_limit;
};
int st_launch_td(nae_ctx* ctx, const StCfg& c, const StView& in, const TdRange& r, const StOut& out, size_t n_streams,
                 float* mid_state, int32_t* offs_dbg, long long offs_stride);
int st_launch_aa(nae_ctx* ctx, const StCfg& c, const StView& in, long long j0, long long j1, const StOut& out,
                 size_t n_streams);
int st_launch_aa_cu(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract, const int* d_tile_n,
                    size_t tiles, long long n_limit, const StOut& out, size_t n_streams);
int st_launch_cu_aa(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract, long long n_cu,
                    long long j1, const StOut& out, size_t n_streams);
int st_launch_cu_table(nae_ctx* ctx, long long pos0, double fract0, double rate, long long count, long long* d_pos, float* d_fract);
void st_tile_starts(const CuTable& tab, long long n_limit, std::vector<int>& tile_n);
int st_launch_cu(nae_ctx* ctx, const StCfg& c, const StView& in, const long long* d_pos, const float* d_fract,
                 long long tab_origin, long long n0, long long n1, const StOut& out, size_t n_streams);

} // namespace nae
