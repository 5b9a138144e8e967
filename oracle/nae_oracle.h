/* nae_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under nodey-audio-editor_amd/ includes, links or calls it.
 *
 * Two kinds of function live here:
 *  (1) literal restatements of the reference's CPU inner loops (K1..K6).  Each cites the reference
 *      file:line it follows (paths relative to /root/reference).  PARITY PIN: the reference ships no tests,
 *      fixtures or golden vectors and cannot be compiled in this image (no FFmpeg /
 *      SoundTouch / Boost / JsonCpp headers, libstdc++-11 lacks <expected>/<print>), so
 *      these are pinned by tests/golden/ vectors generated from an independent numpy-float32 restatement
 *      (tests/golden/gen_golden.py) — "parity pinned by construction, not by reference fixtures".
 *  (2) the builder-specified nodes K7 (tempo/pitch: phase vocoder + rate transposer) and K8 (FFT
 *      spectrum).  The reference has no code for K8 and calls SoundTouch 2.3.2 (absent) for K7, so
 *      versus the reference these are "PARITY UNPINNED"; K8 is pinned against scipy's float64 rfft and K7 against
 *      a float64 numpy restatement of its specification (tests/golden/pv_numpy.py -> k7_golden.npz, 1e-4 tolerance).
 */
#ifndef NAE_ORACLE_H
#define NAE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* sample formats: numeric values equal FFmpeg's AVSampleFormat so a reference-side caller can pass
 * frame->format through unchanged (libavutil/samplefmt.h: U8=0,S16=1,S32=2,FLT=3,DBL=4,U8P=5,S16P=6,
 * S32P=7,FLTP=8). */
enum { ORC_FMT_S16 = 1, ORC_FMT_S32 = 2, ORC_FMT_FLT = 3, ORC_FMT_S16P = 6, ORC_FMT_S32P = 7, ORC_FMT_FLTP = 8 };

/* ---- K1 gain: change_volume<T>, audio-vol.cpp:75-100 (two passes: copy, then scale) ---- */
void orc_change_volume_f32(float* const* dst, const float* const* src, int planes, int elems, float volume);
void orc_change_volume_s16(int16_t* const* dst, const int16_t* const* src, int planes, int elems, float volume);
void orc_change_volume_s32(int32_t* const* dst, const int32_t* const* src, int planes, int elems, float volume);

/* ---- K2 split / merge: audio-velocity.cpp:169-180 (planar->interleaved), swr FLT->FLTP as used at
 *      audio-amix.cpp:263-269 (interleaved->planar, pure copy at equal rate/layout) ---- */
void orc_interleave_f32(const float* const* src_planes, float* dst, size_t S, int ch);
void orc_deinterleave_f32(const float* src, float* const* dst_planes, size_t S, int ch);

/* ---- K3 N-input mix: audio-amix.cpp:293-307 ---- */
void orc_amix_f32(const float* const* inL, const float* const* inR, const float* vol, int n, float* outL,
                  float* outR, size_t S);
/* weight renormalisation done by the GUI each frame: audio-amix.cpp:379-387 */
void orc_amix_normalise_volumes(float* volumes, const uint8_t* locks, int n);

/* ---- K4 channel mix v1: audio-bimix.cpp:310-317 ---- */
void orc_bimix_f32(const float* ll, const float* lr, const float* rl, const float* rr, float bias, float* outL,
                   float* outR, size_t S);

/* ---- K5 channel mix v2: downmix audio-bimix.cpp:624-627,717-720; interleave w/ zero fill :797-803,
 *      :833-850; single-sided tails :736-742,:759-765 ---- */
void orc_bimix2_downmix_f32(const float* l, const float* r, float* mono, size_t S);
void orc_bimix2_interleave_f32(float* dst, const float* earlier, const float* later, size_t unaligned,
                               size_t aligned, int earlier_offset);

/* ---- K6 any format -> interleaved f32: audio-velocity.cpp:150-232.  returns 0, or -1 for an
 *      unsupported format (the reference throws Runtime_error there, :223-228) ---- */
int orc_to_f32_interleaved(int fmt, const void* const* planes, size_t S, int ch, float* dst);

/* ---- sink-side clamp (N4): audio-io.cpp:617-618 ---- */
void orc_clamp_f32(float* data, size_t n);

/* ---- canonical FFT pieces (spec: DESIGN.md §3) ---- */
void orc_fft512_fwd(const float* zin, float* zout);            /* 512 complex, interleaved re/im, natural order */
void orc_rfft1024(const float* xw, float* X);                  /* 1024 windowed reals -> 513 complex bins */
void orc_irfft1024(const float* X, float* y);                  /* 513 bins -> 1024 reals (1/N normalised; Im X[0], Im X[512] ignored) */
int32_t orc_atan2_q32(float im, float re);                     /* phase in turns, Q0.32 wrapping */
const float* orc_hann1024(void);

/* ---- K8 spectrum ---- */
size_t orc_spectrum_frames(size_t T);
void orc_spectrum_f32(const float* src_interleaved, size_t T, int ch, float* dst /* [frames][ch][513] */);

/* ---- K7 tempo/pitch (SoundTouch-shaped parameters: setRate(rate), setPitch(pitch),
 *      audio-velocity.cpp:384-385) ---- */
typedef struct orc_stretch_plan {
    int pv_on, rs_on;
    double tempo_eff;      /* = 1/pitch  (time-stretch factor of the PV stage: out length = in/tempo) */
    double rate_eff;       /* = rate*pitch (resampling factor of the transposer stage)               */
    int64_t ha_q24;        /* analysis hop, Q.24                                                      */
    int32_t d0;            /* floor(Ha)                                                               */
    uint32_t r_q24[2];     /* round(2^24*256/d) for d = d0, d0+1                                      */
    uint64_t step_q32;     /* transposer step, Q32.32                                                 */
    size_t out_len;        /* final output length for the given input length                          */
    size_t mid_len;        /* length of the intermediate signal between the two stages:
                              rs_first == 0: PV-stage samples the transposer needs (== out_len when rs_on == 0)
                              rs_first == 1: transposer output = PV-stage input length                  */
    size_t frames;         /* PV frames to run                                                        */
    int rs_first;          /* both stages on and rate_eff > 1: transposer FIRST (fewer samples reach the vocoder) */
} orc_stretch_plan;
int orc_stretch_plan_make(double rate, double pitch, size_t in_len, orc_stretch_plan* plan);
/* full node on one interleaved buffer; dst holds plan.out_len*ch floats */
int orc_stretch_f32(const float* src_interleaved, size_t L, int ch, double rate, double pitch, float* dst);
/* debugging taps: synthesis phase (Q0.32) of every frame, [frames][ch][513] */
int orc_pv_synth_phase(const float* src_interleaved, size_t L, int ch, const orc_stretch_plan* plan,
                       int32_t* qs);
const float* orc_rs_table(double rate_eff); /* (PHASES+1) x TAPS, rebuilt on every call into a static buffer */

/* ---- K7 option A (N1): SoundTouch-2.3.2-shaped WSOLA + anti-alias FIR + cubic transposer, streaming
 *      (orc_wsola.c; call sites audio-velocity.cpp:369-385,403,298,427).  PARITY UNPINNED. ---- */
typedef struct orc_st orc_st;
int orc_st_create(int sample_rate, int ch, double rate, double pitch, orc_st** h);
void orc_st_destroy(orc_st* s);
void orc_st_put(orc_st* s, const float* interleaved, size_t n);
size_t orc_st_available(const orc_st* s);
size_t orc_st_receive(orc_st* s, float* dst, size_t max);
void orc_st_flush(orc_st* s);
size_t orc_st_offsets(const orc_st* s, int32_t* dst, size_t max); /* overlap offsets chosen so far; returns their count */
void orc_st_params(const orc_st* s, int* v4);                     /* overlap, sequence, seek lengths, samples required */
const float* orc_st_aa_coef(const orc_st* s);                     /* 64 taps */
void orc_st_cubic_weights(float x, float* y4);
size_t orc_st_out_bound(size_t L, double rate, double pitch);
int orc_st_process_f32(const float* src_interleaved, size_t L, int ch, int sample_rate, double rate, double pitch,
                       float* dst, size_t* out_len);

/* ---- N2 input resampler (orc_swr.c): libswresample's default polyphase resampler, restated.  PARITY UNPINNED. ---- */
typedef struct orc_swr_plan {
    int in_rate, out_rate;
    int filter_length, filter_alloc, phase_count;
    int src_incr, dst_incr_div, dst_incr_mod;
    long long index0;      /* position of output 0 in 1/phase_count input samples: -phase_count * ((filter_length - 1) / 2) */
    double factor;
} orc_swr_plan;
int orc_swr_plan_make(int in_rate, int out_rate, orc_swr_plan* plan);              /* 0, -1 invalid, -2 unsupported ratio */
void orc_swr_build_filter(const orc_swr_plan* plan, float* bank /* [phase_count][filter_alloc] */);
void orc_swr_position(const orc_swr_plan* plan, uint64_t n, long long* first_sample, int* phase);
size_t orc_swr_reflection(const orc_swr_plan* plan, size_t n_in);
size_t orc_swr_outputs_upto(const orc_swr_plan* plan, size_t n_avail);
size_t orc_swr_out_len(const orc_swr_plan* plan, size_t n_in);
size_t orc_swr_resample_f32(const orc_swr_plan* plan, const float* bank, const float* x, size_t n_in, size_t stride, float* out,
                            size_t out_stride);

/* ---- synthetic inputs: splitmix64(seed) -> u32 -> float(u>>8)*2^-23 - 1 ---- */
void orc_fill_uniform(float* dst, size_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
