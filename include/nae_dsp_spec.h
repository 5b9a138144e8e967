/* nae_dsp_spec.h — numeric constants of the builder-defined DSP nodes (K7 tempo/pitch, K8 spectrum).
 *
 * The reference has NO code for the FFT spectrum node (FFTW is only a declared dependency,
 * /root/reference/xmake.lua:15,33) and delegates tempo/pitch to SoundTouch 2.3.2, which is not in the
 * tree (/root/reference/src/processor/audio-velocity.cpp:286,369-385).  BASELINE.json:north_star prescribes
 * "hand-written radix-2/4 LDS FFTs plus phase-vocoder analysis/resynthesis", so the algorithm is specified
 * HERE, once, and implemented twice: by the HIP kernels (nodey-audio-editor_amd/csrc) and, independently,
 * by the CPU oracle (oracle/).  Only constants live in this header — no shared code.
 *
 * The full algorithm text is DESIGN.md §3 ("K7/K8 specification").
 */
#ifndef NAE_DSP_SPEC_H
#define NAE_DSP_SPEC_H

/* STFT geometry (BASELINE.json configs[2]: "1024-pt FFT, 75 % overlap") */
#define NAE_FFT_N 1024          /* real frame length                       */
#define NAE_FFT_NC 512          /* packed complex length (N/2)             */
#define NAE_FFT_BINS 513        /* r2c bins k = 0..N/2                     */
#define NAE_HOP 256             /* synthesis hop / spectrum hop (N/4)      */
#define NAE_OLA_GAIN (2.0f / 3.0f) /* 1 / sum_t hann^2 at hop N/4 = 1/1.5  */

/* atan2 -> Q0.32 turns (revision 2 of the K7 / phase specification).  Every step is an IEEE f32 add / mul /
 * fma or a two's-complement integer operation, so C and the GPU give the same 32 bits — and none of them is a division:
 *   ax = |re|, ay = |im|,  mx = max(ax, ay, NAE_ATAN_TINY),  mn = min(ax, ay)
 *   r  = bits_to_float(NAE_RCP_MAGIC - float_to_bits(mx));   three times:  e = fma(-mx, r, 1),  r = fma(r, e, r)
 *   t  = mn * r,  s = t * t,  p = t * Q(s)      Q = Horner with fma over c6 .. c0, coefficients pre-scaled by 2^32:
 *                                               atan(t)/(2 pi) = t * Q(s) / 2^32, degree-6 minimax, max err 5.5e-8 turn
 *   i  = (int32) rint(p)                        first octant, 0 .. 2^29
 *   ay > ax:        i = 0x40000000 - i          (1/4 turn - angle)
 *   sign bit of re: i = 0x80000000 - i          (1/2 turn - angle;  wraps: +1/2 and -1/2 turn are the same phase)
 *   sign bit of im: i = -i
 * The octants follow the SIGN BITS (atan2(+0, -0) is half a turn, as in C); a vanishing bin (mx < 2^-100) has t = 0; a bin
 * with |re| + |im| not below NAE_ATAN_HUGE = 2^100 — which includes NaN, Inf and a sum that overflows — has phase 0 (beyond
 * 1.6e38 the integer seed would wrap; the cut sits where the accuracy guarantee ends).  Accurate for 2^-100 <= mx < 2^100.    */
#define NAE_ATAN_C0 1.591543257e-01f
#define NAE_ATAN_C1 -5.302623659e-02f
#define NAE_ATAN_C2 3.152511641e-02f
#define NAE_ATAN_C3 -2.106151544e-02f
#define NAE_ATAN_C4 1.267249603e-02f
#define NAE_ATAN_C5 -5.348273553e-03f
#define NAE_ATAN_C6 1.084129326e-03f
#define NAE_ATAN_SCALE 4294967296.0f   /* 2^32: multiplying a coefficient by it is exact */
#define NAE_ATAN_TINY 7.888609052e-31f /* 2^-100 */
#define NAE_ATAN_HUGE 1.267650600e+30f /* 2^100  */
#define NAE_RCP_MAGIC 0x7EF311C7u      /* integer seed of 1/x: relative error < 0.13, three Newton steps -> < 1e-7 */

/* 1/sqrt(2) rounded to f32, used by the 8-point butterflies */
#define NAE_SQRT1_2 0.70710678118654752440f

/* phase vocoder fixed-point formats */
#define NAE_HA_FRAC_BITS 24     /* analysis hop Ha = Hs * tempo, Q39.24                      */
#define NAE_R_FRAC_BITS 24      /* synthesis/analysis hop ratio Hs/d_t, Q8.24                 */
#define NAE_TEMPO_MIN (1.0 / 64.0)   /* Ha >= 4 samples  */
#define NAE_TEMPO_MAX 16.0

/* resampler (rate transposer): 16-tap Kaiser-windowed sinc, 128 phases, linear phase interpolation */
#define NAE_RS_TAPS 16
#define NAE_RS_PHASES 128
#define NAE_RS_KAISER_BETA 8.0
#define NAE_RS_CUTOFF 0.94      /* fraction of the narrower Nyquist */
#define NAE_RATE_MIN (1.0 / 16.0)
#define NAE_RATE_MAX 16.0

/* N2 input resampler: libswresample's defaults when only rates / formats / layouts are set (SURVEY.md Appendix B;
 * /root/reference/src/processor/audio-amix.cpp:217-232 never touches the resampler options): polyphase
 * Kaiser-windowed sinc, filter_size 32, phase_shift 10 (at most 1024 phases, nearest phase, no interpolation between phases),
 * kaiser_beta 9, cutoff 0.97, exact_rational on (a ratio out/in whose reduced numerator is <= 1024 uses exactly that many
 * phases: 160 for 44.1 -> 48 kHz, 320 for 22.05 -> 48, 6 for 8 -> 48, 1 for 96 -> 48; others keep 1024).  Restated from public knowledge of FFmpeg 7.1's resample.c — UNPINNED versus FFmpeg. */
#define NAE_SWR_FILTER_SIZE 32
#define NAE_SWR_PHASE_SHIFT 10
#define NAE_SWR_KAISER_BETA 9.0
#define NAE_SWR_CUTOFF 0.97
#define NAE_SWR_MAX_TAPS 512    /* filter_length = ceil(32 / factor): down-conversion by up to ~15x */

/* x86 "integer indefinite" produced by cvttss2si on overflow/NaN: what the reference's
 * truncating float->int32 gain path yields out of range (audio-vol.cpp:98, int32_t case). */
#define NAE_X86_INT_INDEFINITE (-2147483647 - 1)

#endif
