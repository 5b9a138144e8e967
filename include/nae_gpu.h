/* nae_gpu.h — C-ABI of the MI355X (gfx950) audio-DSP hot path.
 *
 * Drop-in boundary for the per-node sample transforms of Stehsaer/nodey-audio-editor
 * (reference paths below are relative to /root/reference).  Every entry point replaces one CPU inner loop
 * that today runs inside a `processor::*::process_payload` fiber (include/infra/processor.hpp:108-113);
 * the adapter classes in nodey-audio-editor_amd/host/ call these from the same place.
 *
 * Conventions
 *  - plain C, no exceptions cross the ABI: every function returns NAE_OK (0) or a negative nae_status;
 *    nae_last_error(ctx) gives the text.  The adapter converts non-zero into
 *    infra::Processor::Runtime_error (include/infra/processor.hpp:64-77).
 *  - all sample pointers are DEVICE pointers (hipMalloc / nae_malloc) unless the name ends in _host.
 *    "plane pointer arrays" (const T* const*) are HOST arrays whose elements are device pointers — the
 *    same shape as AVFrame::data (audio-vol.cpp:185-186).
 *  - every call only ENQUEUES work on the context's HIP stream and returns; nae_sync() blocks,
 *    nae_poll() never blocks (a Boost fiber polls it and yields: src/infra/runner.cpp:65-83 runs every node
 *    on ONE OS thread, so a blocking wait would stall the whole graph).
 *  - sample formats use FFmpeg's AVSampleFormat numbering so frame->format passes through unchanged.
 */
#ifndef NAE_GPU_H
#define NAE_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  2 (round 5): additions since 1 — nae_event_query, nae_ctx_wait_event, nae_debug_graph4_stages, page-locked memory,
 * nae_wsola_*, nae_swr_* — and one CHANGED behaviour: nae_ctx_create of a second device used to return NAE_ERR_UNSUPPORTED behind a
 * process-wide latch; now a context may sit on any device, and an index outside [0, nae_device_count()) is NAE_ERR_INVALID.
 * A caller built against 1 keeps working (nothing was removed or re-typed); a caller that needs the additions checks
 * nae_abi_version() >= 2.
 * 3 (round 6): addition — nae_debug_set (the tuning / A-B switches, formerly 14 environment variables read at context creation). */
#define NAE_ABI_VERSION 3

typedef enum nae_status {
    NAE_OK = 0,
    NAE_ERR_INVALID = -1,     /* bad argument (null pointer, channel count not 1/2 where the reference rejects it, ...) */
    NAE_ERR_UNSUPPORTED = -2, /* sample format / parameter outside the reference's envelope            */
    NAE_ERR_HIP = -3,         /* HIP runtime error; nae_last_error has hipGetErrorString               */
    NAE_ERR_NOMEM = -4,
    NAE_ERR_STATE = -5        /* call order violation on a stateful handle                            */
} nae_status;

/* == AVSampleFormat (libavutil/samplefmt.h) for the formats audio-vol.cpp:188-244 and
 * audio-velocity.cpp:160-229 accept */
typedef enum nae_fmt {
    NAE_FMT_S16 = 1, NAE_FMT_S32 = 2, NAE_FMT_FLT = 3, NAE_FMT_S16P = 6, NAE_FMT_S32P = 7, NAE_FMT_FLTP = 8
} nae_fmt;

typedef struct nae_ctx nae_ctx;
typedef struct nae_event nae_event;
typedef struct nae_stretch nae_stretch;
typedef struct nae_spectrum nae_spectrum;

/* ------------------------------------------------------------------ context / plumbing */
int nae_abi_version(void);
int nae_device_count(void);
/* Devices.  A process may hold contexts on SEVERAL devices (the reference runs every node of a graph in one process on one
 * thread, src/infra/runner.cpp:65-83,142-154, so the drop-in reaches all GPUs of the node from there): a context remembers
 * its device, and every entry point that takes a context — allocation, launches, copies, memset, event record / wait, nae_sync,
 * nae_poll, nae_free — selects that device for the calling thread when the thread's current HIP device differs, and leaves it
 * selected.  `device` outside [0, nae_device_count()) returns
 * NAE_ERR_INVALID.  Several contexts on one device are fine — each has its own stream, so their work overlaps.
 * (More than one device per process is exercised on one-GPU boxes only, through two contexts on device 0 and the
 * rejection of device 1: N > 1 devices from one process are UNMEASURED on hardware.)
 * Threads.  The library keeps no process-global mutable HOST state.  ONE thread at a time drives a given context and the
 * handles / events created from it; different contexts may be created, driven and destroyed from different threads
 * concurrently.  nae_event_query / nae_poll may be called from the thread that drives the context.
 * One piece of DEVICE-global state exists: `g_cu_arrivals` (csrc/kernels_pvpipe.hip), per-CU arrival counters from which the
 * vocoder pipeline's two workgroups of a CU derive which of them starts with the higher issue priority.  It is shared by all
 * contexts of a device and never reset; two contexts running the pipeline concurrently can hand both workgroups of a CU the
 * same parity — a scheduling hint only: results do not depend on it (tests/test_gpu_multi_ctx.py runs two contexts interleaved
 * bit for bit).  The host mirror's own shared state (context pool, device round-robin) is guarded by a mutex
 * (host/processor/gpu-context.hpp); it keeps every node on ONE device unless $NAE_DEVICES asks for more. */
int nae_ctx_create(int device, nae_ctx** out);
int nae_ctx_destroy(nae_ctx* ctx);
int nae_ctx_set_stream(nae_ctx* ctx, void* hip_stream); /* borrow a hipStream_t (e.g. torch's current stream) */
void* nae_ctx_stream(nae_ctx* ctx);
int nae_sync(nae_ctx* ctx);                             /* hipStreamSynchronize                               */
int nae_poll(nae_ctx* ctx);                             /* 1 = idle, 0 = work pending, <0 = error; never blocks */
const char* nae_last_error(nae_ctx* ctx);
const char* nae_device_name(nae_ctx* ctx);

int nae_malloc(nae_ctx* ctx, size_t bytes, void** dptr);
int nae_free(nae_ctx* ctx, void* dptr);
/* page-locked host memory (hipHostMalloc): copies to and from it are truly asynchronous and run at the full PCIe rate, so
 * uploads, kernels and downloads of consecutive batches can overlap on two contexts / streams */
int nae_malloc_host(nae_ctx* ctx, size_t bytes, void** hptr);
int nae_free_host(nae_ctx* ctx, void* hptr);
int nae_memcpy_h2d(nae_ctx* ctx, void* dst, const void* src_host, size_t bytes); /* async on the ctx stream */
int nae_memcpy_d2h(nae_ctx* ctx, void* dst_host, const void* src, size_t bytes); /* async on the ctx stream */
int nae_memcpy_d2d(nae_ctx* ctx, void* dst, const void* src, size_t bytes);
int nae_memset(nae_ctx* ctx, void* dst, int value, size_t bytes);

int nae_event_create(nae_ctx* ctx, nae_event** ev);
int nae_event_record(nae_ctx* ctx, nae_event* ev);      /* on the ctx stream */
/* 1 = everything enqueued before the last nae_event_record(ev) has completed, 0 = still pending, < 0 = error; never blocks.
 * The wait a node's fiber spins on for ITS OWN batch (include/infra/processor.hpp:108-113 — process_payload of one node — while
 * other nodes' work may still be queued on the same or another context): record behind the batch, poll, yield. */
int nae_event_query(nae_event* ev);
/* makes everything enqueued on ctx's stream AFTER this call wait for the work captured by the last record of `ev`
 * (hipStreamWaitEvent): a device-side dependency between two contexts of the same device, nothing blocks on the host */
int nae_ctx_wait_event(nae_ctx* ctx, nae_event* ev);
int nae_event_elapsed_ms(nae_event* start, nae_event* stop, float* ms); /* synchronises on `stop` */
int nae_event_destroy(nae_event* ev);

/* per-kernel timing: when enabled every kernel launch is bracketed by two hipEvents on the ctx stream.
 * nae_prof_get(index) returns the number of distinct kernels seen; call with index = -1 to get only the count. */
int nae_prof_enable(nae_ctx* ctx, int on);
int nae_prof_reset(nae_ctx* ctx);
int nae_prof_get(nae_ctx* ctx, int index, char* name, size_t name_cap, double* total_ms, uint64_t* launches);

/* shader clock (GHz) the GPU holds at this moment, from s_memtime / s_memrealtime stamps of a 1024-workgroup probe kernel
 * launched on the context's stream: lets a benchmark turn its own kernel times into cycles without assuming a clock. */
int nae_debug_clock_ghz(nae_ctx* ctx, double* ghz);

/* Tuning / A-B switches of one context (tests and measurement tools; the defaults are the product).  Takes effect from the next call on.
 * Unknown key or value out of range: NAE_ERR_INVALID.  Keys (value 0 = the library's own choice unless said otherwise):
 *   pv_tile         frames per phase-vocoder time tile (pass 1 and pass 3);  pv_min_ptile  shortest pass-1 tile the library may choose (default 16)
 *   pv_fps          1 | 2 | 4: frames per step of the vocoder pipeline
 *   pv_flow         0 | 1 | 2: launches of at most one workgroup per CU run the one-barrier schedule never / with one frame per step (default 1) / always
 *   pv_lean         1: the pipeline keeps its 64-VGPR two-workgroups-per-CU build even when one workgroup per CU would allow 128
 *   rs_single, rs_direct, no_mix_fuse      1: transposer with one stream per workgroup / the direct (unstaged) kernel / mix and transposer as two launches
 *   spec_generic, spec_narrow              1: skip the interleaved-stereo spectrum kernel / its dword stores instead of 16-byte ones
 *   spec_chunk, spec_fine, spec_fine_rounds   frames per chunk of the stereo spectrum kernel / of the short chunks at a launch's end / how many of those per wave
 *   td_nc           1 | 2 | 4: candidates per thread of the WSOLA search;  st_unfused  1: filter and cubic stage of the WSOLA chain as two launches
 * The same assignments, comma separated, in the environment variable NAE_DEBUG ("pv_flow=2,pv_fps=4") are applied when a context is created
 * (for measuring a program that creates its contexts itself, e.g. bench.py); an unknown key there fails nae_ctx_create with NAE_ERR_INVALID. */
int nae_debug_set(nae_ctx* ctx, const char* key, long long value);

/* test utility: adds the number of differing 32-bit words of two device buffers to the device counter *d_count (zero it
 * first with nae_memset); asynchronous on the context's stream. */
int nae_debug_diff_u32(nae_ctx* ctx, const void* a, const void* b, size_t n_words, uint64_t* d_count);

/* synthetic input: dst[s*stream_stride + i] = uniform[-1,1) from splitmix64 with
 * seed(s) = 0x9E3779B97F4A7C15*(1 + first_stream + s) + input_index, i < n_per_stream.  Benchmark/test utility. */
int nae_fill_uniform_f32(nae_ctx* ctx, float* dst, size_t n_per_stream, size_t stream_stride, size_t n_streams,
                         uint64_t first_stream, uint64_t input_index);

/* ------------------------------------------------------------------ batched signal view
 * element (stream s, channel c, sample-frame i) lives at base[s*stream_stride + c*chan_stride + i*frame_stride]
 * (strides in ELEMENTS).  interleaved [S][ch]: chan_stride 1, frame_stride ch;  planar [ch][S]: chan_stride S,
 * frame_stride 1.  stream_stride 0 = every stream reads the same buffer (a shared source). */
typedef struct nae_sig {
    void* base;
    size_t stream_stride;
    size_t chan_stride;
    size_t frame_stride;
} nae_sig;

/* ------------------------------------------------------------------ K1 gain
 * replaces change_volume<T>, src/processor/audio-vol.cpp:75-100 (dispatch :188-244).
 * dst[p][i] = T(float(src[p][i]) * volume); f32: one IEEE multiply; s16/s32: truncation toward zero,
 * no clamp, x86 out-of-range result (0x80000000, then modular narrowing for s16).  src == dst allowed. */
int nae_gain_f32(nae_ctx* ctx, const float* const* src_planes, float* const* dst_planes, int planes,
                 size_t elems, float volume);
int nae_gain_s16(nae_ctx* ctx, const int16_t* const* src_planes, int16_t* const* dst_planes, int planes,
                 size_t elems, float volume);
int nae_gain_s32(nae_ctx* ctx, const int32_t* const* src_planes, int32_t* const* dst_planes, int planes,
                 size_t elems, float volume);
/* whole-frame entry mirroring the format switch at audio-vol.cpp:184-244: `planes` has 1 pointer for packed
 * formats and `ch` pointers for planar ones; ch must be 1 or 2 (:177-182) */
int nae_gain_frame(nae_ctx* ctx, int fmt, const void* const* src_planes, void* const* dst_planes, size_t S,
                   int ch, float volume);

/* ------------------------------------------------------------------ K2 split / merge
 * deinterleave replaces swr_convert FLT->FLTP at equal rate/layout (audio-amix.cpp:263-269,
 * audio-bimix.cpp:259-265); interleave replaces audio-velocity.cpp:169-180. Bit copies. */
int nae_deinterleave_f32(nae_ctx* ctx, const float* src, float* const* dst_planes, size_t S, int ch);
int nae_interleave_f32(nae_ctx* ctx, const float* const* src_planes, float* dst, size_t S, int ch);
/* batched: n_streams x ch x S through nae_sig views (configs[1]: 1000 buffers of 4096) */
int nae_copy_sig_f32(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams);
/* fused split -> gain -> merge on interleaved stereo (the three nodes of configs[1] in one pass) */
int nae_gain_sig_f32(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams,
                     float volume);

/* ------------------------------------------------------------------ K3 N-input mix
 * replaces the loop at audio-amix.cpp:293-307: L[j] = (((0 + a0L[j]*v0) + a1L[j]*v1) + ...), same for R;
 * multiply and add are never fused.  1 <= n <= 16 (audio-amix.cpp:342). */
int nae_amix_f32(nae_ctx* ctx, const float* const* inL, const float* const* inR, const float* vol_host, int n,
                 float* outL, float* outR, size_t S);
/* batched over streams; inputs may be interleaved (the node's swr FLT->FLTP step is folded in) or planar */
int nae_amix_sig_f32(nae_ctx* ctx, const nae_sig* inputs, const float* vol_host, int n, const nae_sig* out,
                     size_t S, size_t n_streams);

/* ------------------------------------------------------------------ K4 channel mix v1
 * replaces audio-bimix.cpp:310-317: outL = (ll/2 + lr/2)*(1-bias), outR = (rl/2 + rr/2)*(1+bias) */
int nae_bimix_f32(nae_ctx* ctx, const float* ll, const float* lr, const float* rl, const float* rr, float bias,
                  float* outL, float* outR, size_t S);

/* ------------------------------------------------------------------ K5 channel mix v2
 * downmix replaces audio-bimix.cpp:624-627,717-720 (mono = (l+r)*0.5); interleave replaces :797-803,
 * :833-850 and the single-sided tails :736-742,:759-765 (later == NULL, aligned == 0). */
int nae_bimix2_downmix_f32(nae_ctx* ctx, const float* l, const float* r, float* mono, size_t S);
int nae_bimix2_interleave_f32(nae_ctx* ctx, float* dst, const float* earlier, const float* later,
                              size_t unaligned, size_t aligned, int earlier_offset);

/* ------------------------------------------------------------------ K6 any format -> interleaved f32
 * replaces extract_samples_interleaved, audio-velocity.cpp:150-232 (divisors 32768.0f / 32767 /
 * 2147483648.0f / (double)2147483647 kept literally).  NAE_ERR_UNSUPPORTED where the reference throws. */
int nae_to_f32_interleaved(nae_ctx* ctx, int fmt, const void* const* planes, size_t S, int ch, float* dst);

/* sink-side clamp, audio-io.cpp:617-618 */
int nae_clamp_f32(nae_ctx* ctx, float* data, size_t n);

/* ------------------------------------------------------------------ K7 tempo / pitch
 * replaces the SoundTouch object driven by soundtouch_process_payload, audio-velocity.cpp:265-443
 * (setSampleRate/setChannels/setRate/setPitch :381-385, putSamples :403, numSamples :399,414,419,
 * receiveSamples :298, flush :427).  Algorithm: phase vocoder (N=1024, hop 256) + windowed-sinc rate
 * transposer, DESIGN.md §3 — NOT SoundTouch's WSOLA; parity vs SoundTouch is unpinned.
 * Velocity_modifier passes (rate = velocity, pitch = keep_pitch ? 1/velocity : 1) (:445-460);
 * Pitch_modifier passes (rate = 1, pitch = 2^(semitones/12)) (:462-477). */
typedef struct nae_stretch_plan {
    int pv_on, rs_on;
    double tempo_eff, rate_eff;
    int64_t ha_q24;
    int32_t d0;
    uint32_t r_q24[2];
    uint64_t step_q32;
    size_t out_len, mid_len, frames;
    int rs_first;             /* both stages on and rate_eff > 1: the transposer runs FIRST (fewer samples reach the
                                 vocoder; SoundTouch orders its stages by the same rule) and mid_len is ITS output */
} nae_stretch_plan;
int nae_stretch_plan_make(double rate, double pitch, size_t in_len, nae_stretch_plan* plan);

/* block form: n_streams independent signals of in_len sample-frames each; dst receives plan.out_len frames.
 * Workspace is owned by the context and grown on demand (never freed between calls). */
int nae_stretch_block_f32(nae_ctx* ctx, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                          size_t n_streams, const nae_sig* dst);
/* debugging tap used by the parity tests: synthesis phase (Q0.32) in front of every PV tile,
 * dst_host[n_streams][ch][n_tiles][513] int32; returns n_tiles through *n_tiles and the tile length in frames */
int nae_debug_pv_tile_phase(nae_ctx* ctx, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                            size_t n_streams, int32_t* dst_host, size_t dst_capacity, size_t* n_tiles,
                            size_t* tile_frames);

/* SoundTouch-shaped streaming handle (interleaved f32, device pointers) */
int nae_stretch_create(nae_ctx* ctx, int sample_rate, int channels, float rate, float pitch, nae_stretch** h);
int nae_stretch_put(nae_stretch* h, const float* interleaved, size_t S);
int nae_stretch_put_host(nae_stretch* h, const float* interleaved_host, size_t S);
int nae_stretch_flush(nae_stretch* h);
size_t nae_stretch_available(nae_stretch* h);                          /* == SoundTouch::numSamples()    */
int nae_stretch_receive(nae_stretch* h, float* dst, size_t max_frames, size_t* got);
int nae_stretch_receive_host(nae_stretch* h, float* dst_host, size_t max_frames, size_t* got);
int nae_stretch_destroy(nae_stretch* h);

/* ------------------------------------------------------------------ K7 option A (N1): SoundTouch-shaped time-domain chain
 * The same call sites as above (audio-velocity.cpp:369-385 create + setSampleRate/setChannels/setRate/setPitch,
 * :403 putSamples, :399 numSamples, :298 receiveSamples, :427 flush), with SoundTouch 2.3.2's own algorithm instead
 * of the phase vocoder: WSOLA stretcher (8 ms overlap, automatic 40-90 ms sequence / 15-20 ms seek window,
 * exhaustive normalised cross-correlation search), 64-tap anti-alias FIR and 4-point cubic transposer, in the
 * library's stage order (stretcher first for rate > 1, transposer first otherwise) and with its flush rule
 * (zero blocks of 128 frames until round(in / rate) frames exist, at most 200 of them).  The library is not in the
 * reference tree, so the algorithm is restated and parity versus SoundTouch itself is UNPINNED;
 * the GPU result is bit-exact versus oracle/orc_wsola.c.  `rate`, `pitch` as for nae_stretch_create. */
typedef struct nae_wsola nae_wsola;
typedef struct nae_wsola_plan {
    int sample_rate, channels;
    double rate_eff, tempo_eff;      /* rate*pitch, 1/pitch */
    int order;                       /* 0: stretcher, filter, transposer (rate_eff > 1)   1: filter, transposer,
                                        stretcher (== 1)   2: transposer, filter, stretcher (< 1) */
    int overlap_len, seq_len, seek_len, sample_req;
    double nominal_skip;
    size_t in_len, flush_zeros;      /* zeros the flush rule appends */
    size_t n_seq;                    /* WSOLA sequences */
    size_t td_out_len, aa_out_len, cu_out_len;
    size_t out_len;                  /* frames delivered for in_len frames of input */
} nae_wsola_plan;
int nae_wsola_plan_make(int sample_rate, int channels, double rate, double pitch, size_t in_len, nae_wsola_plan* plan);
/* block form: n_streams independent signals of in_len frames; dst receives plan.out_len frames per stream (one put,
 * flush, receive-all).  offsets_dbg (device, optional): [n_streams][n_seq - 1] chosen overlap offsets. */
int nae_wsola_block_f32(nae_ctx* ctx, int sample_rate, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, const nae_sig* dst, int32_t* offsets_dbg);
int nae_wsola_create(nae_ctx* ctx, int sample_rate, int channels, double rate, double pitch, nae_wsola** h);
int nae_wsola_put(nae_wsola* h, const float* interleaved, size_t S);
int nae_wsola_put_host(nae_wsola* h, const float* interleaved_host, size_t S);
int nae_wsola_flush(nae_wsola* h);
size_t nae_wsola_available(const nae_wsola* h);
int nae_wsola_receive(nae_wsola* h, float* dst, size_t max_frames, size_t* got);
int nae_wsola_receive_host(nae_wsola* h, float* dst_host, size_t max_frames, size_t* got);
int nae_wsola_destroy(nae_wsola* h);

/* ------------------------------------------------------------------ N2 input conversion (libswresample's role)
 * replaces the SwrContext every mixer input runs through (audio-amix.cpp:212-240,263-282; audio-bimix.cpp:198-240,
 * 259-294; utility/sw-resample.hpp:55-70): any supported format / mono|stereo / rate -> `out_rate` stereo f32.
 * Identity inputs (same rate, stereo float) are a bit copy — the only case the reference pins.  Everything else restates
 * libswresample's defaults and is UNPINNED versus FFmpeg itself: sample scaling as K6, mono -> stereo as L = R = m/sqrt(2)
 * (swr's default float rematrix), rate change by swr's default polyphase resampler (filter_size 32, at most 1024 phases — exact_rational: 160 for 44.1 -> 48 kHz —, nearest
 * phase, Kaiser beta 9, cutoff 0.97, reflected ends; include/nae_dsp_spec.h, oracle/orc_swr.c).  Ratios beyond ~15x down
 * are refused with NAE_ERR_UNSUPPORTED.
 * convert() follows swr_convert(ctx, out, out_count, in, in_count): it consumes all n_in input frames, hands out at
 * most max_out frames and keeps the rest buffered; planes == NULL drains (the flush idiom at audio-amix.cpp:281-282). */
typedef struct nae_swr nae_swr;
int nae_swr_create(nae_ctx* ctx, int in_fmt, int in_rate, int in_channels, int out_rate, nae_swr** h);
int nae_swr_convert_host(nae_swr* h, const void* const* planes_host, size_t n_in, float* outL_host, float* outR_host,
                         size_t max_out, size_t* n_out);
/* The same with DEVICE output planes and nothing waited for: the upload, the conversion and the two planes are queued on
 * the context's stream and *n_out (known from frame counts alone) is returned at once.  The caller keeps planes_host alive
 * until it has waited for the stream (nae_sync / nae_poll).  For hosts that queue several frames — or several inputs of a
 * mixer — behind one wait; results are those of nae_swr_convert_host. */
int nae_swr_convert(nae_swr* h, const void* const* planes_host, size_t n_in, float* outL_dev, float* outR_dev,
                    size_t max_out, size_t* n_out);
size_t nae_swr_buffered(nae_swr* h);   /* output frames ready without more input */
int nae_swr_destroy(nae_swr* h);
/* mono -> interleaved stereo with gain (the rematrix step above), device pointers */
int nae_mono_to_stereo_f32(nae_ctx* ctx, const float* mono, float* dst_interleaved, size_t S, float gain);

/* ------------------------------------------------------------------ K8 FFT spectrum
 * no reference code (FFTW declared at xmake.lua:15,33, never called).  Spec: per channel, periodic-Hann
 * windowed 1024-point forward r2c DFT every 256 sample-frames, un-normalised (FFTW convention),
 * |X[k]| for k = 0..512.  frames = T < 1024 ? 0 : (T-1024)/256 + 1.
 * dst element (s, frame f, channel c, bin k) at dst_base[s*dst_stream_stride + (f*ch + c)*513 + k]. */
size_t nae_spectrum_frames(size_t T);
int nae_spectrum_block_f32(nae_ctx* ctx, const nae_sig* src, size_t T, int ch, size_t n_streams, float* dst,
                           size_t dst_stream_stride);
/* streaming handle: put interleaved samples, receive whole frames [ch][513] */
int nae_spectrum_create(nae_ctx* ctx, int n_fft, int hop, int channels, nae_spectrum** h);
int nae_spectrum_put(nae_spectrum* h, const float* interleaved, size_t S);
size_t nae_spectrum_available(nae_spectrum* h);                         /* whole frames ready */
int nae_spectrum_receive(nae_spectrum* h, float* dst, size_t max_frames, size_t* got);
int nae_spectrum_destroy(nae_spectrum* h);

/* ------------------------------------------------------------------ the 4-node graph of BASELINE.json
 * input -> mix(2) -> pitch -> FFT spectrum, one launch sequence over n_streams independent streams.
 * Each node still materialises its output in HBM (the plugin contract: one Audio_stream per link,
 * src/infra/runner.cpp:35-50), so the intermediate buffers are caller-provided. */
typedef struct nae_graph4 {
    nae_sig in_a, in_b;       /* two mix inputs, interleaved stereo f32; in_b.stream_stride may be 0 (shared) */
    float vol_a, vol_b;
    nae_sig mix_out;          /* planar stereo (the amix node emits FLTP, audio-amix.cpp:198)                 */
    double rate, pitch;       /* pitch node parameters                                                         */
    nae_sig pitch_out;        /* interleaved stereo (construct_audio_frame_float emits FLT, audio-velocity.cpp:247) */
    float* spec_out;          /* [n_streams][frames][2][513]                                                  */
    size_t spec_stream_stride;
    size_t S;                 /* sample-frames per stream                                                      */
    size_t n_streams;
} nae_graph4;
int nae_graph4_run(nae_ctx* ctx, const nae_graph4* g);
/* the same graph, only the stages selected by `mask`: 1 = mix node (+ the pitch node's transposer when it runs first),
 * 2 = the rest of the pitch node, 4 = spectrum node.  Scheduling experiments and tests; 7 == nae_graph4_run on one stream. */
int nae_debug_graph4_stages(nae_ctx* ctx, const nae_graph4* g, int mask);

#ifdef __cplusplus
}
#endif
#endif
