// nae_internal.h — shared between the translation units of libnae_gpu.so (not installed)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "../../include/nae_gpu.h"
#include "../../include/nae_dsp_spec.h"

namespace nae { struct cf; }

struct nae_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};
    char name[256] = {0};
    int n_cu = 256;              // compute units of the device (launch-shape decisions)
    // read-only tables (built on the host in double, rounded once to f32)
    nae::cf* d_w512 = nullptr;   // exp(-2 pi i k/512),  k = 0..511
    nae::cf* d_t1024 = nullptr;  // exp(-2 pi i k/1024), k = 0..512
    float* d_hann = nullptr;     // periodic Hann, 1024
    unsigned* d_spec_ctr = nullptr;  // work counter of the persistent stereo spectrum kernel (zeroed on the stream in front of every drawing launch; kernels_stft.hip)
    // grow-only workspaces
    void* ws_phase = nullptr; size_t ws_phase_bytes = 0;
    void* ws_mid = nullptr;   size_t ws_mid_bytes = 0;
    float* d_rs_tab = nullptr; double rs_tab_rate = 0.0;
    std::vector<float> h_rs_tab;
    struct nae_wsola_cache* wsola_cache = nullptr;   // plan + workspaces of nae_wsola_block_f32 (nae_wsola.hip)
    int pv_tile = 0;             // frames per phase-vocoder tile; 0 = choose per call (nae_pick_pv_shape)
    int pv_fps = 0;              // pv_fps = 1|2|4: frames per step of the vocoder pipeline (0 = choose per call)
    // tuning / A-B switches: nae_debug_set(ctx, key, value) (include/nae_gpu.h lists the keys; NAE_DEBUG="key=value,..." applies them at context creation)
    bool dbg_st_unfused = false;     // st_unfused: WSOLA chain runs filter and cubic stage as separate launches
    int dbg_td_nc = 0;               // td_nc = 1|2|4: candidates per thread of the WSOLA search (0: by batch size)
    bool dbg_no_mix_fuse = false;    // no_mix_fuse: graph4 runs mix and transposer as separate launches
    bool dbg_rs_single = false;      // rs_single: one stream per transposer workgroup (no coefficient sharing)
    bool dbg_rs_direct = false;      // rs_direct: direct (unstaged) transposer kernel
    bool dbg_spec_generic = false;   // spec_generic: skip the interleaved-stereo spectrum fast path
    bool dbg_spec_narrow = false;    // spec_narrow: the stereo spectrum kernel stores dword pieces (round 1-4 form) instead of 16-byte ones
    int dbg_spec_fine = 0, dbg_spec_fine_rounds = 0;   // spec_fine / spec_fine_rounds: frames of the short chunks at the end of a large launch's list / how many of them per resident wave
    int dbg_spec_chunk = 0;          // spec_chunk: frames one wave of the stereo spectrum kernel walks (0: spec_pick_chunk)
    int dbg_pv_min_ptile = 0;        // pv_min_ptile: shortest pass-1 tile in frames (0: 16)
    int pv_flow = 1;                 // pv_flow = 0|1|2: launches of at most one workgroup per CU run the one-barrier schedule (pv_flow_kernel) never / with one
                                     // frame per step (default: where it is faster, profiles/r05_flow.md) / in every shape
    bool pv_lean = false;            // pv_lean: the vocoder pipeline keeps its 64-VGPR shape even when one workgroup per CU would allow
                                     // 128 (leaves half of the register file and 94 KB of LDS to a co-resident kernel: tools/coresidency.py)
    // per-context, per-device launch state (a kernel attribute is set once per device: the flag lives with the context's device)
    unsigned pv_attr_done = 0;       // bit per pv_pipe_kernel instantiation whose dynamic-LDS attribute has been set through this context
    // optional per-kernel timing (hipEvent pairs on the ctx stream), used by bench.py for the roofline line
    bool prof_on = false;
    struct ProfSlot { const char* name; double total_ms; uint64_t launches; };
    struct ProfPair { int slot; hipEvent_t a, b; };
    std::vector<ProfSlot> prof_slots;
    std::vector<ProfPair> prof_pairs;
};

// RAII: brackets one kernel launch with two events when profiling is on
struct NaeProfScope {
    nae_ctx* ctx; int idx;
    NaeProfScope(nae_ctx* c, const char* name);
    ~NaeProfScope();
};

// Several devices in one process: a context remembers its device and every entry point that allocates, launches or records
// selects it when the calling thread's current device differs (one hipGetDevice on the fast path).  The thread's current device is
// left on the context's.
static inline int nae_use_device(nae_ctx* ctx)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return 0;
    return hipSetDevice(ctx->device) == hipSuccess ? 0 : -3 /* NAE_ERR_HIP */;
}

#define NAE_KLAUNCH(ctx, name_str, ...)          \
    do {                                          \
        (void)nae_use_device(ctx);                \
        NaeProfScope nae_ps_(ctx, name_str);      \
        hipLaunchKernelGGL(__VA_ARGS__);          \
    } while (0)

struct nae_event { hipEvent_t ev; };

int nae_check(nae_ctx* ctx, hipError_t e, const char* what);
int nae_fail(nae_ctx* ctx, int code, const char* what);
int nae_ws_reserve(nae_ctx* ctx, void** p, size_t* have, size_t want);

// kernels_stft.hip
int nae_launch_spectrum(nae_ctx* ctx, const nae_sig* src, size_t T, int ch, size_t n_streams, float* dst,
                        size_t dst_stream_stride);
// a continued stream processes frames / hop blocks [f_origin, f_origin + f_count) per call
struct nae_pv_segment {
    long long f_origin, f_count;
    long long f_limit;            // frames >= f_limit are not available yet (or do not exist)
    long long mid_limit;          // stretched samples >= mid_limit are not stored
    const uint32_t* carry_in;     // [n_streams*ch][520] phase behind frame f_origin-1 (null: zero)
    uint32_t* carry_out;          // receives the phase behind frame f_origin+f_count-1 (null: not wanted)
    bool carry_by_synth = false;  // the segment is synthesised as ONE tile and pass 3 itself writes carry_out (no pass 1)
};
size_t nae_pv_phase_workspace_bytes(size_t n_frames, int ch, size_t n_streams, int tile);
int nae_launch_pv_phase(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, int tile, int synth_tile, uint32_t* phase_ws, const nae_pv_segment* seg);
int nae_launch_pv_synth(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t in_len, int ch,
                        size_t n_streams, int tile, int phase_tile, const uint32_t* phase_ws, const nae_sig* out,
                        const nae_pv_segment* seg, int frames_per_step);
int nae_launch_resample(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* src, size_t src_len, int ch,
                        size_t n_streams, const float* d_tab, const nae_sig* out, size_t j_begin, size_t j_end);
int nae_launch_mix_resample(nae_ctx* ctx, const nae_stretch_plan* pl, const nae_sig* a, const nae_sig* b, float va, float vb,
                            const nae_sig* mix_out, size_t S, size_t n_streams, const float* d_tab, const nae_sig* out);
int nae_ensure_rs_table(nae_ctx* ctx, double rate_eff);
int nae_pick_pv_shape(nae_ctx* ctx, size_t frames, size_t n_sc, int* phase_tile, int* frames_per_step);
constexpr int kPhasePad = 520; // int32 per (stream-channel, tile) record in the phase workspace

// nae_wsola.hip
void nae_wsola_cache_free(nae_ctx* ctx);

// kernels_nodes.hip
int nae_launch_copy_sig(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams,
                        bool scale, float volume);
