// nae_api.hip — context, plumbing, and the host-side orchestration of K7/K8 behind the C ABI (include/nae_gpu.h).
// No CPU compute path exists in this library: every transform is a HIP kernel launch; if HIP is unusable the
// entry points fail with NAE_ERR_HIP.
#include "nae_internal.h"
#include <string>
#include <initializer_list>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

namespace nae { struct cf { float x, y; }; }

int nae_fail(nae_ctx* ctx, int code, const char* what)
{
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s", what);
    return code;
}

int nae_check(nae_ctx* ctx, hipError_t e, const char* what)
{
    if (e == hipSuccess) return NAE_OK;
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
    return NAE_ERR_HIP;
}

NaeProfScope::NaeProfScope(nae_ctx* c, const char* name) : ctx(c), idx(-1)
{
    if (!c || !c->prof_on) return;
    int slot = -1;
    for (size_t i = 0; i < c->prof_slots.size(); i++)
        if (c->prof_slots[i].name == name || !strcmp(c->prof_slots[i].name, name)) { slot = (int)i; break; }
    if (slot < 0) { c->prof_slots.push_back({name, 0.0, 0}); slot = (int)c->prof_slots.size() - 1; }
    nae_ctx::ProfPair pp{slot, nullptr, nullptr};
    if (hipEventCreate(&pp.a) != hipSuccess || hipEventCreate(&pp.b) != hipSuccess) return;
    (void)hipEventRecord(pp.a, c->stream);
    c->prof_pairs.push_back(pp);
    idx = (int)c->prof_pairs.size() - 1;
}
NaeProfScope::~NaeProfScope()
{
    if (idx >= 0) (void)hipEventRecord(ctx->prof_pairs[idx].b, ctx->stream);
}

static void prof_collect(nae_ctx* ctx)
{
    if (ctx->prof_pairs.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& pp : ctx->prof_pairs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) == hipSuccess) {
            ctx->prof_slots[pp.slot].total_ms += ms;
            ctx->prof_slots[pp.slot].launches += 1;
        }
        (void)hipEventDestroy(pp.a);
        (void)hipEventDestroy(pp.b);
    }
    ctx->prof_pairs.clear();
}

int nae_ws_reserve(nae_ctx* ctx, void** p, size_t* have, size_t want)
{
    if (*have >= want && *p) return NAE_OK;
    (void)nae_use_device(ctx);
    if (*p) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return nae_check(ctx, e, "hipStreamSynchronize");
        (void)hipFree(*p);
        *p = nullptr;
        *have = 0;
    }
    size_t bytes = want + want / 8 + 4096;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        (void)nae_check(ctx, e, "hipMalloc(workspace)");
        return NAE_ERR_NOMEM;
    }
    *have = bytes;
    return NAE_OK;
}

static void build_tables(std::vector<nae::cf>& w512, std::vector<nae::cf>& t1024, std::vector<float>& hann)
{
    const double two_pi = 6.283185307179586476925286766559;
    w512.resize(512);
    t1024.resize(513);
    hann.resize(1024);
    for (int k = 0; k < 512; k++) w512[k] = nae::cf{(float)cos(two_pi * k / 512.0), (float)(-sin(two_pi * k / 512.0))};
    for (int k = 0; k <= 512; k++) t1024[k] = nae::cf{(float)cos(two_pi * k / 1024.0), (float)(-sin(two_pi * k / 1024.0))};
    for (int n = 0; n < 1024; n++) hann[n] = (float)(0.5 - 0.5 * cos(two_pi * n / 1024.0));
}

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

static void build_rs_table(double rate_eff, std::vector<float>& tab)
{
    const double pi = 3.14159265358979323846;
    const double c = NAE_RS_CUTOFF * (rate_eff > 1.0 ? 1.0 / rate_eff : 1.0);
    const double half = NAE_RS_TAPS / 2.0;
    const double i0b = bessel_i0(NAE_RS_KAISER_BETA);
    tab.resize((NAE_RS_PHASES + 1) * NAE_RS_TAPS);
    for (int p = 0; p <= NAE_RS_PHASES; p++) {
        double row[NAE_RS_TAPS], sum = 0.0;
        for (int i = 0; i < NAE_RS_TAPS; i++) {
            const double x = (double)(i - (NAE_RS_TAPS / 2 - 1)) - (double)p / NAE_RS_PHASES;
            const double a = x / half;
            double w = 0.0;
            if (a > -1.0 && a < 1.0) w = bessel_i0(NAE_RS_KAISER_BETA * sqrt(1.0 - a * a)) / i0b;
            else if (a == 1.0 || a == -1.0) w = 1.0 / i0b;
            const double arg = pi * c * x;
            const double sinc = (fabs(arg) < 1e-12) ? 1.0 : sin(arg) / arg;
            row[i] = c * sinc * w;
            sum += row[i];
        }
        for (int i = 0; i < NAE_RS_TAPS; i++) tab[p * NAE_RS_TAPS + i] = (float)(row[i] / sum);
    }
}

int nae_ensure_rs_table(nae_ctx* ctx, double rate_eff)
{
    if (ctx->d_rs_tab && ctx->rs_tab_rate == rate_eff) return NAE_OK;
    (void)nae_use_device(ctx);
    if (!ctx->d_rs_tab) {
        hipError_t e = hipMalloc((void**)&ctx->d_rs_tab, (NAE_RS_PHASES + 1) * NAE_RS_TAPS * sizeof(float));
        if (e != hipSuccess) return nae_check(ctx, e, "hipMalloc(rs table)");
    } else {
        hipError_t e = hipStreamSynchronize(ctx->stream); // the previous table may still be in use
        if (e != hipSuccess) return nae_check(ctx, e, "hipStreamSynchronize");
    }
    build_rs_table(rate_eff, ctx->h_rs_tab);
    hipError_t e = hipMemcpy(ctx->d_rs_tab, ctx->h_rs_tab.data(), ctx->h_rs_tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMemcpy(rs table)");
    ctx->rs_tab_rate = rate_eff;
    return NAE_OK;
}

// Shape of the phase vocoder for a block call: frames per step of the pipeline (kernels_pvpipe.hip), synthesis tile, pass-1 tile.
//   * more than 3 stream-channels per CU (>= 1024 at 256 CUs, e.g. 512 stereo streams): four stream-channels per workgroup
//     (frames_per_step 1), ONE tile per stream-channel — no pass 1, nothing analysed twice; from 2048 stream-channels two
//     workgroups share a CU; up to 1024 stream-channels (one workgroup per CU) the launch runs the one-barrier schedule (pv_flow_kernel).
//     (2 to 3 per CU: the frame-interleaved shape below in three rounds, see the code.)
//   * fewer: the four slots of a workgroup work on 2 or 4 consecutive frames of one stream-channel (frame-interleaved), so
//     256 stream-channels (the 128 streams one rank of an 8-GPU job owns) still give every CU a workgroup without cutting
//     a stream into time tiles.  A workgroup of these modes fills a CU's LDS alone: 4 frames per step up to n_cu
//     stream-channels, 2 frames per step up to 2 n_cu.
//   * fewer stream-channels than CUs: time tiles on top (each tile but the last of a stream-channel is analysed twice: pass 1
//     sums its phase increments, pass 3 synthesises it), just enough of them for one workgroup per CU.  Pass 1 is one wave per
//     tile and wants 24 n_cu waves (six per SIMD): it runs on tiles `step` times shorter (*phase_tile), never shorter than 64 frames; the
//     synthesis tile is a multiple of it.
int nae_pick_pv_shape(nae_ctx* ctx, size_t frames, size_t n_sc, int* phase_tile, int* frames_per_step)
{
    const size_t n_cu = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    int fps = n_sc > 2 * n_cu ? 1 : n_sc > n_cu ? 2 : 4;
    // between 2 and 3 stream-channels per CU four stream-channels per workgroup leave a quarter to a half of the CUs without a
    // workgroup; three rounds of four-frames-per-step workgroups are faster there (profiles/r04_shape_sweep.md: 768 stream-channels
    // 3.28 against 3.65 ms, 520: 3.21 against 3.35; at 1024 stream-channels one full round of the first shape wins again)
    if (n_sc > 2 * n_cu && n_sc <= 3 * n_cu) fps = 4;
    if (ctx->pv_fps == 1 || ctx->pv_fps == 2 || ctx->pv_fps == 4) fps = ctx->pv_fps;
    *frames_per_step = fps;
    if (ctx->pv_tile > 0) { *phase_tile = ctx->pv_tile; return ctx->pv_tile; }
    *phase_tile = 64;
    if (frames == 0 || n_sc == 0) return 64;
    // Few LONG stream-channels — at most half as many as CUs, where the frame-interleaved shape below needs time tiles and pass 1 anyway: cut them into as
    // many tiles as the FULL-batch shape wants — one frame per step, 8 n_cu (stream-channel, tile) items = two workgroups per CU at eight waves per SIMD, 1.37x
    // the frames per second of the frame-interleaved shape — when a tile keeps >= 128 frames (its 4 priming / tail frames then cost <= 3 %).  One hour of stereo
    // (BASELINE configs[2]): 1023 tiles of 660 frames per channel.  Pass 1 runs on thirds of a tile (6138 waves: six per SIMD).  Never more than 8 n_cu items:
    // one workgroup beyond two per CU would run alone in a second round.  Between n_cu / 2 and n_cu stream-channels the frame-interleaved shape runs ONE tile
    // per stream-channel (no pass 1) and wins: 96 / 112 / 127 streams of 10 s 1.53 / 1.64 / 1.75 ms per graph step against 1.55 / 1.74 / 2.00 with tiles;
    // 40 and 80 streams tie (profiles/r06_shape_sweep.md).
    if (ctx->pv_fps == 0 && 2 * n_sc <= n_cu) {
        const size_t want = 8 * n_cu / n_sc;                          // tiles per stream-channel, rounded down
        if (want >= 2 && frames >= 128 * want) {
            size_t third = ((frames + want - 1) / want + 2) / 3;          // pass-1 tile: a third of a synthesis tile (6 n_cu x 4 waves), >= 43 frames
            if (third < 64) third = 64;
            *frames_per_step = 1;
            *phase_tile = (int)third;
            return (int)(3 * third);
        }
    }
    // a synthesis tile pays 4 priming / tail frames and keeps >= 64; a pass-1 tile pays one priming frame and may be as short as kMinPhase frames, so that
    // pass 1 still gets its 24 n_cu waves on short streams (40 streams of 10 s: 6160 waves of 29 frames instead of 2800 of 64: profiles/r06_pass1.md)
    const size_t kMinPhase = ctx->dbg_pv_min_ptile > 0 ? (size_t)ctx->dbg_pv_min_ptile : 16;
    const size_t max_synth = (frames + 63) / 64, max_phase = (frames + kMinPhase - 1) / kMinPhase;
    // workgroups wanted: one per CU (two from 2048 stream-channels, where no tiles are needed anyway)
    const size_t wg = (n_sc * (size_t)fps + 3) / 4;                      // workgroups of one tile per stream-channel
    size_t n_synth = wg >= n_cu ? 1 : n_cu / wg;
    if (n_synth > max_synth) n_synth = max_synth;
    size_t n_phase = n_synth == 1 ? 1 : (24 * n_cu + n_sc - 1) / n_sc;   // a single synthesis tile needs no pass 1; else six waves per SIMD
    if (n_phase > max_phase) n_phase = max_phase;
    size_t step = (n_phase + n_synth - 1) / n_synth;
    if (step < 1) step = 1;
    size_t pt = (frames + n_synth * step - 1) / (n_synth * step);
    if (pt < kMinPhase) pt = kMinPhase;
    if (pt * step < 64) pt = (64 + step - 1) / step;
    *phase_tile = (int)pt;
    return (int)(pt * step);
}

extern "C" {

int nae_abi_version(void) { return NAE_ABI_VERSION; }

int nae_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Several GPUs per process: the reference runs every node of a graph in ONE process on ONE thread
// (/root/reference/src/infra/runner.cpp:65-83,142-154), so the drop-in must reach all devices of the node from there.  A context
// remembers its device; allocations and launches go to the calling thread's CURRENT device in HIP, so every entry point selects
// the context's device first when it differs (nae_use_device: one hipGetDevice on the fast path) and leaves it selected.
// No process-global state: launch attributes are tracked per context, so contexts may be created and driven from different
// threads as long as ONE thread at a time drives a given context and its handles (include/nae_gpu.h, "Threads").
int nae_debug_set(nae_ctx* ctx, const char* key, long long value)
{
    if (!ctx || !key) return NAE_ERR_INVALID;
    const std::string k(key);
    const bool flag = value == 0 || value == 1;
    const bool count = value >= 0 && value <= 0x7fffffffll;
    auto one_of = [&](std::initializer_list<long long> ok) { for (long long v : ok) if (v == value) return true; return false; };
    if (k == "pv_tile" && count) ctx->pv_tile = (int)value;
    else if (k == "pv_fps" && one_of({0, 1, 2, 4})) ctx->pv_fps = (int)value;
    else if (k == "pv_flow" && one_of({0, 1, 2})) ctx->pv_flow = (int)value;
    else if (k == "pv_lean" && flag) ctx->pv_lean = value != 0;
    else if (k == "pv_min_ptile" && count) ctx->dbg_pv_min_ptile = (int)value;
    else if (k == "rs_single" && flag) ctx->dbg_rs_single = value != 0;
    else if (k == "rs_direct" && flag) ctx->dbg_rs_direct = value != 0;
    else if (k == "no_mix_fuse" && flag) ctx->dbg_no_mix_fuse = value != 0;
    else if (k == "spec_generic" && flag) ctx->dbg_spec_generic = value != 0;
    else if (k == "spec_narrow" && flag) ctx->dbg_spec_narrow = value != 0;
    else if (k == "spec_chunk" && count) ctx->dbg_spec_chunk = (int)value;
    else if (k == "spec_fine" && count) ctx->dbg_spec_fine = (int)value;
    else if (k == "spec_fine_rounds" && count) ctx->dbg_spec_fine_rounds = (int)value;
    else if (k == "td_nc" && one_of({0, 1, 2, 4})) ctx->dbg_td_nc = (int)value;
    else if (k == "st_unfused" && flag) ctx->dbg_st_unfused = value != 0;
    else return nae_fail(ctx, NAE_ERR_INVALID, "nae_debug_set: unknown key or value out of range");
    return NAE_OK;
}

int nae_ctx_create(int device, nae_ctx** out)
{
    if (!out) return NAE_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NAE_ERR_HIP;
    if (device < 0 || device >= n) return NAE_ERR_INVALID;          // no such device
    if (hipSetDevice(device) != hipSuccess) return NAE_ERR_HIP;
    nae_ctx* ctx = new (std::nothrow) nae_ctx();
    if (!ctx) return NAE_ERR_NOMEM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
        if (prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return NAE_ERR_HIP; }
    ctx->own_stream = true;
    if (const char* e = getenv("NAE_DEBUG")) {
        // "key=value,key=value": the one environment hook of the switches of nae_debug_set
        std::string all(e);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (!kv.empty() && (eq == std::string::npos || nae_debug_set(ctx, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)) != NAE_OK)) {
                fprintf(stderr, "libnae_gpu: NAE_DEBUG: bad assignment '%s'\n", kv.c_str());
                (void)hipStreamDestroy(ctx->stream);
                delete ctx;
                return NAE_ERR_INVALID;
            }
            pos = end + 1;
        }
    }
    std::vector<nae::cf> w512, t1024;
    std::vector<float> hann;
    build_tables(w512, t1024, hann);
    bool ok = hipMalloc((void**)&ctx->d_w512, 512 * sizeof(nae::cf)) == hipSuccess &&
              hipMalloc((void**)&ctx->d_t1024, 520 * sizeof(nae::cf)) == hipSuccess &&
              hipMalloc((void**)&ctx->d_hann, 1024 * sizeof(float)) == hipSuccess &&
              hipMalloc((void**)&ctx->d_spec_ctr, 64) == hipSuccess && hipMemset(ctx->d_spec_ctr, 0, 64) == hipSuccess;
    ok = ok && hipMemcpy(ctx->d_w512, w512.data(), 512 * sizeof(nae::cf), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(ctx->d_t1024, t1024.data(), 513 * sizeof(nae::cf), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(ctx->d_hann, hann.data(), 1024 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { nae_ctx_destroy(ctx); return NAE_ERR_HIP; }
    *out = ctx;
    return NAE_OK;
}

int nae_ctx_destroy(nae_ctx* ctx)
{
    if (!ctx) return NAE_OK;
    (void)nae_use_device(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    prof_collect(ctx);
    if (ctx->d_w512) (void)hipFree(ctx->d_w512);
    if (ctx->d_t1024) (void)hipFree(ctx->d_t1024);
    if (ctx->d_hann) (void)hipFree(ctx->d_hann);
    if (ctx->d_spec_ctr) (void)hipFree(ctx->d_spec_ctr);
    if (ctx->d_rs_tab) (void)hipFree(ctx->d_rs_tab);
    if (ctx->ws_phase) (void)hipFree(ctx->ws_phase);
    if (ctx->ws_mid) (void)hipFree(ctx->ws_mid);
    nae_wsola_cache_free(ctx);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NAE_OK;
}

int nae_ctx_set_stream(nae_ctx* ctx, void* hip_stream)
{
    if (!ctx) return NAE_ERR_INVALID;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return NAE_OK;
}

void* nae_ctx_stream(nae_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int nae_sync(nae_ctx* ctx)
{
    if (!ctx) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int nae_poll(nae_ctx* ctx)
{
    if (!ctx) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return nae_check(ctx, e, "hipStreamQuery");
}

const char* nae_last_error(nae_ctx* ctx) { return ctx ? ctx->err : "null context"; }
const char* nae_device_name(nae_ctx* ctx) { return ctx ? ctx->name : ""; }

int nae_malloc(nae_ctx* ctx, size_t bytes, void** dptr)
{
    if (!ctx || !dptr) return NAE_ERR_INVALID;
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    (void)nae_use_device(ctx);
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) { nae_check(ctx, e, "hipMalloc"); return NAE_ERR_NOMEM; }
    return NAE_OK;
}
int nae_free(nae_ctx* ctx, void* dptr)
{
    if (!dptr) return NAE_OK;
    if (ctx) (void)nae_use_device(ctx);
    return nae_check(ctx, hipFree(dptr), "hipFree");
}
int nae_malloc_host(nae_ctx* ctx, size_t bytes, void** hptr)
{
    if (!ctx || !hptr) return NAE_ERR_INVALID;
    *hptr = nullptr;
    if (bytes == 0) bytes = 16;
    (void)nae_use_device(ctx);
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { nae_check(ctx, e, "hipHostMalloc"); return NAE_ERR_NOMEM; }
    return NAE_OK;
}
int nae_free_host(nae_ctx* ctx, void* hptr)
{
    if (!hptr) return NAE_OK;
    if (ctx) (void)nae_use_device(ctx);
    return nae_check(ctx, hipHostFree(hptr), "hipHostFree");
}
int nae_memcpy_h2d(nae_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(h2d)");
}
int nae_memcpy_d2h(nae_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(d2h)");
}
int nae_memcpy_d2d(nae_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync(d2d)");
}
int nae_memset(nae_ctx* ctx, void* dst, int value, size_t bytes)
{
    if (!ctx || (bytes && !dst)) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream), "hipMemsetAsync");
}

int nae_event_create(nae_ctx* ctx, nae_event** ev)
{
    if (!ctx || !ev) return NAE_ERR_INVALID;
    nae_event* e = new (std::nothrow) nae_event();
    if (!e) return NAE_ERR_NOMEM;
    (void)nae_use_device(ctx);
    if (hipEventCreate(&e->ev) != hipSuccess) { delete e; return NAE_ERR_HIP; }
    *ev = e;
    return NAE_OK;
}
int nae_event_record(nae_ctx* ctx, nae_event* ev)
{
    if (!ctx || !ev) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipEventRecord(ev->ev, ctx->stream), "hipEventRecord");
}
int nae_event_query(nae_event* ev)
{
    if (!ev) return NAE_ERR_INVALID;
    const hipError_t e = hipEventQuery(ev->ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return NAE_ERR_HIP;
}
int nae_ctx_wait_event(nae_ctx* ctx, nae_event* ev)
{
    if (!ctx || !ev) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return nae_check(ctx, hipStreamWaitEvent(ctx->stream, ev->ev, 0), "hipStreamWaitEvent");
}
int nae_event_elapsed_ms(nae_event* start, nae_event* stop, float* ms)
{
    if (!start || !stop || !ms) return NAE_ERR_INVALID;
    if (hipEventSynchronize(stop->ev) != hipSuccess) return NAE_ERR_HIP;
    return hipEventElapsedTime(ms, start->ev, stop->ev) == hipSuccess ? NAE_OK : NAE_ERR_HIP;
}
int nae_event_destroy(nae_event* ev)
{
    if (!ev) return NAE_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return NAE_OK;
}

int nae_prof_enable(nae_ctx* ctx, int on)
{
    if (!ctx) return NAE_ERR_INVALID;
    prof_collect(ctx);
    ctx->prof_on = on != 0;
    return NAE_OK;
}
int nae_prof_reset(nae_ctx* ctx)
{
    if (!ctx) return NAE_ERR_INVALID;
    prof_collect(ctx);
    ctx->prof_slots.clear();
    return NAE_OK;
}
int nae_prof_get(nae_ctx* ctx, int index, char* name, size_t name_cap, double* total_ms, uint64_t* launches)
{
    if (!ctx) return NAE_ERR_INVALID;
    prof_collect(ctx);
    if (index < 0 || index >= (int)ctx->prof_slots.size()) return (int)ctx->prof_slots.size();
    if (name && name_cap) snprintf(name, name_cap, "%s", ctx->prof_slots[index].name);
    if (total_ms) *total_ms = ctx->prof_slots[index].total_ms;
    if (launches) *launches = ctx->prof_slots[index].launches;
    return (int)ctx->prof_slots.size();
}

// ------------------------------------------------------------------------------------------------ K7 plan
// Parameter resolution, DESIGN.md §3.2.  SoundTouch mapping (audio-velocity.cpp:384-385): setRate(rate) and
// setPitch(pitch) give a time-stretch of 1/pitch followed by a resampling of rate*pitch.
int nae_stretch_plan_make(double rate, double pitch, size_t in_len, nae_stretch_plan* pl)
{
    if (!pl) return NAE_ERR_INVALID;
    memset(pl, 0, sizeof *pl);
    if (!(rate > 0.0) || !(pitch > 0.0)) return NAE_ERR_INVALID;
    double tempo = 1.0 / pitch, rho = rate * pitch;
    if (fabs(tempo - 1.0) < 1e-6) tempo = 1.0;
    if (fabs(rho - 1.0) < 1e-6) rho = 1.0;
    pl->pv_on = tempo != 1.0;
    pl->rs_on = rho != 1.0;
    if (pl->pv_on && (tempo < NAE_TEMPO_MIN || tempo > NAE_TEMPO_MAX)) return NAE_ERR_UNSUPPORTED;
    if (pl->rs_on && (rho < NAE_RATE_MIN || rho > NAE_RATE_MAX)) return NAE_ERR_UNSUPPORTED;
    pl->tempo_eff = tempo;
    pl->rate_eff = rho;
    pl->ha_q24 = (int64_t)llround((double)NAE_HOP * tempo * (double)(1 << NAE_HA_FRAC_BITS));
    pl->d0 = (int32_t)(pl->ha_q24 >> NAE_HA_FRAC_BITS);
    for (int i = 0; i < 2; i++) {
        const uint64_t d = (uint64_t)(pl->d0 + i);
        pl->r_q24[i] = (uint32_t)((((uint64_t)NAE_HOP << NAE_R_FRAC_BITS) + d / 2) / d);
    }
    pl->step_q32 = (uint64_t)llround(rho * 4294967296.0);
    pl->out_len = (size_t)floor((double)in_len / (tempo * rho) + 0.5);
    pl->rs_first = pl->pv_on && pl->rs_on && rho > 1.0;
    size_t pv_out;
    if (pl->rs_first) {
        pl->mid_len = (size_t)floor((double)in_len / rho + 0.5);
        pv_out = pl->out_len;
    } else if (pl->rs_on) {
        if (pl->out_len == 0) pl->mid_len = 0;
        else {
            const unsigned __int128 pos = (unsigned __int128)(pl->out_len - 1) * pl->step_q32;
            pl->mid_len = (size_t)(pos >> 32) + NAE_RS_TAPS / 2 + 1;
        }
        pv_out = pl->mid_len;
    } else {
        pl->mid_len = pl->out_len;
        pv_out = pl->out_len;
    }
    pl->frames = pl->pv_on ? (pv_out + NAE_FFT_N / 2 + NAE_HOP - 1) / NAE_HOP + 1 : 0;
    return NAE_OK;
}

static int check_sig(nae_ctx* ctx, const nae_sig* s, const char* what)
{
    if (!s || !s->base) return nae_fail(ctx, NAE_ERR_INVALID, what);
    return NAE_OK;
}

} // extern "C"

// a 2-input mix node in front of the stretch node (graph4): when the transposer runs first it can mix while staging
struct nae_mix_front {
    const nae_sig* a;
    const nae_sig* b;
    float va, vb;
};

// stages: bit 0 = the front stage (the mix node, and the transposer when it runs first), bit 1 = everything behind it
// (nae_debug_graph4_stages: scheduling experiments run the two from separate calls; the intermediate signal stays in the
// context's workspace between them).
static int stretch_block_impl(nae_ctx* ctx, double rate, double pitch, const nae_sig* src, size_t in_len, int ch, size_t n_streams,
                              const nae_sig* dst, const nae_mix_front* front, int stages = 3)
{
    if (!ctx) return NAE_ERR_INVALID;
    int rc;
    if ((rc = check_sig(ctx, src, "null source view")) || (rc = check_sig(ctx, dst, "null destination view"))) return rc;
    if (ch < 1 || ch > 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    nae_stretch_plan pl;
    rc = nae_stretch_plan_make(rate, pitch, in_len, &pl);
    if (rc) return nae_fail(ctx, rc, "rate/pitch outside the supported range");
    // the mix node in front: fused into the transposer when that runs first, else its own launch (src = its output)
    bool mix_pending = front != nullptr && (stages & 1);
    auto run_mix = [&]() -> int {
        if (!mix_pending) return NAE_OK;
        mix_pending = false;
        const nae_sig ins[2] = {*front->a, *front->b};
        const float vol[2] = {front->va, front->vb};
        return nae_amix_sig_f32(ctx, ins, vol, 2, src, in_len, n_streams);
    };
    if (n_streams == 0 || pl.out_len == 0) return run_mix();
    if (!pl.pv_on && !pl.rs_on) {
        if ((rc = run_mix())) return rc;
        return (stages & 2) ? nae_launch_copy_sig(ctx, src, dst, in_len, ch, n_streams, false, 1.0f) : NAE_OK;
    }

    const size_t mid_stride = (pl.mid_len + 3) & ~(size_t)3;
    nae_sig mid{};
    if (pl.pv_on && pl.rs_on) {
        rc = nae_ws_reserve(ctx, &ctx->ws_mid, &ctx->ws_mid_bytes, n_streams * ch * mid_stride * sizeof(float));
        if (rc) return rc;
        mid = nae_sig{ctx->ws_mid, (size_t)ch * mid_stride, mid_stride, 1};   // planar: 16-byte accesses on both sides
    }
    if (pl.rs_on) {
        rc = nae_ensure_rs_table(ctx, pl.rate_eff);
        if (rc) return rc;
    }
    // stage order: transposer first when it shrinks the signal (rate_eff > 1), else vocoder first
    const nae_sig* pv_src = src;
    size_t pv_in_len = in_len;
    const nae_sig* pv_dst = dst;
    long long pv_out_len = (long long)pl.out_len;
    if (pl.rs_first && !(stages & 1)) {
        pv_src = &mid;
        pv_in_len = pl.mid_len;
    } else if (pl.rs_first) {
        rc = 1;
        if (mix_pending && ch == 2) {
            rc = nae_launch_mix_resample(ctx, &pl, front->a, front->b, front->va, front->vb, src, in_len, n_streams, ctx->d_rs_tab, &mid);
            if (rc < 0) return rc;
            if (rc == 0) mix_pending = false;
        }
        if (rc == 1) {
            if ((rc = run_mix())) return rc;
            rc = nae_launch_resample(ctx, &pl, src, in_len, ch, n_streams, ctx->d_rs_tab, &mid, 0, pl.mid_len);
            if (rc) return rc;
        }
        pv_src = &mid;
        pv_in_len = pl.mid_len;
    } else if (pl.pv_on && pl.rs_on) {
        pv_dst = &mid;
        pv_out_len = (long long)pl.mid_len;
    }
    if ((rc = run_mix())) return rc;     // vocoder first / transposer only: the mix is a launch of its own
    if (!(stages & 2)) return NAE_OK;
    if (pl.pv_on) {
        int phase_tile = 0, fps = 1;
        const int tile = nae_pick_pv_shape(ctx, pl.frames, n_streams * ch, &phase_tile, &fps);
        rc = nae_ws_reserve(ctx, &ctx->ws_phase, &ctx->ws_phase_bytes, nae_pv_phase_workspace_bytes(pl.frames, ch, n_streams, phase_tile));
        if (rc) return rc;
        nae_pv_segment seg{0, (long long)pl.frames, (long long)pl.frames, pv_out_len, nullptr, nullptr};
        rc = nae_launch_pv_phase(ctx, &pl, pv_src, pv_in_len, ch, n_streams, phase_tile, tile, static_cast<uint32_t*>(ctx->ws_phase), &seg);
        if (rc) return rc;
        rc = nae_launch_pv_synth(ctx, &pl, pv_src, pv_in_len, ch, n_streams, tile, phase_tile, static_cast<const uint32_t*>(ctx->ws_phase), pv_dst, &seg, fps);
        if (rc) return rc;
    }
    if (pl.rs_on && !pl.rs_first) {
        const nae_sig* rs_src = pl.pv_on ? &mid : src;
        const size_t rs_src_len = pl.pv_on ? pl.mid_len : in_len;
        rc = nae_launch_resample(ctx, &pl, rs_src, rs_src_len, ch, n_streams, ctx->d_rs_tab, dst, 0, pl.out_len);
        if (rc) return rc;
    }
    return NAE_OK;
}

extern "C" {

int nae_stretch_block_f32(nae_ctx* ctx, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                          size_t n_streams, const nae_sig* dst)
{
    return stretch_block_impl(ctx, rate, pitch, src, in_len, ch, n_streams, dst, nullptr);
}

int nae_debug_pv_tile_phase(nae_ctx* ctx, double rate, double pitch, const nae_sig* src, size_t in_len, int ch,
                            size_t n_streams, int32_t* dst_host, size_t dst_capacity, size_t* n_tiles_out,
                            size_t* tile_frames)
{
    if (!ctx || !dst_host || !n_tiles_out || !tile_frames) return NAE_ERR_INVALID;
    int rc = check_sig(ctx, src, "null source view");
    if (rc) return rc;
    nae_stretch_plan pl;
    rc = nae_stretch_plan_make(rate, pitch, in_len, &pl);
    if (rc) return nae_fail(ctx, rc, "rate/pitch outside the supported range");
    if (!pl.pv_on) return nae_fail(ctx, NAE_ERR_STATE, "phase vocoder stage is bypassed for these parameters");
    const int tile = ctx->pv_tile > 0 ? ctx->pv_tile : 64;
    const size_t n_tiles = (pl.frames + tile - 1) / tile;
    *n_tiles_out = n_tiles;
    *tile_frames = (size_t)tile;
    const size_t need = n_streams * ch * n_tiles * NAE_FFT_BINS;
    if (dst_capacity < need) return nae_fail(ctx, NAE_ERR_INVALID, "destination too small");
    const size_t ws_bytes = nae_pv_phase_workspace_bytes(pl.frames, ch, n_streams, tile);
    rc = nae_ws_reserve(ctx, &ctx->ws_phase, &ctx->ws_phase_bytes, ws_bytes);
    if (rc) return rc;
    const nae_sig* pv_src = src;
    size_t pv_in_len = in_len;
    nae_sig mid{};
    if (pl.rs_first) {
        const size_t mid_stride = (pl.mid_len + 3) & ~(size_t)3;
        rc = nae_ws_reserve(ctx, &ctx->ws_mid, &ctx->ws_mid_bytes, n_streams * ch * mid_stride * sizeof(float));
        if (rc) return rc;
        mid = nae_sig{ctx->ws_mid, (size_t)ch * mid_stride, mid_stride, 1};
        rc = nae_ensure_rs_table(ctx, pl.rate_eff);
        if (rc) return rc;
        rc = nae_launch_resample(ctx, &pl, src, in_len, ch, n_streams, ctx->d_rs_tab, &mid, 0, pl.mid_len);
        if (rc) return rc;
        pv_src = &mid;
        pv_in_len = pl.mid_len;
    }
    nae_pv_segment seg{0, (long long)pl.frames, (long long)pl.frames, 0, nullptr, nullptr};
    rc = nae_launch_pv_phase(ctx, &pl, pv_src, pv_in_len, ch, n_streams, tile, tile, static_cast<uint32_t*>(ctx->ws_phase), &seg);
    if (rc) return rc;
    std::vector<int32_t> tmp(ws_bytes / sizeof(int32_t));
    hipError_t e = hipMemcpyAsync(tmp.data(), ctx->ws_phase, ws_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return nae_check(ctx, e, "copy phase workspace");
    for (size_t rec = 0; rec < n_streams * ch * n_tiles; rec++)
        memcpy(dst_host + rec * NAE_FFT_BINS, tmp.data() + rec * kPhasePad, NAE_FFT_BINS * sizeof(int32_t));
    return NAE_OK;
}

// ------------------------------------------------------------------------------------------------ K8
size_t nae_spectrum_frames(size_t T) { return T < NAE_FFT_N ? 0 : (T - NAE_FFT_N) / NAE_HOP + 1; }

int nae_spectrum_block_f32(nae_ctx* ctx, const nae_sig* src, size_t T, int ch, size_t n_streams, float* dst,
                           size_t dst_stream_stride)
{
    if (!ctx || !dst) return NAE_ERR_INVALID;
    int rc = check_sig(ctx, src, "null source view");
    if (rc) return rc;
    if (ch < 1 || ch > 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    return nae_launch_spectrum(ctx, src, T, ch, n_streams, dst, dst_stream_stride);
}

// ------------------------------------------------------------------------------------------------ graph
// mask: 1 = mix node (+ the pitch node's transposer when that runs first), 2 = the rest of the pitch node, 4 = spectrum node
static int graph4_stages(nae_ctx* ctx, const nae_graph4* g, int mask)
{
    // node 1+2: the two inputs feed the 2-input mixer (audio-amix.cpp:86-324 with input_num = 2)
    // node 3: pitch (audio-velocity.cpp:462-477).  When its transposer runs first the mix happens inside that launch
    // (same arithmetic, same mix_out contents); otherwise the mix is launched on its own in front.
    if (!g->in_a.base || !g->in_b.base || !g->mix_out.base) return nae_fail(ctx, NAE_ERR_INVALID, "nae_graph4_run: null signal");
    const nae_mix_front front{&g->in_a, &g->in_b, g->vol_a, g->vol_b};
    int rc = NAE_OK;
    if (mask & 3) rc = stretch_block_impl(ctx, g->rate, g->pitch, &g->mix_out, g->S, 2, g->n_streams, &g->pitch_out, &front, mask & 3);
    if (rc || !(mask & 4)) return rc;
    nae_stretch_plan pl;
    rc = nae_stretch_plan_make(g->rate, g->pitch, g->S, &pl);
    if (rc) return rc;
    // node 4: spectrum
    return nae_spectrum_block_f32(ctx, &g->pitch_out, pl.out_len, 2, g->n_streams, g->spec_out, g->spec_stream_stride);
}

int nae_graph4_run(nae_ctx* ctx, const nae_graph4* g)
{
    if (!ctx || !g) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return graph4_stages(ctx, g, 7);
}

int nae_debug_graph4_stages(nae_ctx* ctx, const nae_graph4* g, int mask)
{
    if (!ctx || !g) return NAE_ERR_INVALID;
    (void)nae_use_device(ctx);
    return graph4_stages(ctx, g, mask);
}

} // extern "C"
