// kernels_nodes.hip — K1..K6: the elementwise / data-movement nodes, for gfx950.
//
// All of these are HBM-bound streaming kernels: 16 B per lane per access (dwordx4), 256-thread workgroups,
// grid capped at 256 CUs x 8 workgroups and grid-strided beyond that, scalar tail.  This file is compiled with
// -ffp-contract=off: the reference's x86-64 baseline build cannot fuse a*b+c, so neither may we
// (bit-exactness of K3/K4).
//
// Reference loops replaced (paths relative to /root/reference/src/processor):
//   K1 gain          audio-vol.cpp:75-100           K4 bimix v1   audio-bimix.cpp:310-317
//   K2 split/merge   audio-velocity.cpp:169-180,    K5 bimix v2   audio-bimix.cpp:624-627, 797-803, 833-850
//                    audio-amix.cpp:263-269         K6 to-f32     audio-velocity.cpp:150-232
//   K3 amix          audio-amix.cpp:293-307         clamp         audio-io.cpp:617-618
#include "nae_internal.h"
#include <algorithm>

namespace nae {

constexpr int kBlock = 256;
#ifndef NAE_MAX_GRID_PER_CU
#define NAE_MAX_GRID_PER_CU 128     // 819 MB gain: 8 -> 5.2 TB/s, 32 -> 5.6, 128 -> 5.9 (tools/c2.py)
#endif
constexpr unsigned kMaxGrid = 256 * NAE_MAX_GRID_PER_CU;

static inline unsigned grid_for(size_t work_items)
{
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g == 0) g = 1;
    return (unsigned)g;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// x86 cvttss2si: truncation, "integer indefinite" outside int32 (what audio-vol.cpp:98 yields for int32_t)
__device__ __forceinline__ int cvtt_x86(float f)
{
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : NAE_X86_INT_INDEFINITE;
}

struct GainF32 {
    float vol;
    __device__ __forceinline__ float operator()(float x) const { return x * vol; }
};
struct GainS16 {
    float vol;
    __device__ __forceinline__ short operator()(short x) const { return (short)(unsigned short)(unsigned)cvtt_x86((float)x * vol); }
};
struct GainS32 {
    float vol;
    __device__ __forceinline__ int operator()(int x) const { return cvtt_x86((float)x * vol); }
};
struct ClampF32 {
    __device__ __forceinline__ float operator()(float v) const { return (v < -1.0f) ? -1.0f : (1.0f < v) ? 1.0f : v; }
};

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int n = 4; };
template <> struct Vec16<int> { using type = int4; static constexpr int n = 4; };
template <> struct Vec16<short> { using type = int4; static constexpr int n = 8; };

struct PlanePtrs { const void* src[2]; void* dst[2]; };

// dst[p][i] = op(src[p][i]);  blockIdx.y = plane.  `vec` = 16-byte path is legal for this launch.
template <typename T, typename Op>
__global__ __launch_bounds__(kBlock) void map_planes_kernel(PlanePtrs pp, size_t n, Op op, bool vec)
{
    const T* __restrict__ src = static_cast<const T*>(pp.src[blockIdx.y]);
    T* __restrict__ dst = static_cast<T*>(pp.dst[blockIdx.y]);
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    constexpr int W = Vec16<T>::n;
    size_t done = 0;
    if (vec) {
        using V = typename Vec16<T>::type;
        const size_t nv = n / W;
        const V* s4 = reinterpret_cast<const V*>(src);
        V* d4 = reinterpret_cast<V*>(dst);
        for (size_t i = tid; i < nv; i += stride) {
            V v = s4[i];
            T* e = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int k = 0; k < W; k++) e[k] = op(e[k]);
            d4[i] = v;
        }
        done = nv * W;
    }
    for (size_t i = done + tid; i < n; i += stride) dst[i] = op(src[i]);
}

template <typename T, typename Op>
static int launch_map_planes(nae_ctx* ctx, const void* const* src, void* const* dst, int planes, size_t n, Op op,
                             const char* what)
{
    if (!ctx || !src || !dst) return NAE_ERR_INVALID;
    if (planes < 1 || planes > 2) return nae_fail(ctx, NAE_ERR_INVALID, "plane count must be 1 or 2");
    if (n == 0) return NAE_OK;
    PlanePtrs pp{};
    bool vec = true;
    for (int p = 0; p < planes; p++) {
        if (!src[p] || !dst[p]) return nae_fail(ctx, NAE_ERR_INVALID, "null plane pointer");
        pp.src[p] = src[p];
        pp.dst[p] = dst[p];
        vec = vec && aligned16(src[p]) && aligned16(dst[p]);
    }
    const size_t items = vec ? (n / Vec16<T>::n + Vec16<T>::n) : n;
    NAE_KLAUNCH(ctx, what, (map_planes_kernel<T, Op>), dim3(grid_for(items), planes), dim3(kBlock), 0, ctx->stream, pp, n,
                       op, vec);
    return nae_check(ctx, hipGetLastError(), what);
}

// ------------------------------------------------------------------------------------------------ K2 / sig copy
struct SigD { float* base; long long ss, cs, fs; };
enum CopyMode { kGeneric = 0, kI2P = 1, kP2I = 2, kFlat = 3 };

// stereo interleaved -> planar (optionally scaled): each thread moves 4 sample-frames (2 x float4 in)
template <bool kScale>
__global__ __launch_bounds__(kBlock) void copy_i2p_kernel(SigD src, SigD dst, long long S, float vol)
{
    const long long s = blockIdx.y;
    const float* __restrict__ in = src.base + s * src.ss;
    float* __restrict__ oL = dst.base + s * dst.ss;
    float* __restrict__ oR = oL + dst.cs;
    const long long q = S / 4;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < q; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i];
        const float4 b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        float4 l{a.x, a.z, b.x, b.z}, r{a.y, a.w, b.y, b.w};
        if (kScale) { l.x *= vol; l.y *= vol; l.z *= vol; l.w *= vol; r.x *= vol; r.y *= vol; r.z *= vol; r.w *= vol; }
        reinterpret_cast<float4*>(oL)[i] = l;
        reinterpret_cast<float4*>(oR)[i] = r;
    }
    for (long long i = q * 4 + tid; i < S; i += stride) {
        float l = in[2 * i], r = in[2 * i + 1];
        if (kScale) { l *= vol; r *= vol; }
        oL[i] = l; oR[i] = r;
    }
}

template <bool kScale>
__global__ __launch_bounds__(kBlock) void copy_p2i_kernel(SigD src, SigD dst, long long S, float vol)
{
    const long long s = blockIdx.y;
    const float* __restrict__ iL = src.base + s * src.ss;
    const float* __restrict__ iR = iL + src.cs;
    float* __restrict__ out = dst.base + s * dst.ss;
    const long long q = S / 4;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < q; i += stride) {
        float4 l = reinterpret_cast<const float4*>(iL)[i];
        float4 r = reinterpret_cast<const float4*>(iR)[i];
        if (kScale) { l.x *= vol; l.y *= vol; l.z *= vol; l.w *= vol; r.x *= vol; r.y *= vol; r.z *= vol; r.w *= vol; }
        reinterpret_cast<float4*>(out)[2 * i] = float4{l.x, r.x, l.y, r.y};
        reinterpret_cast<float4*>(out)[2 * i + 1] = float4{l.z, r.z, l.w, r.w};
    }
    for (long long i = q * 4 + tid; i < S; i += stride) {
        float l = iL[i], r = iR[i];
        if (kScale) { l *= vol; r *= vol; }
        out[2 * i] = l; out[2 * i + 1] = r;
    }
}

// same element order on both sides and contiguous per stream: flat 16-byte copy of S*ch elements per stream
template <bool kScale>
__global__ __launch_bounds__(kBlock) void copy_flat_kernel(SigD src, SigD dst, long long n, float vol)
{
    const long long s = blockIdx.y;
    const float* __restrict__ in = src.base + s * src.ss;
    float* __restrict__ out = dst.base + s * dst.ss;
    const long long q = n / 4;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < q; i += stride) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        if (kScale) { v.x *= vol; v.y *= vol; v.z *= vol; v.w *= vol; }
        reinterpret_cast<float4*>(out)[i] = v;
    }
    for (long long i = q * 4 + tid; i < n; i += stride) out[i] = kScale ? in[i] * vol : in[i];
}

template <bool kScale>
__global__ __launch_bounds__(kBlock) void copy_generic_kernel(SigD src, SigD dst, long long S, int ch, float vol)
{
    const long long s = blockIdx.y;
    const long long n = S * ch;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long e = tid; e < n; e += stride) {
        const long long i = e / ch;
        const int c = (int)(e % ch);
        const float v = src.base[s * src.ss + c * src.cs + i * src.fs];
        dst.base[s * dst.ss + c * dst.cs + i * dst.fs] = kScale ? v * vol : v;
    }
}

static inline bool is_interleaved(const nae_sig* v, int ch) { return v->chan_stride == 1 && v->frame_stride == (size_t)ch; }
static inline bool is_planar(const nae_sig* v) { return v->frame_stride == 1; }
static inline bool view_aligned(const nae_sig* v, size_t extra_stride)
{
    return aligned16(v->base) && (v->stream_stride % 4 == 0) && (extra_stride % 4 == 0);
}

} // namespace nae
using namespace nae;

int nae_launch_copy_sig(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams,
                        bool scale, float volume)
{
    if (!ctx || !src || !dst || !src->base || !dst->base) return ctx ? nae_fail(ctx, NAE_ERR_INVALID, "null signal view") : NAE_ERR_INVALID;
    if (ch < 1) return nae_fail(ctx, NAE_ERR_INVALID, "channel count < 1");
    if (S == 0 || n_streams == 0) return NAE_OK;
    SigD s{static_cast<float*>(src->base), (long long)src->stream_stride, (long long)src->chan_stride, (long long)src->frame_stride};
    SigD d{static_cast<float*>(dst->base), (long long)dst->stream_stride, (long long)dst->chan_stride, (long long)dst->frame_stride};
    int mode = kGeneric;
    if (ch == 2 && is_interleaved(src, 2) && is_planar(dst) && view_aligned(src, 0) && view_aligned(dst, dst->chan_stride)) mode = kI2P;
    else if (ch == 2 && is_planar(src) && is_interleaved(dst, 2) && view_aligned(src, src->chan_stride) && view_aligned(dst, 0)) mode = kP2I;
    else if (((is_interleaved(src, ch) && is_interleaved(dst, ch)) ||
              (is_planar(src) && is_planar(dst) && src->chan_stride == S && dst->chan_stride == S) || ch == 1) &&
             (ch > 1 || (src->frame_stride == 1 && dst->frame_stride == 1)) && view_aligned(src, 0) && view_aligned(dst, 0))
        mode = kFlat;
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        SigD ss = s, dd = d;
        ss.base += (long long)s0 * ss.ss;
        dd.base += (long long)s0 * dd.ss;
        const long long flat = (long long)S * ch;
        unsigned gx = grid_for(mode == kGeneric ? (size_t)flat : (size_t)flat / 4 + 1);
        if ((size_t)gx * ns > (size_t)kMaxGrid * 4) { gx = (unsigned)(((size_t)kMaxGrid * 4 + ns - 1) / ns); if (gx == 0) gx = 1; }
        dim3 grid(gx, ns), block(kBlock);
#define NAE_LAUNCH(K, ...)                                                                        \
        do {                                                                                      \
            if (scale) NAE_KLAUNCH(ctx, #K, (K<true>), grid, block, 0, ctx->stream, __VA_ARGS__);   \
            else NAE_KLAUNCH(ctx, #K, (K<false>), grid, block, 0, ctx->stream, __VA_ARGS__);        \
        } while (0)
        switch (mode) {
        case kI2P: NAE_LAUNCH(copy_i2p_kernel, ss, dd, (long long)S, volume); break;
        case kP2I: NAE_LAUNCH(copy_p2i_kernel, ss, dd, (long long)S, volume); break;
        case kFlat: NAE_LAUNCH(copy_flat_kernel, ss, dd, flat, volume); break;
        default: NAE_LAUNCH(copy_generic_kernel, ss, dd, (long long)S, ch, volume); break;
        }
#undef NAE_LAUNCH
        int rc = nae_check(ctx, hipGetLastError(), "copy_sig kernel");
        if (rc) return rc;
    }
    return NAE_OK;
}

namespace nae {
// ------------------------------------------------------------------------------------------------ K3 amix
struct MixPlanes { const float* inL[16]; const float* inR[16]; float vol[16]; int n; };

__global__ __launch_bounds__(kBlock) void amix_planes_kernel(MixPlanes a, float* __restrict__ outL,
                                                            float* __restrict__ outR, size_t S, bool vec)
{
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t done = 0;
    if (vec) {
        const size_t q = S / 4;
        for (size_t j = tid; j < q; j += stride) {
            float4 l{0.f, 0.f, 0.f, 0.f}, r{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < a.n; i++) {
                const float4 x = reinterpret_cast<const float4*>(a.inL[i])[j];
                const float4 y = reinterpret_cast<const float4*>(a.inR[i])[j];
                const float v = a.vol[i];
                l.x += x.x * v; l.y += x.y * v; l.z += x.z * v; l.w += x.w * v;
                r.x += y.x * v; r.y += y.y * v; r.z += y.z * v; r.w += y.w * v;
            }
            reinterpret_cast<float4*>(outL)[j] = l;
            reinterpret_cast<float4*>(outR)[j] = r;
        }
        done = q * 4;
    }
    for (size_t j = done + tid; j < S; j += stride) {
        float l = 0.0f, r = 0.0f;
        for (int i = 0; i < a.n; i++) {
            l += a.inL[i][j] * a.vol[i];
            r += a.inR[i][j] * a.vol[i];
        }
        outL[j] = l;
        outR[j] = r;
    }
}

struct MixSigs { const float* base[16]; long long ss[16], cs[16], fs[16]; float vol[16]; int n; };

// fast path: every input interleaved stereo, output planar; 4 sample-frames per thread
__global__ __launch_bounds__(kBlock) void amix_i2p_kernel(MixSigs a, SigD out, long long S)
{
    const long long s = blockIdx.y;
    float* __restrict__ oL = out.base + s * out.ss;
    float* __restrict__ oR = oL + out.cs;
    const long long q = S / 4;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long j = tid; j < q; j += stride) {
        float4 l{0.f, 0.f, 0.f, 0.f}, r{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < a.n; i++) {
            const float4* in = reinterpret_cast<const float4*>(a.base[i] + s * a.ss[i]);
            const float4 x = in[2 * j], y = in[2 * j + 1];
            const float v = a.vol[i];
            l.x += x.x * v; r.x += x.y * v; l.y += x.z * v; r.y += x.w * v;
            l.z += y.x * v; r.z += y.y * v; l.w += y.z * v; r.w += y.w * v;
        }
        reinterpret_cast<float4*>(oL)[j] = l;
        reinterpret_cast<float4*>(oR)[j] = r;
    }
    for (long long j = q * 4 + tid; j < S; j += stride) {
        float l = 0.0f, r = 0.0f;
        for (int i = 0; i < a.n; i++) {
            const float* in = a.base[i] + s * a.ss[i];
            l += in[2 * j] * a.vol[i];
            r += in[2 * j + 1] * a.vol[i];
        }
        oL[j] = l;
        oR[j] = r;
    }
}

__global__ __launch_bounds__(kBlock) void amix_generic_kernel(MixSigs a, SigD out, long long S)
{
    const long long s = blockIdx.y;
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long e = tid; e < 2 * S; e += stride) {
        const long long j = e >> 1;
        const int c = (int)(e & 1);
        float acc = 0.0f;
        for (int i = 0; i < a.n; i++) acc += a.base[i][s * a.ss[i] + c * a.cs[i] + j * a.fs[i]] * a.vol[i];
        out.base[s * out.ss + c * out.cs + j * out.fs] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ K4 bimix v1
__global__ __launch_bounds__(kBlock) void bimix_kernel(const float* __restrict__ ll, const float* __restrict__ lr,
                                                      const float* __restrict__ rl, const float* __restrict__ rr,
                                                      float bias_minus, float bias_plus, float* __restrict__ outL,
                                                      float* __restrict__ outR, size_t S, bool vec)
{
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t done = 0;
    if (vec) {
        const size_t q = S / 4;
        for (size_t i = tid; i < q; i += stride) {
            const float4 a = reinterpret_cast<const float4*>(ll)[i], b = reinterpret_cast<const float4*>(lr)[i];
            const float4 c = reinterpret_cast<const float4*>(rl)[i], d = reinterpret_cast<const float4*>(rr)[i];
            float4 l, r;
            l.x = (a.x * 0.5f + b.x * 0.5f) * bias_minus; l.y = (a.y * 0.5f + b.y * 0.5f) * bias_minus;
            l.z = (a.z * 0.5f + b.z * 0.5f) * bias_minus; l.w = (a.w * 0.5f + b.w * 0.5f) * bias_minus;
            r.x = (c.x * 0.5f + d.x * 0.5f) * bias_plus; r.y = (c.y * 0.5f + d.y * 0.5f) * bias_plus;
            r.z = (c.z * 0.5f + d.z * 0.5f) * bias_plus; r.w = (c.w * 0.5f + d.w * 0.5f) * bias_plus;
            reinterpret_cast<float4*>(outL)[i] = l;
            reinterpret_cast<float4*>(outR)[i] = r;
        }
        done = q * 4;
    }
    for (size_t i = done + tid; i < S; i += stride) {
        outL[i] = (ll[i] * 0.5f + lr[i] * 0.5f) * bias_minus; // x/2 == x*0.5f exactly (correctly rounded, same value)
        outR[i] = (rl[i] * 0.5f + rr[i] * 0.5f) * bias_plus;
    }
}

// ------------------------------------------------------------------------------------------------ K5 bimix v2
__global__ __launch_bounds__(kBlock) void downmix_kernel(const float* __restrict__ l, const float* __restrict__ r,
                                                        float* __restrict__ mono, size_t S, bool vec)
{
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t done = 0;
    if (vec) {
        const size_t q = S / 4;
        for (size_t i = tid; i < q; i += stride) {
            const float4 a = reinterpret_cast<const float4*>(l)[i], b = reinterpret_cast<const float4*>(r)[i];
            // (float)((double)(l+r) * 0.5) == (l+r)*0.5f : the halving is exact or rounds the same real value
            reinterpret_cast<float4*>(mono)[i] = float4{(a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f, (a.w + b.w) * 0.5f};
        }
        done = q * 4;
    }
    for (size_t i = done + tid; i < S; i += stride) mono[i] = (l[i] + r[i]) * 0.5f;
}

__global__ __launch_bounds__(kBlock) void bimix2_interleave_kernel(float* __restrict__ dst,
                                                                  const float* __restrict__ earlier,
                                                                  const float* __restrict__ later, size_t unaligned,
                                                                  size_t aligned, int earlier_offset)
{
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n = unaligned + aligned;
    for (size_t i = tid; i < n; i += stride) {
        const float e = earlier[i];
        const float o = (i < unaligned) ? 0.0f : later[i - unaligned];
        float2 v = earlier_offset == 0 ? float2{e, o} : float2{o, e};
        reinterpret_cast<float2*>(dst)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ K6 to f32
struct ConvPlanes { const void* p[2]; };
template <int kFmt>
__global__ __launch_bounds__(kBlock) void to_f32_kernel(ConvPlanes pl, size_t S, int ch, float* __restrict__ dst)
{
    const size_t n = S * (size_t)ch;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t e = tid; e < n; e += stride) {
        const size_t i = e / ch;
        const int c = (int)(e % ch);
        float v;
        if (kFmt == NAE_FMT_FLT) v = static_cast<const float*>(pl.p[0])[e];
        else if (kFmt == NAE_FMT_FLTP) v = static_cast<const float*>(pl.p[c])[i];
        else if (kFmt == NAE_FMT_S16) v = (float)static_cast<const short*>(pl.p[0])[e] / 32768.0f;
        else if (kFmt == NAE_FMT_S16P) v = (float)static_cast<const short*>(pl.p[c])[i] / 32767.0f; // IEEE-correct f32 divide
        else if (kFmt == NAE_FMT_S32) v = (float)static_cast<const int*>(pl.p[0])[e] / 2147483648.0f;
        else v = (float)((double)static_cast<const int*>(pl.p[c])[i] / 2147483647.0);                // f64 divide, then narrow
        dst[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------ mono -> stereo
__global__ __launch_bounds__(kBlock) void mono_to_stereo_kernel(const float* __restrict__ mono, float* __restrict__ dst,
                                                               size_t S, float gain)
{
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = tid; i < S; i += stride) {
        const float v = mono[i] * gain;
        reinterpret_cast<float2*>(dst)[i] = float2{v, v};
    }
}

// ------------------------------------------------------------------------------------------------ synthetic input
// counter-based splitmix64: element i of a stream is the (i+1)-th output of the generator seeded with `seed`
__global__ __launch_bounds__(kBlock) void fill_uniform_kernel(float* __restrict__ dst, size_t n, size_t stream_stride,
                                                             unsigned long long first_stream,
                                                             unsigned long long input_index)
{
    const size_t s = blockIdx.y;
    const unsigned long long seed = 0x9E3779B97F4A7C15ull * (1ull + first_stream + s) + input_index;
    float* __restrict__ o = dst + s * stream_stride;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = tid; i < n; i += stride) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const unsigned u = (unsigned)(z >> 32);
        o[i] = (float)(u >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
}

} // namespace nae

// ================================================================================================ C ABI
extern "C" {

int nae_gain_f32(nae_ctx* ctx, const float* const* src, float* const* dst, int planes, size_t elems, float volume)
{
    return launch_map_planes<float>(ctx, (const void* const*)src, (void* const*)dst, planes, elems, GainF32{volume}, "gain_f32");
}
int nae_gain_s16(nae_ctx* ctx, const int16_t* const* src, int16_t* const* dst, int planes, size_t elems, float volume)
{
    return launch_map_planes<short>(ctx, (const void* const*)src, (void* const*)dst, planes, elems, GainS16{volume}, "gain_s16");
}
int nae_gain_s32(nae_ctx* ctx, const int32_t* const* src, int32_t* const* dst, int planes, size_t elems, float volume)
{
    return launch_map_planes<int>(ctx, (const void* const*)src, (void* const*)dst, planes, elems, GainS32{volume}, "gain_s32");
}

int nae_gain_frame(nae_ctx* ctx, int fmt, const void* const* src, void* const* dst, size_t S, int ch, float volume)
{
    if (!ctx) return NAE_ERR_INVALID;
    if (ch != 1 && ch != 2) return nae_fail(ctx, NAE_ERR_INVALID, "Invalid channel count: only mono and stereo audio are supported"); // audio-vol.cpp:177-182
    switch (fmt) { // audio-vol.cpp:188-244
    case NAE_FMT_FLT: return nae_gain_f32(ctx, (const float* const*)src, (float* const*)dst, 1, S * ch, volume);
    case NAE_FMT_FLTP: return nae_gain_f32(ctx, (const float* const*)src, (float* const*)dst, ch, S, volume);
    case NAE_FMT_S16: return nae_gain_s16(ctx, (const int16_t* const*)src, (int16_t* const*)dst, 1, S * ch, volume);
    case NAE_FMT_S16P: return nae_gain_s16(ctx, (const int16_t* const*)src, (int16_t* const*)dst, ch, S, volume);
    case NAE_FMT_S32: return nae_gain_s32(ctx, (const int32_t* const*)src, (int32_t* const*)dst, 1, S * ch, volume);
    case NAE_FMT_S32P: return nae_gain_s32(ctx, (const int32_t* const*)src, (int32_t* const*)dst, ch, S, volume);
    default: return nae_fail(ctx, NAE_ERR_UNSUPPORTED, "Audio format is not supported (FLT, S16, S32 and planar variants only)");
    }
}

int nae_fill_uniform_f32(nae_ctx* ctx, float* dst, size_t n_per_stream, size_t stream_stride, size_t n_streams,
                         uint64_t first_stream, uint64_t input_index)
{
    if (!ctx || !dst) return NAE_ERR_INVALID;
    if (n_per_stream == 0 || n_streams == 0) return NAE_OK;
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        unsigned gx = grid_for(n_per_stream);
        if ((size_t)gx * ns > (size_t)kMaxGrid * 4) { gx = (unsigned)(((size_t)kMaxGrid * 4 + ns - 1) / ns); if (gx == 0) gx = 1; }
        NAE_KLAUNCH(ctx, "fill_uniform_kernel", fill_uniform_kernel, dim3(gx, ns), dim3(kBlock), 0, ctx->stream,
                    dst + s0 * stream_stride, n_per_stream, stream_stride, (unsigned long long)(first_stream + s0),
                    (unsigned long long)input_index);
        int rc = nae_check(ctx, hipGetLastError(), "fill_uniform_kernel");
        if (rc) return rc;
    }
    return NAE_OK;
}

// test utility: number of 32-bit words that differ between two device buffers, added to *d_count (full-size parity tests
// compare whole outputs on the device instead of copying gigabytes to the host)
__global__ __launch_bounds__(256) void diff_u32_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n,
                                                      unsigned long long* __restrict__ d_count)
{
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) local += a[i] != b[i];
    for (int off = 32; off; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(d_count, local);
}

int nae_debug_diff_u32(nae_ctx* ctx, const void* a, const void* b, size_t n_words, uint64_t* d_count)
{
    if (!ctx || !d_count || (n_words && (!a || !b))) return NAE_ERR_INVALID;
    if (n_words == 0) return NAE_OK;
    size_t blocks = (n_words + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    (void)nae_use_device(ctx);
    hipLaunchKernelGGL(diff_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(a),
                       static_cast<const uint32_t*>(b), n_words, reinterpret_cast<unsigned long long*>(d_count));
    return nae_check(ctx, hipGetLastError(), "diff_u32_kernel");
}

// shader clock the chip holds right now: every wave runs a short dependent FMA chain between two pairs of
// (s_memtime, s_memrealtime) stamps; clock = delta cycles / delta 100-MHz ticks.  Launched by bench.py directly behind
// its timed steps (the DVFS state of the load is still in force) so that cycle figures need no assumed GHz.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters)
{
    float a = 1.0f + 1e-6f * (float)threadIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
    asm volatile("" : "+v"(a));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && a != 0.0f) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
    }
}

int nae_debug_clock_ghz(nae_ctx* ctx, double* ghz)
{
    if (!ctx || !ghz) return NAE_ERR_INVALID;
    const int blocks = 1024, waves = blocks * 4;
    unsigned long long* d = nullptr;
    (void)nae_use_device(ctx);
    hipError_t e = hipMalloc((void**)&d, sizeof(unsigned long long) * 2 * waves);
    if (e != hipSuccess) return nae_check(ctx, e, "hipMalloc(clock probe)");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, 20000);
    std::vector<unsigned long long> h(2 * waves);
    e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return nae_check(ctx, e, "clock probe");
    std::vector<double> g(waves);
    for (int w = 0; w < waves; w++) g[w] = h[2 * w + 1] ? (double)h[2 * w] / (double)h[2 * w + 1] * 0.1 : 0.0;
    std::nth_element(g.begin(), g.begin() + waves / 2, g.end());
    *ghz = g[waves / 2];
    return NAE_OK;
}

int nae_mono_to_stereo_f32(nae_ctx* ctx, const float* mono, float* dst, size_t S, float gain)
{
    if (!ctx || !mono || !dst) return NAE_ERR_INVALID;
    if (S == 0) return NAE_OK;
    if (reinterpret_cast<uintptr_t>(dst) & 7) return nae_fail(ctx, NAE_ERR_INVALID, "dst must be 8-byte aligned");
    NAE_KLAUNCH(ctx, "mono_to_stereo_kernel", mono_to_stereo_kernel, dim3(grid_for(S)), dim3(kBlock), 0, ctx->stream, mono, dst, S, gain);
    return nae_check(ctx, hipGetLastError(), "mono_to_stereo_kernel");
}

int nae_clamp_f32(nae_ctx* ctx, float* data, size_t n)
{
    const void* s[1] = {data};
    void* d[1] = {data};
    return launch_map_planes<float>(ctx, s, d, 1, n, ClampF32{}, "clamp_f32");
}

int nae_copy_sig_f32(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams)
{
    return nae_launch_copy_sig(ctx, src, dst, S, ch, n_streams, false, 1.0f);
}
int nae_gain_sig_f32(nae_ctx* ctx, const nae_sig* src, const nae_sig* dst, size_t S, int ch, size_t n_streams, float volume)
{
    return nae_launch_copy_sig(ctx, src, dst, S, ch, n_streams, true, volume);
}

int nae_deinterleave_f32(nae_ctx* ctx, const float* src, float* const* dst_planes, size_t S, int ch)
{
    if (!ctx || !src || !dst_planes) return NAE_ERR_INVALID;
    if (ch != 1 && ch != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    if (ch == 2 && dst_planes[1] == dst_planes[0] + S) {
        nae_sig s{const_cast<float*>(src), 0, 1, 2}, d{dst_planes[0], 0, S, 1};
        return nae_launch_copy_sig(ctx, &s, &d, S, 2, 1, false, 1.0f);
    }
    for (int c = 0; c < ch; c++) { // independent plane buffers: one strided copy per plane
        nae_sig s{const_cast<float*>(src) + c, 0, 0, (size_t)ch}, d{dst_planes[c], 0, 0, 1};
        int rc = nae_launch_copy_sig(ctx, &s, &d, S, 1, 1, false, 1.0f);
        if (rc) return rc;
    }
    return NAE_OK;
}

int nae_interleave_f32(nae_ctx* ctx, const float* const* src_planes, float* dst, size_t S, int ch)
{
    if (!ctx || !src_planes || !dst) return NAE_ERR_INVALID;
    if (ch != 1 && ch != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    if (ch == 2 && src_planes[1] == src_planes[0] + S) {
        nae_sig s{const_cast<float*>(src_planes[0]), 0, S, 1}, d{dst, 0, 1, 2};
        return nae_launch_copy_sig(ctx, &s, &d, S, 2, 1, false, 1.0f);
    }
    for (int c = 0; c < ch; c++) {
        nae_sig s{const_cast<float*>(src_planes[c]), 0, 0, 1}, d{dst + c, 0, 0, (size_t)ch};
        int rc = nae_launch_copy_sig(ctx, &s, &d, S, 1, 1, false, 1.0f);
        if (rc) return rc;
    }
    return NAE_OK;
}

int nae_amix_f32(nae_ctx* ctx, const float* const* inL, const float* const* inR, const float* vol, int n, float* outL,
                 float* outR, size_t S)
{
    if (!ctx || !inL || !inR || !vol || !outL || !outR) return NAE_ERR_INVALID;
    if (n < 1 || n > 16) return nae_fail(ctx, NAE_ERR_INVALID, "amix input count must be 1..16"); // audio-amix.cpp:342
    if (S == 0) return NAE_OK;
    MixPlanes a{};
    a.n = n;
    bool vec = aligned16(outL) && aligned16(outR);
    for (int i = 0; i < n; i++) {
        if (!inL[i] || !inR[i]) return nae_fail(ctx, NAE_ERR_INVALID, "null amix input plane");
        a.inL[i] = inL[i]; a.inR[i] = inR[i]; a.vol[i] = vol[i];
        vec = vec && aligned16(inL[i]) && aligned16(inR[i]);
    }
    NAE_KLAUNCH(ctx, "amix_planes_kernel", amix_planes_kernel, dim3(grid_for(vec ? S / 4 + 4 : S)), dim3(kBlock), 0, ctx->stream, a, outL, outR, S, vec);
    return nae_check(ctx, hipGetLastError(), "amix_planes_kernel");
}

int nae_amix_sig_f32(nae_ctx* ctx, const nae_sig* inputs, const float* vol, int n, const nae_sig* out, size_t S,
                     size_t n_streams)
{
    if (!ctx || !inputs || !vol || !out || !out->base) return NAE_ERR_INVALID;
    if (n < 1 || n > 16) return nae_fail(ctx, NAE_ERR_INVALID, "amix input count must be 1..16");
    if (S == 0 || n_streams == 0) return NAE_OK;
    MixSigs a{};
    a.n = n;
    bool fast = is_planar(out) && view_aligned(out, out->chan_stride);
    for (int i = 0; i < n; i++) {
        if (!inputs[i].base) return nae_fail(ctx, NAE_ERR_INVALID, "null amix input");
        a.base[i] = static_cast<const float*>(inputs[i].base);
        a.ss[i] = (long long)inputs[i].stream_stride; a.cs[i] = (long long)inputs[i].chan_stride; a.fs[i] = (long long)inputs[i].frame_stride;
        a.vol[i] = vol[i];
        fast = fast && is_interleaved(&inputs[i], 2) && view_aligned(&inputs[i], 0);
    }
    SigD o{static_cast<float*>(out->base), (long long)out->stream_stride, (long long)out->chan_stride, (long long)out->frame_stride};
    for (size_t s0 = 0; s0 < n_streams; s0 += 65535) {
        const unsigned ns = (unsigned)((n_streams - s0 < 65535) ? n_streams - s0 : 65535);
        MixSigs aa = a;
        for (int i = 0; i < n; i++) aa.base[i] += (long long)s0 * aa.ss[i];
        SigD oo = o;
        oo.base += (long long)s0 * oo.ss;
        unsigned gx = grid_for(fast ? S / 4 + 1 : 2 * S);
        if ((size_t)gx * ns > (size_t)kMaxGrid * 4) { gx = (unsigned)(((size_t)kMaxGrid * 4 + ns - 1) / ns); if (gx == 0) gx = 1; }
        if (fast) NAE_KLAUNCH(ctx, "amix_i2p_kernel", amix_i2p_kernel, dim3(gx, ns), dim3(kBlock), 0, ctx->stream, aa, oo, (long long)S);
        else NAE_KLAUNCH(ctx, "amix_generic_kernel", amix_generic_kernel, dim3(gx, ns), dim3(kBlock), 0, ctx->stream, aa, oo, (long long)S);
        int rc = nae_check(ctx, hipGetLastError(), "amix_sig kernel");
        if (rc) return rc;
    }
    return NAE_OK;
}

int nae_bimix_f32(nae_ctx* ctx, const float* ll, const float* lr, const float* rl, const float* rr, float bias,
                  float* outL, float* outR, size_t S)
{
    if (!ctx || !ll || !lr || !rl || !rr || !outL || !outR) return NAE_ERR_INVALID;
    if (S == 0) return NAE_OK;
    const float bias_minus = (1 - bias), bias_plus = (1 + bias); // audio-bimix.cpp:310-311
    const bool vec = aligned16(ll) && aligned16(lr) && aligned16(rl) && aligned16(rr) && aligned16(outL) && aligned16(outR);
    NAE_KLAUNCH(ctx, "bimix_kernel", bimix_kernel, dim3(grid_for(vec ? S / 4 + 4 : S)), dim3(kBlock), 0, ctx->stream, ll, lr, rl, rr,
                       bias_minus, bias_plus, outL, outR, S, vec);
    return nae_check(ctx, hipGetLastError(), "bimix_kernel");
}

int nae_bimix2_downmix_f32(nae_ctx* ctx, const float* l, const float* r, float* mono, size_t S)
{
    if (!ctx || !l || !r || !mono) return NAE_ERR_INVALID;
    if (S == 0) return NAE_OK;
    const bool vec = aligned16(l) && aligned16(r) && aligned16(mono);
    NAE_KLAUNCH(ctx, "downmix_kernel", downmix_kernel, dim3(grid_for(vec ? S / 4 + 4 : S)), dim3(kBlock), 0, ctx->stream, l, r, mono, S, vec);
    return nae_check(ctx, hipGetLastError(), "downmix_kernel");
}

int nae_bimix2_interleave_f32(nae_ctx* ctx, float* dst, const float* earlier, const float* later, size_t unaligned,
                              size_t aligned, int earlier_offset)
{
    if (!ctx || !dst || !earlier) return NAE_ERR_INVALID;
    if (aligned > 0 && !later) return nae_fail(ctx, NAE_ERR_INVALID, "later stream missing");
    if (earlier_offset != 0 && earlier_offset != 1) return nae_fail(ctx, NAE_ERR_INVALID, "earlier_offset must be 0 or 1");
    if (unaligned + aligned == 0) return NAE_OK;
    if (reinterpret_cast<uintptr_t>(dst) & 7) return nae_fail(ctx, NAE_ERR_INVALID, "dst must be 8-byte aligned");
    NAE_KLAUNCH(ctx, "bimix2_interleave_kernel", bimix2_interleave_kernel, dim3(grid_for(unaligned + aligned)), dim3(kBlock), 0, ctx->stream, dst,
                       earlier, later ? later : earlier, unaligned, aligned, earlier_offset);
    return nae_check(ctx, hipGetLastError(), "bimix2_interleave_kernel");
}

int nae_to_f32_interleaved(nae_ctx* ctx, int fmt, const void* const* planes, size_t S, int ch, float* dst)
{
    if (!ctx || !planes || !dst) return NAE_ERR_INVALID;
    if (ch != 1 && ch != 2) return nae_fail(ctx, NAE_ERR_INVALID, "channel count must be 1 or 2");
    const bool planar = (fmt == NAE_FMT_FLTP || fmt == NAE_FMT_S16P || fmt == NAE_FMT_S32P);
    ConvPlanes pl{};
    for (int c = 0; c < (planar ? ch : 1); c++) {
        if (!planes[c]) return nae_fail(ctx, NAE_ERR_INVALID, "null plane pointer");
        pl.p[c] = planes[c];
    }
    if (S == 0) return NAE_OK;
    const dim3 grid(grid_for(S * ch)), block(kBlock);
    switch (fmt) {
    case NAE_FMT_FLT: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_FLT>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    case NAE_FMT_FLTP: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_FLTP>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    case NAE_FMT_S16: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_S16>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    case NAE_FMT_S16P: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_S16P>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    case NAE_FMT_S32: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_S32>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    case NAE_FMT_S32P: NAE_KLAUNCH(ctx, "to_f32_kernel", (to_f32_kernel<NAE_FMT_S32P>), grid, block, 0, ctx->stream, pl, S, ch, dst); break;
    default: return nae_fail(ctx, NAE_ERR_UNSUPPORTED, "Unsupported sample format"); // audio-velocity.cpp:223-228
    }
    return nae_check(ctx, hipGetLastError(), "to_f32_kernel");
}

} // extern "C"
