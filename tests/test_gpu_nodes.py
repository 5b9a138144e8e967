"""GPU parity, K1..K6: HIP kernels (through the C ABI) vs the CPU oracle and the golden vectors — bit-exact."""
import numpy as np
import pytest

import orc
from conftest import rel_rms

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 2: np.uint16}[a.dtype.itemsize])


def assert_bits(a, b, what=""):
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if not np.array_equal(bits(a), bits(b)):
        bad = np.flatnonzero(bits(a) != bits(b))
        raise AssertionError(f"{what}: {bad.size} of {a.size} elements differ, first at {bad[:5]}: {a.flat[bad[0]]!r} vs {b.flat[bad[0]]!r}")


def test_native_library_is_loaded(ctx, nae):
    """the round-end driver records which .so the test process loaded: make sure it is ours, on a gfx950"""
    maps = open("/proc/self/maps").read()
    assert "libnae_gpu.so" in maps
    assert "gfx950" in ctx.name(), ctx.name()


# ---------------------------------------------------------------- K1
@pytest.mark.parametrize("vol", [0.0, 0.70710678, 1.0, 10.0])
def test_k1_gain_f32_golden(ctx, golden, vol):
    g = golden["nodes"]
    x = g["k1_f32_in"]
    d_in, d_out = ctx.array(x), ctx.empty(x.shape, np.float32)
    ctx.gain(np.float32, [d_in.ptr], [d_out.ptr], x.size, vol)          # FLT packed: one plane
    assert_bits(d_out.download(), g[f"k1_f32_v{vol}"], "packed")
    # FLTP: two planes, deliberately mis-aligned second plane (odd length 4099 -> +4099*4 bytes)
    planes = np.concatenate([x[0::2], x[1::2]])
    S = x.size // 2
    d_p, d_q = ctx.array(planes), ctx.empty(planes.shape, np.float32)
    ctx.gain(np.float32, [d_p.at(0), d_p.at(S)], [d_q.at(0), d_q.at(S)], S, vol)
    out = d_q.download()
    assert_bits(out[:S], g[f"k1_f32_v{vol}"][0::2].copy(), "plane0")
    assert_bits(out[S:], g[f"k1_f32_v{vol}"][1::2].copy(), "plane1")
    for a in (d_in, d_out, d_p, d_q):
        a.free()


@pytest.mark.parametrize("dtype,key,vols", [(np.int16, "s16", [0.5, 0.70710678, 1.0, 3.0, 10.0]),
                                            (np.int32, "s32", [0.5, 0.70710678, 1.0, 3.0])])
def test_k1_gain_int_golden(ctx, golden, dtype, key, vols):
    g = golden["nodes"]
    x = g[f"k1_{key}_in"]
    d_in, d_out = ctx.array(x), ctx.empty(x.shape, dtype)
    for vol in vols:
        ctx.gain(dtype, [d_in.ptr], [d_out.ptr], x.size, vol)
        got = d_out.download()
        assert_bits(got, g[f"k1_{key}_v{vol}"], f"{key} vol {vol}")
        assert_bits(got, orc.change_volume([x], vol)[0], "oracle")
    # unaligned start (offset by one element) takes the scalar path
    ctx.gain(dtype, [d_in.at(1)], [d_out.at(1)], x.size - 1, 0.5)
    assert_bits(d_out.download()[1:], orc.change_volume([x[1:].copy()], 0.5)[0], "unaligned")
    d_in.free(); d_out.free()


def test_k1_gain_frame_dispatch_and_errors(ctx, nae):
    x = orc.fill_uniform(2 * 1152, 5)
    d_in, d_out = ctx.array(x), ctx.empty(x.shape, np.float32)
    assert ctx.gain_frame(nae.FMT_FLT, [d_in.ptr], [d_out.ptr], 1152, 2, 0.25) == 0
    assert_bits(d_out.download(), orc.change_volume([x], 0.25)[0])
    assert ctx.gain_frame(nae.FMT_FLTP, [d_in.at(0), d_in.at(1152)], [d_out.at(0), d_out.at(1152)], 1152, 2, 2.0) == 0
    assert_bits(d_out.download(), orc.change_volume([x], 2.0)[0])
    assert ctx.gain_frame(nae.FMT_FLT, [d_in.ptr], [d_out.ptr], 1152, 3, 1.0) == -1    # audio-vol.cpp:177-182
    assert ctx.gain_frame(4, [d_in.ptr], [d_out.ptr], 1152, 2, 1.0) == -2               # DBL: audio-vol.cpp:238-243
    assert ctx.gain_frame(nae.FMT_FLT, [d_in.ptr], [d_out.ptr], 0, 2, 1.0) == 0        # empty frame
    d_in.free(); d_out.free()


def test_k1_in_place_and_large(ctx):
    n = (1 << 22) + 3
    x = orc.fill_uniform(n, 9)
    d = ctx.array(x)
    ctx.gain(np.float32, [d.ptr], [d.ptr], n, 0.70710678)
    assert_bits(d.download(), orc.change_volume([x], 0.70710678)[0])
    d.free()


# ---------------------------------------------------------------- K2
@pytest.mark.parametrize("S", [1, 3, 4, 1152, 4096, 4099])
def test_k2_split_merge(ctx, S):
    x = orc.fill_uniform(2 * S, 11 + S)
    d_x, d_pl, d_y = ctx.array(x), ctx.empty(2 * S, np.float32), ctx.empty(2 * S, np.float32)
    ctx.deinterleave(d_x.ptr, [d_pl.at(0), d_pl.at(S)], S, 2)
    pl = d_pl.download()
    L, R = orc.deinterleave(x, 2)
    assert_bits(pl[:S], L); assert_bits(pl[S:], R)
    ctx.interleave([d_pl.at(0), d_pl.at(S)], d_y.ptr, S, 2)
    assert_bits(d_y.download(), x)
    for a in (d_x, d_pl, d_y):
        a.free()


def test_k2_separate_plane_buffers_and_mono(ctx):
    S = 1000
    x = orc.fill_uniform(2 * S, 3)
    d_x, d_l, d_r, d_y = ctx.array(x), ctx.empty(S), ctx.empty(S + 5), ctx.empty(2 * S)
    ctx.deinterleave(d_x.ptr, [d_l.ptr, d_r.at(1)], S, 2)          # non-adjacent, mis-aligned planes
    assert_bits(d_l.download(), x[0::2].copy()); assert_bits(d_r.download()[1:S + 1], x[1::2].copy())
    ctx.interleave([d_l.ptr, d_r.at(1)], d_y.ptr, S, 2)
    assert_bits(d_y.download(), x)
    ctx.interleave([d_x.ptr], d_y.ptr, 2 * S, 1)                    # mono: plain copy
    assert_bits(d_y.download(), x)
    for a in (d_x, d_l, d_r, d_y):
        a.free()


def test_k2_config1_split_gain_merge_chain(ctx, nae):
    """BASELINE.json configs[1]: split -> gain -> merge over 1000 buffers of 4096 stereo sample-frames"""
    n, S, vol = 1000, 4096, 0.7071
    x = orc.fill_uniform(n * S * 2, orc.stream_seed(0))
    d_x, d_pl, d_g, d_y = ctx.array(x), ctx.empty(x.size), ctx.empty(x.size), ctx.empty(x.size)
    inter = lambda a: nae.Sig.interleaved(a.ptr, S, 2)
    planar = lambda a: nae.Sig.planar(a.ptr, S, 2)
    ctx.copy_sig(inter(d_x), planar(d_pl), S, 2, n)                               # split
    ctx.gain(np.float32, [d_pl.ptr], [d_g.ptr], x.size, vol)                       # gain (FLTP planes back to back)
    ctx.copy_sig(planar(d_g), inter(d_y), S, 2, n)                                # merge
    ref = orc.change_volume([x], vol)[0]
    assert_bits(d_y.download(), ref, "three-node chain")
    pl = d_pl.download().reshape(n, 2, S)
    xs = x.reshape(n, S, 2)
    assert np.array_equal(pl[:, 0, :], xs[:, :, 0]) and np.array_equal(pl[:, 1, :], xs[:, :, 1])
    ctx.gain_sig(inter(d_x), inter(d_y), S, 2, n, vol)                            # fused form
    assert_bits(d_y.download(), ref, "fused")
    for a in (d_x, d_pl, d_g, d_y):
        a.free()


# ---------------------------------------------------------------- K3
@pytest.mark.parametrize("n", [1, 2, 16])
def test_k3_amix_golden(ctx, golden, n):
    g = golden["nodes"]
    ins = g["k3_in"]
    S = ins.shape[1] // 2
    planes = np.concatenate([np.concatenate([ins[i][0::2], ins[i][1::2]]) for i in range(n)])
    d_in, d_o = ctx.array(planes), ctx.empty(2 * S)
    inL = [d_in.at(2 * S * i) for i in range(n)]
    inR = [d_in.at(2 * S * i + S) for i in range(n)]
    ctx.amix(inL, inR, g[f"k3_vol_n{n}"], d_o.at(0), d_o.at(S), S)
    out = d_o.download()
    assert_bits(out[:S], g[f"k3_L_n{n}"], "L"); assert_bits(out[S:], g[f"k3_R_n{n}"], "R")
    d_in.free(); d_o.free()


def test_k3_amix_sig_config0(ctx, nae):
    """BASELINE.json configs[0]: 2-input mix, 10 s of 48 kHz stereo f32 (interleaved in, planar out)"""
    S = 480000
    a = orc.fill_uniform(2 * S, orc.stream_seed(0, 0))
    b = orc.fill_uniform(2 * S, orc.stream_seed(0, 1))
    d_a, d_b, d_o = ctx.array(a), ctx.array(b), ctx.empty(2 * S)
    ctx.amix_sig([nae.Sig.interleaved(d_a.ptr, S, 2), nae.Sig.interleaved(d_b.ptr, S, 2)], [0.5, 0.5],
                 nae.Sig.planar(d_o.ptr, S, 2), S, 1)
    out = d_o.download()
    L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
    assert_bits(out[:S], L, "L"); assert_bits(out[S:], R, "R")
    for x in (d_a, d_b, d_o):
        x.free()


def test_k3_amix_sig_batched_shared_input_and_generic(ctx, nae):
    n_streams, S = 7, 1153
    a = orc.fill_uniform(n_streams * 2 * S, 21)
    b = orc.fill_uniform(2 * S, 22)                      # shared second input: stream_stride 0
    d_a, d_b, d_o = ctx.array(a), ctx.array(b), ctx.empty(n_streams * 2 * S)
    ctx.amix_sig([nae.Sig.interleaved(d_a.ptr, S, 2), nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)], [0.3, 0.7],
                 nae.Sig.planar(d_o.ptr, S, 2), S, n_streams)
    out = d_o.download().reshape(n_streams, 2, S)
    for s in range(n_streams):
        xs = a.reshape(n_streams, S, 2)[s]
        L, R = orc.amix([xs[:, 0], b[0::2]], [xs[:, 1], b[1::2]], [0.3, 0.7])
        assert_bits(out[s, 0], L, f"stream {s} L"); assert_bits(out[s, 1], R, f"stream {s} R")
    # generic path: planar inputs, interleaved output
    pa = np.ascontiguousarray(a.reshape(n_streams, S, 2).transpose(0, 2, 1)).reshape(-1)
    d_pa = ctx.array(pa)
    ctx.amix_sig([nae.Sig.planar(d_pa.ptr, S, 2), nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)], [0.3, 0.7],
                 nae.Sig.interleaved(d_o.ptr, S, 2), S, n_streams)
    out2 = d_o.download().reshape(n_streams, S, 2)
    assert np.array_equal(out2.transpose(0, 2, 1), out)
    for x in (d_a, d_b, d_o, d_pa):
        x.free()


def test_k3_rejects_bad_input_count(ctx, nae):
    d = ctx.empty(16)
    with pytest.raises(nae.NaeError):
        ctx.amix([d.ptr] * 17, [d.ptr] * 17, [1.0] * 17, d.ptr, d.ptr, 4)
    d.free()


# ---------------------------------------------------------------- K4 / K5 / K6 / clamp
@pytest.mark.parametrize("bias", [-1.0, -0.3, 0.0, 0.25, 1.0])
def test_k4_bimix_golden(ctx, golden, bias):
    g = golden["nodes"]
    q = g["k4_in"]
    S = q.shape[1]
    d_q, d_o = ctx.array(q.reshape(-1)), ctx.empty(2 * S)
    ctx.bimix(d_q.at(0), d_q.at(S), d_q.at(2 * S), d_q.at(3 * S), bias, d_o.at(0), d_o.at(S), S)
    out = d_o.download()
    assert_bits(out[:S], g[f"k4_L_b{bias}"]); assert_bits(out[S:], g[f"k4_R_b{bias}"])
    d_q.free(); d_o.free()


def test_k5_bimix2_golden(ctx, golden):
    g = golden["nodes"]
    q = g["k4_in"]
    S = q.shape[1]
    d_q, d_m, d_i = ctx.array(q.reshape(-1)), ctx.empty(S), ctx.empty(2 * 537)
    ctx.bimix2_downmix(d_q.at(0), d_q.at(S), d_m.ptr, S)
    assert_bits(d_m.download(), g["k5_mono"])
    ctx.bimix2_interleave(d_i.ptr, d_q.at(2 * S), d_q.at(3 * S), 37, 500, 0)
    assert_bits(d_i.download(), g["k5_inter_e0"])
    ctx.bimix2_interleave(d_i.ptr, d_q.at(2 * S), d_q.at(3 * S), 37, 500, 1)
    assert_bits(d_i.download(), g["k5_inter_e1"])
    ctx.bimix2_interleave(d_i.ptr, d_q.at(2 * S), None, 100, 0, 0)      # single-sided tail
    t = d_i.download()[:200]
    assert np.array_equal(t[0::2], q[2][:100]) and not t[1::2].any()
    # subnormal halving rounds like the reference's double multiply
    tiny = np.array([1e-45, 3e-45, 1.2e-38, -1e-45] * 4, np.float32)
    d_t, d_z, d_r = ctx.array(tiny), ctx.array(np.zeros_like(tiny)), ctx.empty(tiny.size)
    ctx.bimix2_downmix(d_t.ptr, d_z.ptr, d_r.ptr, tiny.size)
    assert_bits(d_r.download(), orc.bimix2_downmix(tiny, np.zeros_like(tiny)))
    for a in (d_q, d_m, d_i, d_t, d_z, d_r):
        a.free()


def test_k6_to_f32_golden(ctx, golden, nae):
    g = golden["nodes"]
    a16, a32, af = g["k6_s16_planes"], g["k6_s32_planes"], g["k6_f32_planes"]
    S = a16.shape[1]
    d_o = ctx.empty(2 * S)
    d16p, d16 = ctx.array(a16.reshape(-1)), ctx.array(a16.T.reshape(-1).copy())
    d32p, d32 = ctx.array(a32.reshape(-1)), ctx.array(a32.T.reshape(-1).copy())
    dfp = ctx.array(af.reshape(-1))
    cases = [(nae.FMT_S16, [d16.ptr], "k6_s16"), (nae.FMT_S16P, [d16p.at(0), d16p.at(S)], "k6_s16p"),
             (nae.FMT_S32, [d32.ptr], "k6_s32"), (nae.FMT_S32P, [d32p.at(0), d32p.at(S)], "k6_s32p"),
             (nae.FMT_FLTP, [dfp.at(0), dfp.at(S)], "k6_fltp")]
    for fmt, planes, key in cases:
        assert ctx.to_f32_interleaved(fmt, planes, S, 2, d_o.ptr) == 0
        assert_bits(d_o.download(), g[key], key)
    assert ctx.to_f32_interleaved(4, [dfp.ptr], S, 2, d_o.ptr) == -2     # unsupported: audio-velocity.cpp:223-228
    for a in (d_o, d16p, d16, d32p, d32, dfp):
        a.free()


def test_clamp(ctx):
    x = np.concatenate([orc.fill_uniform(1001, 1) * 3, np.array([np.inf, -np.inf, np.nan, -0.0], np.float32)])
    d = ctx.array(x)
    ctx.clamp(d.ptr, x.size)
    assert_bits(d.download(), orc.clamp(x))
    d.free()


def test_poll_is_nonblocking(ctx):
    d = ctx.array(orc.fill_uniform(1 << 20, 2))
    ctx.gain(np.float32, [d.ptr], [d.ptr], d.size, 0.5)
    assert ctx.poll() in (0, 1)
    ctx.sync()
    assert ctx.poll() == 1
    d.free()


def test_fill_uniform_matches_oracle_generator(ctx):
    n, ns = 10007, 3
    d = ctx.empty(ns * n)
    ctx.fill_uniform(d.ptr, n, n, ns, 5, 1)
    got = d.download().reshape(ns, n)
    for s in range(ns):
        assert_bits(got[s], orc.fill_uniform(n, orc.stream_seed(5 + s, 1)), f"stream {s}")
    d.free()


def test_n2_input_conversion(ctx, nae):
    """nae_swr: identity input is a bit copy; mono s16 at 44.1 kHz -> 48 kHz stereo equals the oracle composition
    K6 -> m/sqrt(2) -> libswresample-default polyphase resampler (restated: unpinned versus FFmpeg itself)"""
    import ctypes as C
    lib = ctx.lib
    # --- identity: 48 kHz stereo FLT in uneven chunks, swr_convert semantics (max_out smaller than what is buffered)
    S = 5000
    x = orc.fill_uniform(2 * S, 71)
    h = C.c_void_p()
    assert lib.nae_swr_create(ctx.h, nae.FMT_FLT, 48000, 2, 48000, C.byref(h)) == 0
    outL, outR, got = [], [], C.c_size_t()
    pos = 0
    for n in (1152, 1152, 2000, 696):
        chunk = np.ascontiguousarray(x[2 * pos: 2 * (pos + n)])
        planes = (C.c_void_p * 1)(chunk.ctypes.data)
        L, R = np.zeros(1000, np.float32), np.zeros(1000, np.float32)
        assert lib.nae_swr_convert_host(h, planes, n, L.ctypes.data, R.ctypes.data, 1000, C.byref(got)) == 0
        outL.append(L[: got.value].copy()); outR.append(R[: got.value].copy())
        pos += n
    while True:
        L, R = np.zeros(1000, np.float32), np.zeros(1000, np.float32)
        assert lib.nae_swr_convert_host(h, None, 0, L.ctypes.data, R.ctypes.data, 1000, C.byref(got)) == 0
        if got.value == 0:
            break
        outL.append(L[: got.value].copy()); outR.append(R[: got.value].copy())
    assert lib.nae_swr_destroy(h) == 0
    assert_bits(np.concatenate(outL), x[0::2].copy(), "identity L")
    assert_bits(np.concatenate(outR), x[1::2].copy(), "identity R")
    # --- mono s16 @ 44.1 kHz, one put + drain: K6 scaling, m / sqrt(2), then the swr-default polyphase resampler
    n = 22050
    m = (orc.fill_uniform(n, 72) * 30000).astype(np.int16)
    assert lib.nae_swr_create(ctx.h, nae.FMT_S16, 44100, 1, 48000, C.byref(h)) == 0
    planes = (C.c_void_p * 1)(m.ctypes.data)
    cap = 30000
    L, R = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    assert lib.nae_swr_convert_host(h, planes, n, L.ctypes.data, R.ctypes.data, cap, C.byref(got)) == 0
    k1 = got.value
    assert 0 < k1 < 24000                                # the last half filter length waits for the drain
    assert lib.nae_swr_convert_host(h, None, 0, L[k1:].ctypes.data, R[k1:].ctypes.data, cap - k1, C.byref(got)) == 0
    total = k1 + got.value
    assert lib.nae_swr_destroy(h) == 0
    rc, f = orc.to_f32_interleaved(orc.FMT_S16, [m], n, 1)
    st = (f * np.float32(0.70710678118654752440)).astype(np.float32)
    ref = orc.swr_resample(st, 44100, 48000)
    assert total == ref.size == 24000
    assert_bits(L[:total].copy(), ref, "resampled L")                    # bit-exact vs the oracle (oracle/orc_swr.c)
    assert_bits(R[:total].copy(), ref, "resampled R")
    assert lib.nae_swr_create(ctx.h, 4, 48000, 2, 48000, C.byref(h)) == -2     # AV_SAMPLE_FMT_DBL
    assert lib.nae_swr_create(ctx.h, nae.FMT_FLT, 48000, 6, 48000, C.byref(h)) == -1


def swr_drive(ctx, nae, fmt, in_rate, ch, planes_of, n_total, chunks, max_out=4096, queued=False):
    """swr_convert-style driving of nae_swr: put the chunks (receiving at most max_out frames per call), then drain.
    queued: through nae_swr_convert — device planes, nothing waited for until every call has been queued"""
    import ctypes as C
    lib = ctx.lib
    h, got = C.c_void_p(), C.c_size_t()
    assert lib.nae_swr_create(ctx.h, fmt, in_rate, ch, 48000, C.byref(h)) == 0
    outL, outR, pos, i = [], [], 0, 0
    keep, dev = [], []          # host planes stay alive until the stream has been waited for

    def call(planes, n):
        if queued:
            d = ctx.empty(2 * max_out, np.float32)
            assert lib.nae_swr_convert(h, planes, n, d.ptr, d.ptr + 4 * max_out, max_out, C.byref(got)) == 0
            dev.append((d, got.value))
        else:
            L, R = np.zeros(max_out, np.float32), np.zeros(max_out, np.float32)
            assert lib.nae_swr_convert_host(h, planes, n, L.ctypes.data, R.ctypes.data, max_out, C.byref(got)) == 0
            outL.append(L[: got.value].copy()); outR.append(R[: got.value].copy())
        return got.value

    while pos < n_total:
        n = min(chunks[i % len(chunks)], n_total - pos)
        arrs = planes_of(pos, n)
        keep.append(arrs)
        call((C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs]), n)
        pos += n
        i += 1
    while call(None, 0):
        pass
    if queued:
        ctx.sync()
        for d, n in dev:
            both = d.download()
            outL.append(both[:n].copy()); outR.append(both[max_out: max_out + n].copy())
    assert lib.nae_swr_destroy(h) == 0
    return np.concatenate(outL), np.concatenate(outR)


@pytest.mark.parametrize("fmt_name,in_rate,ch", [("FLT", 44100, 2), ("S16P", 22050, 1), ("FLTP", 48000, 2), ("S32", 48000, 1)])
def test_n2_queued_convert_equals_host_convert(ctx, nae, fmt_name, in_rate, ch):
    """nae_swr_convert (device planes, no wait between calls: what the mixer nodes queue per waiting frame) delivers the
    frames of nae_swr_convert_host bit for bit, call by call"""
    rng = np.random.default_rng(77)
    n = 9000
    fmt = getattr(nae, "FMT_" + fmt_name)
    planar = fmt_name.endswith("P")
    if fmt_name.startswith("FLT"):
        x = rng.uniform(-1, 1, (n, ch)).astype(np.float32)
    elif fmt_name.startswith("S16"):
        x = rng.integers(-32768, 32767, (n, ch)).astype(np.int16)
    else:
        x = rng.integers(-2**31, 2**31 - 1, (n, ch)).astype(np.int32)

    def planes_of(p, k):
        if planar:
            return [np.ascontiguousarray(x[p: p + k, c]) for c in range(ch)]
        return [np.ascontiguousarray(x[p: p + k].reshape(-1))]

    chunks = [1152, 1152, 300, 2049]
    a = swr_drive(ctx, nae, fmt, in_rate, ch, planes_of, n, chunks, max_out=1152)
    b = swr_drive(ctx, nae, fmt, in_rate, ch, planes_of, n, chunks, max_out=1152, queued=True)
    assert a[0].size == b[0].size and a[0].size > 0
    assert_bits(b[0], a[0], f"{fmt_name} L")
    assert_bits(b[1], a[1], f"{fmt_name} R")


@pytest.mark.parametrize("in_rate", [44100, 22050, 96000, 8000, 88200])
def test_n2_resampler_golden_and_chunking(ctx, nae, golden, in_rate):
    """the swr-default resampler on the golden signals: bit-exact vs the oracle, <= 1e-4 relative RMS vs the float64
    golden (tests/golden/swr_numpy.py), and independent of how the input is cut into swr_convert calls"""
    g = golden["swr_golden"]
    for name in ("tones", "noise", "impulse"):
        x, ref64 = g[name + "_in"], g[f"{name}_{in_rate}_48000"]
        st = np.stack([x, -x], 1).reshape(-1).astype(np.float32)            # stereo FLT: R = -L
        for chunks in ([6000], [1152, 1152, 577, 3001], [64, 1, 700]):
            L, R = swr_drive(ctx, nae, nae.FMT_FLT, in_rate, 2, lambda p, n: [np.ascontiguousarray(st[2 * p: 2 * (p + n)])], x.size, chunks,
                             max_out=1500)
            ref = orc.swr_resample(x, in_rate, 48000)
            assert_bits(L, ref, f"{name} {in_rate} L chunks {chunks[:2]}")
            assert_bits(R, orc.swr_resample(-x, in_rate, 48000), f"{name} {in_rate} R")
            assert L.size == ref64.size
        assert rel_rms(L, ref64) <= 1e-4


def test_many_streams_and_empty_calls(ctx, nae):
    """more streams than blockIdx.y can index in one launch (65535), and zero-sized calls"""
    n, S = 70001, 8
    x = orc.fill_uniform(n * S * 2, 81)
    d_x, d_pl, d_y = ctx.array(x), ctx.empty(x.size), ctx.empty(x.size)
    ctx.copy_sig(nae.Sig.interleaved(d_x.ptr, S, 2), nae.Sig.planar(d_pl.ptr, S, 2), S, 2, n)
    pl = d_pl.download().reshape(n, 2, S)
    assert np.array_equal(pl[:, 0, :], x.reshape(n, S, 2)[:, :, 0]) and np.array_equal(pl[:, 1, :], x.reshape(n, S, 2)[:, :, 1])
    ctx.amix_sig([nae.Sig.interleaved(d_x.ptr, S, 2), nae.Sig.interleaved(d_x.ptr, S, 2)], [0.25, 0.5], nae.Sig.planar(d_y.ptr, S, 2), S, n)
    y = d_y.download().reshape(n, 2, S)
    ref = ((np.float32(0) + x * np.float32(0.25)).astype(np.float32) + (x * np.float32(0.5)).astype(np.float32)).astype(np.float32)
    assert np.array_equal(y[:, 0, :], ref.reshape(n, S, 2)[:, :, 0]) and np.array_equal(y[:, 1, :], ref.reshape(n, S, 2)[:, :, 1])
    ctx.fill_uniform(d_y.ptr, 4, 4, n, 3, 2)
    f = d_y.download()[: 4 * n].reshape(n, 4)
    for s in (0, 65534, 65535, 65536, 70000):
        assert_bits(f[s], orc.fill_uniform(4, orc.stream_seed(3 + s, 2)), f"stream {s}")
    # zero-sized work is a no-op, not an error
    ctx.copy_sig(nae.Sig.interleaved(d_x.ptr, 0, 2), nae.Sig.planar(d_pl.ptr, 0, 2), 0, 2, 5)
    ctx.amix_sig([nae.Sig.interleaved(d_x.ptr, S, 2)], [1.0], nae.Sig.planar(d_y.ptr, S, 2), S, 0)
    ctx.gain(np.float32, [d_x.ptr], [d_y.ptr], 0, 0.5)
    ctx.spectrum_block(nae.Sig.interleaved(d_x.ptr, 100, 2), 100, 2, 3, d_y.ptr, 0)
    ctx.stretch_block(1.0, 2 ** (3 / 12), nae.Sig.interleaved(d_x.ptr, 0, 2), 0, 2, 4, nae.Sig.interleaved(d_y.ptr, 0, 2))
    ctx.sync()
    for a in (d_x, d_pl, d_y):
        a.free()


def test_non_finite_input_does_not_fault(ctx, nae):
    """NaN / Inf samples flow through every node without a GPU fault (values are outside the bit-exact guarantee)"""
    S = 6000
    x = orc.fill_uniform(2 * S, 83)
    x[100] = np.nan; x[2001] = np.inf; x[4002] = -np.inf
    d_x, d_y = ctx.array(x), ctx.empty(4 * S)
    p = 2 ** (3 / 12)
    pl = ctx.stretch_plan(1.0, p, S)
    ctx.stretch_block(1.0, p, nae.Sig.interleaved(d_x.ptr, S, 2), S, 2, 1, nae.Sig.interleaved(d_y.ptr, pl.out_len, 2))
    ctx.spectrum_block(nae.Sig.interleaved(d_x.ptr, S, 2), S, 2, 1, d_y.ptr, 0)
    ctx.sync()
    assert ctx.poll() == 1
    d_x.free(); d_y.free()


def _place(rng, n_streams, S, ch, layout):
    """a random nae_sig layout: returns (host buffer size in floats, Sig builder, index array [s][c][i] -> element)"""
    pad_e = int(rng.integers(0, 4))                       # misalign the base by 0..3 floats
    if layout == "interleaved":
        fs = ch + int(rng.integers(0, 3))                 # frames may be padded
        cs = 1
        span = S * fs
    elif layout == "planar":
        fs = 1
        cs = S + int(rng.integers(0, 5))
        span = ch * cs
    else:                                                 # "strided": arbitrary non-overlapping strides
        fs = int(rng.integers(1, 4)) * ch + int(rng.integers(0, 2))
        cs = 1 if fs >= ch else S * fs
        span = S * fs + ch
    ss = span + int(rng.integers(0, 7))
    idx = (np.arange(n_streams)[:, None, None] * ss + np.arange(ch)[None, :, None] * cs + np.arange(S)[None, None, :] * fs) + pad_e
    return pad_e + n_streams * ss + 8, (pad_e, ss, cs, fs), idx


def test_random_signal_layouts(ctx, nae):
    """K2/K1/K3 through every kernel path the views can select (16-byte fast paths, interleaved<->planar, flat, generic
    strided), with misaligned bases, padded strides and odd lengths; expected values by plain numpy indexing"""
    rng = np.random.default_rng(20260104)
    for case in range(36):
        n_streams = int(rng.integers(1, 6))
        S = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 1000, 4099]))
        ch = int(rng.integers(1, 3))
        lay_src, lay_dst = rng.choice(["interleaved", "planar", "strided"], 2)
        aligned = case % 3 == 0                           # every third case: clean layouts, so the fast paths run too
        if aligned:
            lay_src, lay_dst = rng.choice(["interleaved", "planar"], 2)
        n_src, (pe_s, ss_s, cs_s, fs_s), idx_s = _place(rng, n_streams, S, ch, lay_src)
        n_dst, (pe_d, ss_d, cs_d, fs_d), idx_d = _place(rng, n_streams, S, ch, lay_dst)
        if aligned:
            S = (S + 3) & ~3
            n_src, (pe_s, ss_s, cs_s, fs_s), idx_s = (n_streams * S * ch, (0, S * ch, 1 if lay_src == "interleaved" else S, ch if lay_src == "interleaved" else 1), None)
            n_dst, (pe_d, ss_d, cs_d, fs_d), idx_d = (n_streams * S * ch, (0, S * ch, 1 if lay_dst == "interleaved" else S, ch if lay_dst == "interleaved" else 1), None)
            mk = lambda ss, cs, fs: np.arange(n_streams)[:, None, None] * ss + np.arange(ch)[None, :, None] * cs + np.arange(S)[None, None, :] * fs
            idx_s, idx_d = mk(ss_s, cs_s, fs_s), mk(ss_d, cs_d, fs_d)
        src = rng.uniform(-1, 1, n_src).astype(np.float32)
        d_src, d_dst = ctx.array(src), ctx.array(np.full(n_dst, 7.0, np.float32))
        s_sig = nae.Sig(d_src.at(pe_s), ss_s, cs_s, fs_s)
        d_sig = nae.Sig(d_dst.at(pe_d), ss_d, cs_d, fs_d)
        vol = float(np.float32(rng.uniform(0, 2)))
        # gain (copy is gain with volume 1: checked on a few cases)
        if case % 4 == 0:
            ctx.copy_sig(s_sig, d_sig, S, ch, n_streams)
            want_vals = src[idx_s]
        else:
            ctx.gain_sig(s_sig, d_sig, S, ch, n_streams, vol)
            want_vals = (src[idx_s] * np.float32(vol)).astype(np.float32)
        got = d_dst.download()
        want = np.full(n_dst, 7.0, np.float32)
        want[idx_d] = want_vals
        assert np.array_equal(got, want), (case, lay_src, lay_dst, n_streams, S, ch)          # and nothing outside the view is touched
        # 2-input mix of the same source with itself shifted by one stream's worth of data (stereo only)
        if ch == 2:
            src2 = rng.uniform(-1, 1, n_src).astype(np.float32)
            d_src2 = ctx.array(src2)
            s2_sig = nae.Sig(d_src2.at(pe_s), ss_s, cs_s, fs_s)
            d_dst.upload(np.full(n_dst, 7.0, np.float32))
            v = [float(np.float32(rng.uniform(0, 1))), float(np.float32(rng.uniform(0, 1)))]
            ctx.amix_sig([s_sig, s2_sig], v, d_sig, S, n_streams)
            acc = (np.float32(0.0) + src[idx_s] * np.float32(v[0])).astype(np.float32)
            acc = (acc + (src2[idx_s] * np.float32(v[1])).astype(np.float32)).astype(np.float32)
            want = np.full(n_dst, 7.0, np.float32)
            want[idx_d] = acc
            assert np.array_equal(d_dst.download(), want), ("amix", case, lay_src, lay_dst, n_streams, S)
            d_src2.free()
        d_src.free(); d_dst.free()
