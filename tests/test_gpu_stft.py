"""GPU parity, K7 (tempo/pitch) and K8 (FFT spectrum), and the 4-node graph — through the C ABI.

Bars: K8 is bit-exact against the oracle (the phase path is specified to the operation) and <= 1e-4 relative RMS
against the float64 DFT golden; K7's integer synthesis phases are bit-exact and its samples are within the
1e-4 RMS tolerance BASELINE.json states for float paths."""
import os

import numpy as np
import pytest

import orc
from conftest import rel_rms

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: "<= 1e-4 RMS for float mix/resample/FFT"


def gpu_spectrum(ctx, nae, x, ch, n_streams=1, planar=False):
    T = x.size // (ch * n_streams)
    F = ctx.spectrum_frames(T)
    d_x, d_o = ctx.array(x), ctx.empty(max(1, n_streams * F * ch * 513))
    sig = nae.Sig.planar(d_x.ptr, T, ch) if planar else nae.Sig.interleaved(d_x.ptr, T, ch)
    ctx.spectrum_block(sig, T, ch, n_streams, d_o.ptr, F * ch * 513)
    out = d_o.download()[: n_streams * F * ch * 513].reshape(n_streams, F, ch, 513)
    d_x.free(); d_o.free()
    return out


@pytest.mark.parametrize("name", ["tone", "noise", "impulse"])
def test_k8_spectrum_golden_and_bit_exact(ctx, nae, golden, name):
    g = golden["spectrum"]
    x = g[f"{name}_in"]
    got = gpu_spectrum(ctx, nae, x, 1)[0, :, 0, :]
    assert rel_rms(got, g[f"{name}_mag"]) <= TOL
    assert rel_rms(got, g[f"{name}_mag"]) <= 2e-6
    ref = orc.spectrum(x, 1)[:, 0, :]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "spectrum not bit-identical to the oracle"


def test_k8_stereo_batched_layouts_and_edges(ctx, nae):
    n_streams, T = 5, 1024 + 256 * 9 + 77
    x = orc.fill_uniform(n_streams * T * 2, 31)
    got = gpu_spectrum(ctx, nae, x, 2, n_streams)
    for s in range(n_streams):
        ref = orc.spectrum(x.reshape(n_streams, -1)[s], 2)
        assert np.array_equal(got[s].view(np.uint32), ref.view(np.uint32)), s
    # planar source gives the same result
    xp = np.ascontiguousarray(x.reshape(n_streams, T, 2).transpose(0, 2, 1)).reshape(-1)
    assert np.array_equal(gpu_spectrum(ctx, nae, xp, 2, n_streams, planar=True), got)
    # too short: zero frames, nothing launched, no error
    assert gpu_spectrum(ctx, nae, orc.fill_uniform(2 * 1023, 1), 2).shape == (1, 0, 2, 513)
    assert gpu_spectrum(ctx, nae, orc.fill_uniform(1024, 1), 1).shape == (1, 1, 1, 513)


def test_k8_wide_stores_every_alignment_case(ctx, nae):
    """the stereo kernel writes a frame's 4104 bytes as 16-byte line-aligned pieces when the destination allows it: an ODD frame count
    makes the streams alternate between the two 16-byte phases (a chunk that starts on the odd phase writes its first 8 bytes alone, one
    that ends on the even phase its last 8), a destination that is only 8-byte aligned or has an odd stream stride takes the dword
    pieces; every case is bit-identical to the oracle and leaves the bytes around the result untouched"""
    n_streams, T = 7, 1024 + 256 * 12 + 3                      # 13 frames per stream
    F = ctx.spectrum_frames(T)
    assert F == 13
    x = orc.fill_uniform(n_streams * T * 2, 32)
    ref = np.stack([orc.spectrum(x.reshape(n_streams, -1)[s], 2) for s in range(n_streams)])
    d_x = ctx.array(x)
    sig = nae.Sig.interleaved(d_x.ptr, T, 2)
    n = F * 2 * 513
    for lead, stride in ((0, n), (2, n), (0, n + 1), (4, n + 2), (1, n)):     # floats in front of the result / stream stride
        guard = np.float32(-7.0)
        total = lead + n_streams * stride + 8
        d_o = ctx.array(np.full(total, guard, np.float32))
        ctx.spectrum_block(sig, T, 2, n_streams, d_o.at(lead), stride)
        out = d_o.download()
        d_o.free()
        for st in range(n_streams):
            got = out[lead + st * stride: lead + st * stride + n].reshape(F, 2, 513)
            assert np.array_equal(got.view(np.uint32), ref[st].view(np.uint32)), (lead, stride, st)
            if stride > n:
                assert (out[lead + st * stride + n: lead + (st + 1) * stride] == guard).all(), "the gap between two streams is untouched"
        assert (out[:lead] == guard).all() and (out[lead + n_streams * stride:] == guard).all(), "nothing written outside the result"
    d_x.free()


def test_k8_guided_chunk_list_equals_the_generic_kernel(ctx, nae):
    """a large batch runs the persistent launch: 16-frame chunks, 8-frame chunks for the last streams, every chunk but a wave's first
    drawn from a device counter that a hipMemsetAsync on the stream zeroes in front of every drawing launch (launches with no more items than waves never
    touch it).  8200 streams x 77 frames (odd: both store phases, a 13-frame tail chunk)
    against the one-wave-per-frame generic kernel on a second context, compared on the device; twice in a row (the counter must be back at
    zero), and three streams against the oracle"""
    import os
    n_streams, F = 8200, 77
    T = 1024 + 256 * (F - 1)
    n = F * 2 * 513
    d_x, d_a, d_b = ctx.empty(n_streams * T * 2), ctx.empty(n_streams * n), ctx.empty(n_streams * n)
    ctx.fill_uniform(d_x.ptr, T * 2, T * 2, n_streams, 0, 0)
    sig = nae.Sig.interleaved(d_x.ptr, T, 2)
    with nae.Context(0) as generic:
        generic.debug_set("spec_generic", 1)
        generic.spectrum_block(sig, T, 2, n_streams, d_b.ptr, n)
        generic.sync()
    d_cnt = ctx.array(np.zeros(2, np.uint64))
    for rep in range(2):
        ctx._ck(ctx.lib.nae_memset(ctx.h, d_a.ptr, 0xFF, n_streams * n * 4))
        ctx.spectrum_block(sig, T, 2, n_streams, d_a.ptr, n)
        ctx.diff_words(d_a.ptr, d_b.ptr, n_streams * n, d_cnt.at(rep))
    ctx.sync()
    assert d_cnt.download().tolist() == [0, 0], "words differing between the persistent stereo kernel and the generic one (first, second launch)"
    for st in (0, 4100, 8199):
        xs = np.empty(T * 2, np.float32)
        ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, xs.ctypes.data, d_x.at(st * T * 2), xs.nbytes))
        got = np.empty(n, np.float32)
        ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, got.ctypes.data, d_a.at(st * n), got.nbytes))
        ctx.sync()
        assert np.array_equal(got.view(np.uint32), orc.spectrum(xs, 2).reshape(-1).view(np.uint32)), st
    for d in (d_x, d_a, d_b, d_cnt):
        d.free()


@pytest.mark.parametrize("planar", [False, True])
def test_k8_non_finite_samples_stay_in_their_frames(ctx, nae, planar):
    """one NaN and one Inf sample in the left channel: exactly the frames whose window covers them are non-finite (as in the
    oracle), the right channel and every other frame stay bit-exact — the stereo kernel keeps 6 of a frame's 8 input rows in
    registers for the next frame and walks 32 frames per wave, so a slip there would spread them"""
    T = 1024 + 256 * 70 + 5
    x = orc.fill_uniform(2 * T, 91).reshape(T, 2).copy()
    bad = {3000: np.nan, 11111: np.inf}
    for pos, v in bad.items():
        x[pos, 0] = v
    flat = np.ascontiguousarray(x.T).reshape(-1) if planar else x.reshape(-1)
    got = gpu_spectrum(ctx, nae, flat, 2, planar=planar)[0]
    ref = orc.spectrum(x.reshape(-1), 2)
    F = got.shape[0]
    touched = np.zeros(F, bool)
    for pos in bad:
        for f in range(F):
            touched[f] |= 256 * f <= pos < 256 * f + 1024
    assert touched.sum() == 8
    assert np.array_equal(got[:, 1].view(np.uint32), ref[:, 1].view(np.uint32)), "right channel"
    assert np.array_equal(got[~touched, 0].view(np.uint32), ref[~touched, 0].view(np.uint32)), "clean frames of the left channel"
    assert not np.isfinite(got[touched, 0]).any() and not np.isfinite(ref[touched, 0]).any()
    assert np.isfinite(got[~touched]).all()


def test_k8_linearity_at_full_size(ctx, nae):
    """size-independent property at the C5 per-stream size (10 s): spectrum(2x) == 2*spectrum(x), bit-exact"""
    T = 480000
    x = orc.fill_uniform(2 * T, 77)
    a = gpu_spectrum(ctx, nae, x, 2)
    b = gpu_spectrum(ctx, nae, (2 * x).astype(np.float32), 2)
    assert a.shape == (1, 1872, 2, 513)
    assert np.array_equal((2 * a).astype(np.float32), b)
    # and a strided sample of frames against the oracle
    for f in (0, 1, 935, 1871):
        ref = orc.spectrum(x[2 * 256 * f: 2 * (256 * f + 1024)], 2)[0]
        assert np.array_equal(a[0, f].view(np.uint32), ref.view(np.uint32))


def gpu_stretch(ctx, nae, x, ch, rate, pitch, n_streams=1, planar_in=False, planar_out=False):
    L = x.size // (ch * n_streams)
    pl = ctx.stretch_plan(rate, pitch, L)
    d_x, d_o = ctx.array(x), ctx.empty(max(1, n_streams * pl.out_len * ch))
    src = nae.Sig.planar(d_x.ptr, L, ch) if planar_in else nae.Sig.interleaved(d_x.ptr, L, ch)
    dst = nae.Sig.planar(d_o.ptr, pl.out_len, ch) if planar_out else nae.Sig.interleaved(d_o.ptr, pl.out_len, ch)
    ctx.stretch_block(rate, pitch, src, L, ch, n_streams, dst)
    out = d_o.download()[: n_streams * pl.out_len * ch]
    d_x.free(); d_o.free()
    return out, pl


def tone(L, amp=(0.5, 0.25), f=(1000.0, 3300.0)):
    n = np.arange(L)
    return sum(a * np.sin(2 * np.pi * fr * n / 48000) for a, fr in zip(amp, f)).astype(np.float32)


@pytest.mark.parametrize("rate,pitch", [(1.0, 2 ** (3 / 12)), (1.0, 2 ** (-5 / 12)), (1.5, 1 / 1.5), (0.5, 2.0)])
def test_k7_integer_phases_bit_exact(ctx, nae, rate, pitch):
    """the synthesis phase in front of every tile equals the oracle's Q0.32 phase, on NOISE (the chaotic case)"""
    L, ch = 40000, 2
    x = orc.fill_uniform(L * ch, 41)
    d_x = ctx.array(x)
    got, tile = ctx.debug_pv_tile_phase(rate, pitch, nae.Sig.interleaved(d_x.ptr, L, ch), L, ch, 1)
    d_x.free()
    qs = orc.pv_synth_phase(x, ch, rate, pitch)          # [frames, ch, 513]
    n_tiles = got.shape[2]
    assert n_tiles >= 2
    for j in range(n_tiles):
        for c in range(ch):
            ref = qs[j * tile - 1, c] if j > 0 else np.zeros(513, np.int32)
            assert np.array_equal(got[0, c, j], ref), (j, c, int(np.count_nonzero(got[0, c, j] != ref)))


@pytest.mark.parametrize("rate,pitch,kind", [(1.0, 2 ** (3 / 12), "tone"), (1.0, 2 ** (3 / 12), "noise"),
                                             (1.0, 2 ** (-7 / 12), "noise"), (1.5, 1 / 1.5, "tone"),
                                             (0.6, 1 / 0.6, "noise"), (1.5, 1.0, "noise"), (0.5, 1.0, "tone"),
                                             (3.0, 1 / 3.0, "noise")])
def test_k7_stretch_vs_oracle(ctx, nae, rate, pitch, kind):
    L, ch = 30000, 2
    if kind == "tone":
        m = tone(L)
        x = np.stack([m, 0.5 * m], 1).reshape(-1).astype(np.float32)
    else:
        x = orc.fill_uniform(L * ch, 43)
    got, pl = gpu_stretch(ctx, nae, x, ch, rate, pitch)
    ref = orc.stretch(x, ch, rate, pitch)
    assert got.size == ref.size == pl.out_len * ch
    assert pl.out_len == int(np.floor(L / rate + 0.5))
    assert np.isfinite(got).all()
    assert rel_rms(got, ref) <= TOL, rel_rms(got, ref)
    # worst sample error relative to the signal's RMS stays small too (no isolated garbage)
    assert np.abs(got - ref).max() <= 1e-3 * np.sqrt(np.mean(ref.astype(np.float64) ** 2))


def test_k7_golden_regression_and_mono(ctx, nae, golden):
    g = golden["k7_regression"]
    for name in ("pitch_up3", "tempo_1p5", "rate_0p8"):
        rate, pitch = g[name + "_params"]
        got, _ = gpu_stretch(ctx, nae, g["in"], 1, float(rate), float(pitch))
        assert rel_rms(got, g[name]) <= TOL, name


def test_k7_identity_and_layouts(ctx, nae):
    L, ch = 5000, 2
    x = orc.fill_uniform(L * ch, 45)
    got, _ = gpu_stretch(ctx, nae, x, ch, 1.0, 1.0)
    assert np.array_equal(got, x)                                      # both stages bypassed: bit copy
    # planar in / planar out equals interleaved in / interleaved out
    p = 2 ** (3 / 12)
    a, pl = gpu_stretch(ctx, nae, x, ch, 1.0, p)
    xp = np.ascontiguousarray(x.reshape(L, ch).T).reshape(-1)
    b, _ = gpu_stretch(ctx, nae, xp, ch, 1.0, p, planar_in=True, planar_out=True)
    assert np.array_equal(b.reshape(ch, pl.out_len).T.reshape(-1), a)
    # stretch only (no transposer) writes straight to the destination in both layouts
    a, pl = gpu_stretch(ctx, nae, x, ch, 1.25, 0.8)
    b, _ = gpu_stretch(ctx, nae, xp, ch, 1.25, 0.8, planar_in=True, planar_out=True)
    assert np.array_equal(b.reshape(ch, pl.out_len).T.reshape(-1), a)


@pytest.mark.parametrize("name", ["up3", "down4", "tempo_1p5", "tempo_0p7", "rate2_up5", "rate_0p8"])
def test_k7_against_the_specification_golden(ctx, nae, golden, name):
    """GPU against tests/golden/k7_golden.npz (float64 numpy restatement of the K7 specification), no oracle in the loop"""
    g = golden["k7_golden"]
    ch, rate, pitch = g[name + "_params"]
    got, _ = gpu_stretch(ctx, nae, g[str(g[name + "_src"])], int(ch), float(rate), float(pitch))
    assert got.size == g[name].size
    assert rel_rms(got, g[name]) <= TOL


def test_k7_non_finite_sample_is_confined(ctx, nae):
    """a NaN sample in one channel: the output is non-finite only where the oracle's is (the frames whose window covers it,
    after the transposer), the other channel and the rest of the stream stay within tolerance — phases of the poisoned frames are
    0 on both sides (atan2 of a non-positive maximum), so the recurrence carries on identically"""
    L, ch, pitch = 60000, 2, 2 ** (3 / 12)
    x = (0.5 * orc.fill_uniform(L * ch, 43)).reshape(L, ch).copy()
    x[30001, 0] = np.nan
    got = gpu_stretch(ctx, nae, x.reshape(-1), ch, 1.0, pitch)[0].reshape(-1, ch)
    ref = orc.stretch(x.reshape(-1), ch, 1.0, pitch).reshape(-1, ch)
    assert got.shape == ref.shape
    bad_ref, bad_got = ~np.isfinite(ref), ~np.isfinite(got)
    assert not bad_ref[:, 1].any() and not bad_got[:, 1].any(), "the clean channel stays finite"
    assert 1000 < bad_ref[:, 0].sum() < 4000
    # same poisoned span up to the transposer's 16-tap reach at its edges
    lo, hi = np.flatnonzero(bad_ref[:, 0])[[0, -1]]
    glo, ghi = np.flatnonzero(bad_got[:, 0])[[0, -1]]
    assert abs(int(lo) - int(glo)) <= 16 and abs(int(hi) - int(ghi)) <= 16, (lo, hi, glo, ghi)
    ok = np.ones(ref.shape[0], bool)
    ok[min(lo, glo) - 16: max(hi, ghi) + 17] = False
    assert rel_rms(got[ok], ref[ok]) <= TOL, rel_rms(got[ok], ref[ok])


def test_k7_batched_streams_are_independent(ctx, nae):
    n_streams, L, ch = 9, 12000, 2
    x = orc.fill_uniform(n_streams * L * ch, 47)
    p = 2 ** (3 / 12)
    got, pl = gpu_stretch(ctx, nae, x, ch, 1.0, p, n_streams)
    got = got.reshape(n_streams, -1)
    for s in range(n_streams):      # the transposer takes 4 streams per workgroup: every slot of a group, and a partial group
        one, _ = gpu_stretch(ctx, nae, x.reshape(n_streams, -1)[s].copy(), ch, 1.0, p)
        assert np.array_equal(one, got[s]), s
        if s in (0, 4, 8):
            assert rel_rms(got[s], orc.stretch(x.reshape(n_streams, -1)[s], ch, 1.0, p)) <= TOL


def test_k7_edge_lengths(ctx, nae):
    p = 2 ** (3 / 12)
    for L in (0, 1, 255, 256, 1023, 1025):
        x = orc.fill_uniform(max(L, 1) * 2, 49)[: L * 2]
        if L == 0:
            pl = ctx.stretch_plan(1.0, p, 0)
            assert pl.out_len == 0
            continue
        got, pl = gpu_stretch(ctx, nae, x, 2, 1.0, p)
        ref = orc.stretch(x, 2, 1.0, p)
        assert got.size == ref.size
        assert rel_rms(got, ref) <= TOL or np.sqrt(np.mean(ref.astype(np.float64) ** 2)) < 1e-6


def test_k7_full_size_properties(ctx, nae):
    """C3-shaped size-independent checks on a long stream (60 s stereo): length, finiteness, tone lands on pitch,
    and time-shift invariance of the tiling (prefix of the long run == the run on the prefix, away from the end)."""
    L, ch = 60 * 48000, 2
    m = tone(L)
    x = np.stack([m, m], 1).reshape(-1).astype(np.float32)
    p = 2 ** (3 / 12)
    got, pl = gpu_stretch(ctx, nae, x, ch, 1.0, p)
    assert pl.out_len == L and np.isfinite(got).all()
    seg = got.reshape(L, ch)[48000 * 20: 48000 * 20 + 65536, 0].astype(np.float64)
    sp = np.abs(np.fft.rfft(seg * np.hanning(seg.size)))
    assert abs(np.argmax(sp) * 48000 / seg.size - 1000 * p) < 2.0
    short, _ = gpu_stretch(ctx, nae, x[: 2 * 200000].copy(), ch, 1.0, p)
    assert rel_rms(got[: 2 * 150000], short[: 2 * 150000]) <= 1e-6
    # against the oracle on the first 2 s
    ref = orc.stretch(x[: 2 * 200000], ch, 1.0, p)
    assert rel_rms(got[: 2 * 96000], ref[: 2 * 96000]) <= TOL


@pytest.mark.parametrize("n_streams,S,semis,rate,shared_b,planar_mix,vols", [
    (3, 20000, 3, 1.0, True, True, (0.5, 0.5)),        # transposer first: the mix is fused into its staging
    (6, 8193, 5, 1.0, False, True, (0.3, 0.9)),        # one full group of 4 streams + a partial one, odd length
    (5, 3000, 3, 1.0, True, False, (1.0, 0.25)),       # interleaved mix output; shorter than 3 transposer tiles
    (2, 20000, -4, 1.0, True, True, (0.5, 0.5)),       # vocoder first: the mix stays a launch of its own
    (2, 12000, 0, 1.5, True, True, (0.5, 0.5)),        # plain rate change (transposer only)
    (1, 6000, 0, 1.0, True, True, (0.5, 0.5)),         # identity pitch node
])
def test_graph4_matches_node_by_node_oracle(ctx, nae, n_streams, S, semis, rate, shared_b, planar_mix, vols):
    """input -> mix(2) -> pitch -> spectrum: every node's output against the oracle run node by node"""
    a = orc.fill_uniform(n_streams * S * 2, 51)
    b = orc.fill_uniform((1 if shared_b else n_streams) * S * 2, 52)
    p = 2 ** (semis / 12)
    pl = ctx.stretch_plan(rate, p, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_a, d_b = ctx.array(a), ctx.array(b)
    d_mix, d_pitch, d_spec = ctx.empty(n_streams * S * 2), ctx.empty(max(1, n_streams * pl.out_len * 2)), ctx.empty(max(1, n_streams * F * 2 * 513))
    g = nae.Graph4()
    g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
    g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=shared_b)
    g.vol_a, g.vol_b = vols
    g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2) if planar_mix else nae.Sig.interleaved(d_mix.ptr, S, 2)
    g.rate, g.pitch = rate, p
    g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
    g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
    g.S, g.n_streams = S, n_streams
    ctx.graph4(g)
    mix, pitch, spec = d_mix.download(), d_pitch.download(), d_spec.download()
    for s in range(n_streams):
        xs = a.reshape(n_streams, S, 2)[s]
        bs = b.reshape(-1, S, 2)[0 if shared_b else s]
        L, R = orc.amix([xs[:, 0], bs[:, 0]], [xs[:, 1], bs[:, 1]], list(vols))
        m = mix.reshape(n_streams, 2, S)[s] if planar_mix else mix.reshape(n_streams, S, 2)[s].T
        assert np.array_equal(m[0], L) and np.array_equal(m[1], R), s                      # mix: bit-exact
        ref_p = orc.stretch(np.stack([L, R], 1).reshape(-1), 2, rate, p)
        got_p = pitch[: n_streams * pl.out_len * 2].reshape(n_streams, -1)[s]
        assert rel_rms(got_p, ref_p) <= TOL                                                # pitch: tolerance
        if F:
            got_s = spec[: n_streams * F * 2 * 513].reshape(n_streams, F, 2, 513)[s]
            ref_s = orc.spectrum(got_p, 2)                                                 # spectrum of what the GPU fed it
            assert np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32))
            assert rel_rms(got_s, orc.spectrum(ref_p, 2)) <= TOL                           # end to end
    for d in (d_a, d_b, d_mix, d_pitch, d_spec):
        d.free()


def stream_stretch(ctx, x, ch, rate, pitch, put_sizes, recv_chunk=3456, device_put=False):
    """drive the SoundTouch-shaped handle the way audio-velocity.cpp:344-440 does: put a frame, drain what is ready"""
    import ctypes as C
    lib = ctx.lib
    L = x.size // ch
    h = C.c_void_p()
    assert lib.nae_stretch_create(ctx.h, 48000, ch, rate, pitch, C.byref(h)) == 0
    outs, avail_before_flush, pos, i = [], 0, 0, 0
    d_x = ctx.array(x) if device_put else None

    def drain(limit):
        nonlocal outs
        while lib.nae_stretch_available(h) > limit:
            buf = np.empty(recv_chunk * ch, np.float32)
            got = C.c_size_t()
            assert lib.nae_stretch_receive_host(h, buf.ctypes.data, recv_chunk, C.byref(got)) == 0
            outs.append(buf[: got.value * ch])

    while pos < L:
        n = min(put_sizes[i % len(put_sizes)], L - pos)
        i += 1
        if device_put:
            assert lib.nae_stretch_put(h, d_x.at(pos * ch), n) == 0
        else:
            chunk = np.ascontiguousarray(x[pos * ch:(pos + n) * ch])
            assert lib.nae_stretch_put_host(h, chunk.ctypes.data, n) == 0
        pos += n
        avail_before_flush = max(avail_before_flush, lib.nae_stretch_available(h))
        drain(int(1152 / rate))
    assert lib.nae_stretch_flush(h) == 0
    drain(0)
    assert lib.nae_stretch_destroy(h) == 0
    if d_x is not None:
        d_x.free()
    return np.concatenate(outs) if outs else np.zeros(0, np.float32), avail_before_flush


@pytest.mark.parametrize("rate,pitch", [(1.0, 2 ** (3 / 12)), (1.0, 2 ** (-4 / 12)), (1.5, 1 / 1.5), (0.7, 1 / 0.7), (1.5, 1.0),
                                        (0.8, 1.0), (1.0, 1.0)])
def test_stretch_streaming_equals_block(ctx, nae, rate, pitch):
    """incremental put/receive yields exactly the block result, and output becomes available BEFORE flush"""
    L, ch = 60000, 2
    x = orc.fill_uniform(L * ch, 61)
    rate, pitch = float(np.float32(rate)), float(np.float32(pitch))   # the handle takes floats (setRate/setPitch)
    blk, pl = gpu_stretch(ctx, nae, x, ch, rate, pitch)
    for sizes, dev in (([1152], False), ([1152, 4096, 37, 9000, 1, 20000], True), ([L], False)):
        y, early = stream_stretch(ctx, x, ch, rate, pitch, sizes, device_put=dev)
        assert y.size == blk.size, (sizes, y.size, blk.size)
        assert np.array_equal(y, blk), (sizes, int(np.count_nonzero(y != blk)))
        if len(sizes) == 1 and sizes[0] == 1152:
            assert early > 0      # numSamples() grows while the stream is still being fed


def test_streaming_handles(ctx, nae):
    import ctypes as C
    lib = ctx.lib
    L, ch = 20000, 2
    x = orc.fill_uniform(L * ch, 61)
    p = float(np.float32(2 ** (3 / 12)))   # the SoundTouch-shaped API takes float parameters (setPitch(float))
    h = C.c_void_p()
    assert lib.nae_stretch_create(ctx.h, 96000, ch, 1.0, p, C.byref(h)) == -2     # audio-velocity.cpp:371-379
    assert lib.nae_stretch_create(ctx.h, 48000, 3, 1.0, p, C.byref(h)) == -1
    # mono stream
    xm = np.ascontiguousarray(x[0::2])
    ym, _ = stream_stretch(ctx, xm, 1, 1.0, p, [1000])
    bm, _ = gpu_stretch(ctx, nae, xm, 1, 1.0, p)
    assert np.array_equal(ym, bm)
    # --- spectrum: chunked puts give the same frames as the block call
    hs = C.c_void_p()
    assert lib.nae_spectrum_create(ctx.h, 1024, 256, ch, C.byref(hs)) == 0
    d_x = ctx.array(x)
    frames = []
    pos = 0
    for n in (500, 1000, 3000, 1, 255, L):
        n = min(n, L - pos)
        assert lib.nae_spectrum_put(hs, d_x.at(pos * ch), n) == 0
        pos += n
        k = lib.nae_spectrum_available(hs)
        if k:
            d_o = ctx.empty(k * ch * 513)
            got = C.c_size_t()
            assert lib.nae_spectrum_receive(hs, d_o.ptr, k, C.byref(got)) == 0 and got.value == k
            ctx.sync()
            frames.append(d_o.download().reshape(k, ch, 513))
            d_o.free()
    assert lib.nae_spectrum_destroy(hs) == 0
    allf = np.concatenate(frames)
    block = gpu_spectrum(ctx, nae, x, ch)[0]
    assert np.array_equal(allf, block)
    # --- partial receives: take FEWER frames than are ready between puts (the handle compacts what is left), then the rest
    assert lib.nae_spectrum_create(ctx.h, 1024, 256, ch, C.byref(hs)) == 0
    frames, pos, got = [], 0, C.c_size_t()
    for n in (5000, 3000, 7000, L):
        n = min(n, L - pos)
        assert lib.nae_spectrum_put(hs, d_x.at(pos * ch), n) == 0
        pos += n
        k = lib.nae_spectrum_available(hs)
        take = max(1, k // 3) if k else 0
        if take:
            d_o = ctx.empty(take * ch * 513)
            assert lib.nae_spectrum_receive(hs, d_o.ptr, take, C.byref(got)) == 0 and got.value == take
            ctx.sync()
            frames.append(d_o.download().reshape(take, ch, 513))
            d_o.free()
            assert lib.nae_spectrum_available(hs) == k - take
    while lib.nae_spectrum_available(hs):
        k = min(7, lib.nae_spectrum_available(hs))
        d_o = ctx.empty(k * ch * 513)
        assert lib.nae_spectrum_receive(hs, d_o.ptr, k, C.byref(got)) == 0 and got.value == k
        ctx.sync()
        frames.append(d_o.download().reshape(k, ch, 513))
        d_o.free()
    assert lib.nae_spectrum_destroy(hs) == 0
    d_x.free()
    assert np.array_equal(np.concatenate(frames), block)


def test_k7_k8_odd_layouts_equal_clean_layouts(ctx, nae):
    """misaligned bases and padded strides select the non-vectorised load / store paths of the STFT kernels and of the
    transposer; the arithmetic is the same, so the results must equal the clean-layout results bit for bit"""
    rng = np.random.default_rng(7)
    n_streams, L, ch = 3, 9000, 2
    x = orc.fill_uniform(n_streams * L * ch, 91)
    for rate, pitch in ((1.0, 2 ** (3 / 12)), (1.0, 2 ** (-4 / 12)), (1.3, 1.0)):
        clean, pl = gpu_stretch(ctx, nae, x, ch, rate, pitch, n_streams)
        clean = clean.reshape(n_streams, pl.out_len, ch)
        for planar_in, planar_out in ((False, False), (True, False), (False, True)):
            pe_s, pe_d = int(rng.integers(1, 4)), int(rng.integers(1, 4))
            if planar_in:
                cs_s, fs_s = L + 3, 1
                ss_s = ch * cs_s + 5
            else:
                cs_s, fs_s = 1, ch + 1
                ss_s = L * fs_s + 5
            if planar_out:
                cs_d, fs_d = pl.out_len + 1, 1
                ss_d = ch * cs_d + 3
            else:
                cs_d, fs_d = 1, ch
                ss_d = pl.out_len * ch + 1
            src = np.zeros(pe_s + n_streams * ss_s + 8, np.float32)
            xs = x.reshape(n_streams, L, ch)
            idx_s = np.arange(n_streams)[:, None, None] * ss_s + np.arange(L)[None, :, None] * fs_s + np.arange(ch)[None, None, :] * cs_s + pe_s
            src[idx_s] = xs
            d_src, d_dst = ctx.array(src), ctx.array(np.full(pe_d + n_streams * ss_d + 8, 5.0, np.float32))
            ctx.stretch_block(rate, pitch, nae.Sig(d_src.at(pe_s), ss_s, cs_s, fs_s), L, ch, n_streams, nae.Sig(d_dst.at(pe_d), ss_d, cs_d, fs_d))
            got = d_dst.download()
            idx_d = np.arange(n_streams)[:, None, None] * ss_d + np.arange(pl.out_len)[None, :, None] * fs_d + np.arange(ch)[None, None, :] * cs_d + pe_d
            assert np.array_equal(got[idx_d].view(np.uint32), clean.view(np.uint32)), (rate, pitch, planar_in, planar_out)
            mask = np.ones(got.size, bool)
            mask[idx_d] = False
            assert np.all(got[mask] == 5.0)                          # nothing outside the view is written
            # spectrum of the odd-layout source equals the spectrum of the clean one
            F = ctx.spectrum_frames(L)
            d_o = ctx.empty(n_streams * F * ch * 513)
            ctx.spectrum_block(nae.Sig(d_src.at(pe_s), ss_s, cs_s, fs_s), L, ch, n_streams, d_o.ptr, F * ch * 513)
            ref = gpu_spectrum(ctx, nae, x, ch, n_streams)
            assert np.array_equal(d_o.download().reshape(ref.shape).view(np.uint32), ref.view(np.uint32))
            d_src.free(); d_dst.free(); d_o.free()


def test_k7_many_tiles_equal_one_tile_bit_for_bit(nae):
    """A long lone stream is cut into hundreds of time tiles: pass 1 sums each tile's phase increments (one wave per tile), pass 2 scans them — with
    256 tiles or more per stream-channel by 16 threads per bin (pv_scan_chunked_kernel) — and pass 3 synthesises every tile from its carried phase.
    92 s of stereo in 269 tiles of 64 frames, and in the library's own choice, equal the same stream run as ONE tile (no pass 1, no scan) bit for bit."""
    ch, L, rate, pitch = 2, 4_400_000, 1.0, 2 ** (3 / 12)
    x = (0.5 * orc.fill_uniform(L * ch, 99)).astype(np.float32)
    outs = {}
    for key, knobs in (("one tile", {"pv_fps": 1, "pv_tile": 1000000}), ("269 tiles", {"pv_tile": 64}), ("library", {}), ("pass-1 tiles >= 64", {"pv_min_ptile": 64}),
                       ("frame-interleaved + tiles", {"pv_fps": 4}), ("16-frame tiles", {"pv_tile": 16})):
        with nae.Context(0) as c:
            for k, v in knobs.items():
                c.debug_set(k, v)
            c.prof_reset(); c.prof_enable(True)
            outs[key] = gpu_stretch(c, nae, x, ch, rate, pitch)[0]
            c.prof_enable(False)
            launched = set(c.prof_report())
        assert ("pv_scan_kernel" in launched) == (key != "one tile"), (key, launched)
    for key in outs:
        assert np.array_equal(outs[key].view(np.uint32), outs["one tile"].view(np.uint32)), key


def test_k7_continued_segments_of_256_tiles_carry_the_phase_through_the_chunked_scan(nae):
    """A stream handle fed in pieces of >= 256 pass-1 tiles (`debug_set("pv_tile", 64)`: 64-frame tiles, 4 M sample-frames per put = 290 tiles) runs pass 2 as
    pv_scan_chunked_kernel WITH a phase carried in and out of every segment (the block-mode test above has neither).  The concatenated output equals
    the block call of a context with the library's own shape, bit for bit."""
    ch, L, rate, pitch = 2, 9_000_000, 1.0, float(np.float32(2 ** (3 / 12)))
    x = (0.5 * orc.fill_uniform(L * ch, 123)).astype(np.float32)
    with nae.Context(0) as c:
        blk, _ = gpu_stretch(c, nae, x, ch, rate, pitch)
    with nae.Context(0) as c:
        c.debug_set("pv_tile", 64)
        c.prof_reset(); c.prof_enable(True)
        y, _ = stream_stretch(c, x, ch, rate, pitch, [4_000_000, 4_000_000, 1_000_000], recv_chunk=1 << 18, device_put=True)
        c.prof_enable(False)
        assert "pv_scan_kernel" in c.prof_report()
    assert y.size == blk.size
    assert np.array_equal(y.view(np.uint32), blk.view(np.uint32)), int(np.count_nonzero(y != blk))


@pytest.mark.parametrize("ch,n_streams,L,rate,pitch", [(2, 3, 30000, 1.0, 2 ** (3 / 12)), (1, 5, 21001, 1.0, 2 ** (-4 / 12)), (2, 2, 9000, 1.5, 1 / 1.5)])
def test_k7_pipeline_modes_agree_bit_for_bit(nae, ch, n_streams, L, rate, pitch):
    """The vocoder pipeline runs a stream-channel through its four roles one frame per step (large batches) or 2 / 4 consecutive
    frames per step (frame-interleaved: small batches), with or without time tiles (pass 1 + scan).  All shapes deliver the same
    samples bit for bit — same integer phases, same overlap-add order — and match the oracle within the tolerance.  The shapes are
    forced through nae_debug_set (pv_flow / pv_fps / pv_tile)."""
    x = (0.5 * orc.fill_uniform(n_streams * L * ch, 77)).astype(np.float32)
    outs = {}
    for flow in (0, 2):                  # two barriers per step, one buffer per hand-off / one barrier, doubled hand-off buffers
        for fps in (1, 2, 4):
            for tile in (0, 64):
                with nae.Context(0) as c:
                    c.debug_set("pv_flow", flow).debug_set("pv_fps", fps).debug_set("pv_tile", tile)
                    outs[(flow, fps, tile)] = gpu_stretch(c, nae, x, ch, rate, pitch, n_streams=n_streams)[0]
    first = outs[(0, 1, 0)]
    for key, o in outs.items():
        assert np.array_equal(o.view(np.uint32), first.view(np.uint32)), key
    per = first.reshape(n_streams, -1)
    for s in (0, n_streams - 1):
        ref = orc.stretch(x.reshape(n_streams, -1)[s], ch, rate, pitch)
        assert rel_rms(per[s], ref) <= TOL
