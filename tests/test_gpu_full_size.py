"""GPU, BASELINE.json's configs at their full sizes on one GPU.

C5 (configs[4]): 1024 streams x 10 s through input -> mix(2) -> pitch -> spectrum, and the 128 / 512 streams one rank of the
8-GPU / 2-GPU job owns.  The oracle cannot run 1024 streams in seconds, so the whole batch is pinned through a size-independent
property — streams share no state, hence EVERY stream of the batch must equal the same stream run ALONE, bit for bit
(compared on the device) — and a few streams are compared with the oracle directly (the bars of the small tests: mix
and spectrum-of-its-input bit-exact, pitch within 1e-4 relative RMS).
C3 (configs[2]): the pitch node on one hour of stereo, against the oracle over the whole hour.
C4 (configs[3]): the graph on 8 channels x 96 kHz x 60 s as four stereo pairs."""
import numpy as np
import pytest

import orc
from conftest import rel_rms

pytestmark = pytest.mark.gpu


def slice_of(ctx, d, offset, count, dtype=np.float32):
    out = np.empty(count, dtype)
    ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, out.ctypes.data, d.at(offset), out.nbytes))
    ctx.sync()
    return out


def run_graph(ctx, nae, d_a, d_b, n_streams, S, p, first_stream=0, want_wsola=False, pitch_pad=0):
    pl = ctx.stretch_plan(1.0, p, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_mix, d_pitch, d_spec = ctx.empty(n_streams * S * 2), ctx.empty(n_streams * pl.out_len * 2 + pitch_pad), ctx.empty(n_streams * F * 2 * 513)
    g = nae.Graph4()
    g.in_a = nae.Sig.interleaved(d_a.at(first_stream * S * 2), S, 2)
    g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
    g.vol_a = g.vol_b = 0.5
    g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
    g.rate, g.pitch = 1.0, p
    g.pitch_out = nae.Sig.interleaved(d_pitch.at(pitch_pad), pl.out_len, 2)
    g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
    g.S, g.n_streams = S, n_streams
    ctx.graph4(g)
    d_w = None
    if want_wsola:
        wpl = ctx.wsola_plan(48000, 2, 1.0, p, S)
        d_w = ctx.empty(n_streams * wpl.out_len * 2)
        ctx.wsola_block(48000, 1.0, p, g.mix_out, S, 2, n_streams, nae.Sig.interleaved(d_w.ptr, wpl.out_len, 2))
    return d_mix, d_pitch, d_spec, d_w, pl, F


def test_c5_batch_equals_streams_run_alone_and_the_oracle(ctx, nae):
    """EVERY one of the 1024 streams of the batch equals the same stream run alone, bit for bit, in all four outputs
    (compared on the device: nae_debug_diff_u32); three of them are also compared with the oracle."""
    n_streams, S, p = 1024, 480000, 2 ** (3 / 12)
    d_a, d_b = ctx.empty(n_streams * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, 0, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    d_mix, d_pitch, d_spec, d_w, pl, F = run_graph(ctx, nae, d_a, d_b, n_streams, S, p, want_wsola=True)
    assert pl.out_len == S and F == 1872
    wlen = ctx.wsola_plan(48000, 2, 1.0, p, S).out_len
    # --- every stream on its own (1-stream launches take other code paths: no stream grouping, time tiles + pass 1)
    d_cnt = ctx.empty(4, np.uint64).zero()
    for s in range(n_streams):
        m1, p1, s1, w1, _, _ = run_graph(ctx, nae, d_a, d_b, 1, S, p, first_stream=s, want_wsola=True)
        ctx.diff_words(d_mix.at(s * S * 2), m1.ptr, S * 2, d_cnt.at(0))
        ctx.diff_words(d_pitch.at(s * pl.out_len * 2), p1.ptr, pl.out_len * 2, d_cnt.at(1))
        ctx.diff_words(d_spec.at(s * F * 2 * 513), s1.ptr, F * 2 * 513, d_cnt.at(2))
        ctx.diff_words(d_w.at(s * wlen * 2), w1.ptr, wlen * 2, d_cnt.at(3))
        for d in (m1, p1, s1, w1):
            d.free()                                   # (synchronises the stream first)
    assert d_cnt.download().tolist() == [0, 0, 0, 0], "words differing between batch and lone runs: mix, pitch, spectrum, wsola"
    # --- three streams against the oracle
    b = orc.fill_uniform(S * 2, orc.stream_seed(0, 1))
    for s in (0, 517, 1023):
        mix = slice_of(ctx, d_mix, s * S * 2, S * 2)
        pitch = slice_of(ctx, d_pitch, s * pl.out_len * 2, pl.out_len * 2)
        spec = slice_of(ctx, d_spec, s * F * 2 * 513, F * 2 * 513)
        wsola = slice_of(ctx, d_w, s * wlen * 2, wlen * 2)
        a = orc.fill_uniform(S * 2, orc.stream_seed(s, 0))
        L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
        assert np.array_equal(mix[:S], L) and np.array_equal(mix[S:], R)
        mixed = np.stack([L, R], 1).reshape(-1)
        assert rel_rms(pitch, orc.stretch(mixed, 2, 1.0, p)) <= 1e-4
        assert np.array_equal(spec.view(np.uint32), orc.spectrum(pitch, 2).reshape(-1).view(np.uint32))
        assert np.array_equal(wsola.view(np.uint32), orc.st_process(mixed, 2, 48000, 1.0, p).view(np.uint32))
    for d in (d_a, d_b, d_mix, d_pitch, d_spec, d_w, d_cnt):
        d.free()


def test_c5_eighth_of_the_job_128_streams(ctx, nae):
    """What one rank of the 8-GPU job owns (BASELINE.json configs[4]: 1024 streams over 8 GPUs): 128 streams are 256
    stream-channels = one workgroup per CU, so the vocoder runs frame-interleaved (four consecutive frames of a stream-channel per
    step, no time tiles, nothing analysed twice), while a lone stream is cut into time tiles on top (pass 1 + scan + carried
    phases).  Every stream equals its lone run bit for bit; three streams are compared with the oracle."""
    n_streams, S, p, first = 128, 480000, 2 ** (3 / 12), 896            # rank 7 of 8 owns streams 896..1023
    d_a, d_b = ctx.empty(n_streams * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, first, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    d_mix, d_pitch, d_spec, _, pl, F = run_graph(ctx, nae, d_a, d_b, n_streams, S, p)
    d_cnt = ctx.empty(3, np.uint64).zero()
    for s in range(n_streams):
        m1, p1, s1, _, _, _ = run_graph(ctx, nae, d_a, d_b, 1, S, p, first_stream=s)
        ctx.diff_words(d_mix.at(s * S * 2), m1.ptr, S * 2, d_cnt.at(0))
        ctx.diff_words(d_pitch.at(s * pl.out_len * 2), p1.ptr, pl.out_len * 2, d_cnt.at(1))
        ctx.diff_words(d_spec.at(s * F * 2 * 513), s1.ptr, F * 2 * 513, d_cnt.at(2))
        for d in (m1, p1, s1):
            d.free()
    assert d_cnt.download().tolist() == [0, 0, 0]
    b = orc.fill_uniform(S * 2, orc.stream_seed(0, 1))
    for s in (0, 61, 127):
        mix = slice_of(ctx, d_mix, s * S * 2, S * 2)
        pitch = slice_of(ctx, d_pitch, s * pl.out_len * 2, pl.out_len * 2)
        spec = slice_of(ctx, d_spec, s * F * 2 * 513, F * 2 * 513)
        a = orc.fill_uniform(S * 2, orc.stream_seed(first + s, 0))
        L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
        assert np.array_equal(mix[:S], L) and np.array_equal(mix[S:], R)
        ref_p = orc.stretch(np.stack([L, R], 1).reshape(-1), 2, 1.0, p)
        assert rel_rms(pitch, ref_p) <= 1e-4
        assert np.array_equal(spec.view(np.uint32), orc.spectrum(pitch, 2).reshape(-1).view(np.uint32))
        assert rel_rms(spec, orc.spectrum(ref_p, 2).reshape(-1)) <= 1e-4
    for d in (d_a, d_b, d_mix, d_pitch, d_spec, d_cnt):
        d.free()


def test_c5_half_of_the_job_512_streams(ctx, nae):
    """What one rank of a 2-GPU job owns (512 streams = 1024 stream-channels = one workgroup per CU, one frame per step): the vocoder then runs
    its one-barrier pipeline (pv_flow_kernel: doubled hand-off buffers, dense block stores through the used-up buffer's tail).  On 2 s per
    stream: all four outputs equal, bit for bit, those of a context that keeps the two-barrier pipeline (`debug_set("pv_flow", 0)`); the same with the pitch
    output 8 bytes off its 16-byte alignment (both kernels then fall back to dword stores); two streams against the oracle."""
    import os
    n_streams, S, p, first = 512, 96000, 2 ** (3 / 12), 512              # rank 1 of 2 owns streams 512..1023
    d_a, d_b = ctx.empty(n_streams * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, first, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    ctx.prof_reset(); ctx.prof_enable(True)
    d_mix, d_pitch, d_spec, _, pl, F = run_graph(ctx, nae, d_a, d_b, n_streams, S, p)
    ctx.sync(); ctx.prof_enable(False)
    assert "pv_flow_kernel" in ctx.prof_report(), sorted(ctx.prof_report())
    _, d_pitch_off, d_spec_off, _, _, _ = run_graph(ctx, nae, d_a, d_b, n_streams, S, p, pitch_pad=2)
    d_cnt = ctx.empty(5, np.uint64).zero()
    ctx.diff_words(d_pitch.ptr, d_pitch_off.at(2), n_streams * pl.out_len * 2, d_cnt.at(3))
    ctx.diff_words(d_spec.ptr, d_spec_off.ptr, n_streams * F * 2 * 513, d_cnt.at(4))
    if True:
        with nae.Context(0) as c2:
            c2.debug_set("pv_flow", 0)
            ctx.sync()
            c2.prof_reset(); c2.prof_enable(True)
            m2, p2, s2, _, _, _ = run_graph(c2, nae, d_a, d_b, n_streams, S, p)
            c2.sync(); c2.prof_enable(False)
            assert "pv_pipe_kernel" in c2.prof_report() and "pv_flow_kernel" not in c2.prof_report()
            ctx.diff_words(d_mix.ptr, m2.ptr, n_streams * S * 2, d_cnt.at(0))
            ctx.diff_words(d_pitch.ptr, p2.ptr, n_streams * pl.out_len * 2, d_cnt.at(1))
            ctx.diff_words(d_spec.ptr, s2.ptr, n_streams * F * 2 * 513, d_cnt.at(2))
            assert d_cnt.download().tolist() == [0, 0, 0, 0, 0]
    b = orc.fill_uniform(S * 2, orc.stream_seed(0, 1))
    for s in (0, 511):
        pitch = slice_of(ctx, d_pitch, s * pl.out_len * 2, pl.out_len * 2)
        a = orc.fill_uniform(S * 2, orc.stream_seed(first + s, 0))
        L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
        assert rel_rms(pitch, orc.stretch(np.stack([L, R], 1).reshape(-1), 2, 1.0, p)) <= 1e-4
    for d in (d_a, d_b, d_mix, d_pitch, d_spec, d_pitch_off, d_spec_off, d_cnt):
        d.free()


def test_c3_one_hour_stereo_pitch(ctx, nae):
    """BASELINE.json configs[2]: the pitch node (+3 semitones, phase vocoder) on ONE stream of 1 h of 48 kHz stereo
    (172.8 M sample-frames, 1.38 GB): length, finiteness, the whole output against the oracle (relative RMS <= 1e-4, also
    on the first and the last two seconds alone: Q0.32 phases carried over 675 000 frames and ~1023 tiles per channel), and the start of
    the run against a run on a prefix (time tiles are cut differently)."""
    L, ch, p = 3600 * 48000, 2, 2 ** (3 / 12)
    d_x = ctx.empty(L * ch)
    ctx.fill_uniform(d_x.ptr, L * ch, L * ch, 1, 3, 0)
    pl = ctx.stretch_plan(1.0, p, L)
    assert pl.out_len == L
    d_y = ctx.empty(pl.out_len * ch)
    ctx.stretch_block(1.0, p, nae.Sig.interleaved(d_x.ptr, L, ch), L, ch, 1, nae.Sig.interleaved(d_y.ptr, pl.out_len, ch))
    got = d_y.download()
    assert np.isfinite(got).all()
    x = orc.fill_uniform(L * ch, orc.stream_seed(3, 0))
    ref = orc.stretch(x, ch, 1.0, p)                       # about a minute of one host core
    assert ref.size == got.size
    assert rel_rms(got, ref) <= 1e-4
    w = 2 * 48000 * ch
    assert rel_rms(got[:w], ref[:w]) <= 1e-4 and rel_rms(got[-w:], ref[-w:]) <= 1e-4
    # prefix run (its tiles end elsewhere); compared away from its end
    n_short = 30 * 48000
    d_s = ctx.empty(n_short * ch)
    ctx.stretch_block(1.0, p, nae.Sig.interleaved(d_x.ptr, n_short, ch), n_short, ch, 1, nae.Sig.interleaved(d_s.ptr, n_short, ch))
    short = d_s.download()
    k = 25 * 48000 * ch
    assert rel_rms(got[:k], short[:k]) <= 1e-6
    # the SoundTouch-shaped chain on a length its oracle covers in seconds (10 min): bit-exact
    n_w = 600 * 48000
    wpl = ctx.wsola_plan(48000, ch, 1.0, p, n_w)
    d_w = ctx.empty(wpl.out_len * ch)
    ctx.wsola_block(48000, 1.0, p, nae.Sig.interleaved(d_x.ptr, n_w, ch), n_w, ch, 1, nae.Sig.interleaved(d_w.ptr, wpl.out_len, ch))
    ref_w = orc.st_process(x[: n_w * ch], ch, 48000, 1.0, p)
    assert np.array_equal(d_w.download().view(np.uint32), ref_w.view(np.uint32))
    for d in (d_x, d_y, d_s, d_w):
        d.free()


def test_c4_four_stereo_pairs_96k_graph(ctx, nae):
    """BASELINE.json configs[3]: the 4-node graph on 8 channels x 96 kHz x 60 s, taken as 4 independent stereo pairs (the
    reference handles mono / stereo only; nothing in the graph depends on the sample rate).  Every pair equals
    the pair run alone bit for bit; pair 2 is compared with the oracle node by node."""
    n_pairs, S, p = 4, 60 * 96000, 2 ** (3 / 12)
    d_a, d_b = ctx.empty(n_pairs * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_pairs, 40, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 40, 1)
    d_mix, d_pitch, d_spec, _, pl, F = run_graph(ctx, nae, d_a, d_b, n_pairs, S, p)
    assert pl.out_len == S and F == (S - 1024) // 256 + 1
    d_cnt = ctx.empty(3, np.uint64).zero()
    for s in range(n_pairs):
        m1, p1, s1, _, _, _ = run_graph(ctx, nae, d_a, d_b, 1, S, p, first_stream=s)
        ctx.diff_words(d_mix.at(s * S * 2), m1.ptr, S * 2, d_cnt.at(0))
        ctx.diff_words(d_pitch.at(s * pl.out_len * 2), p1.ptr, pl.out_len * 2, d_cnt.at(1))
        ctx.diff_words(d_spec.at(s * F * 2 * 513), s1.ptr, F * 2 * 513, d_cnt.at(2))
        for d in (m1, p1, s1):
            d.free()
    assert d_cnt.download().tolist() == [0, 0, 0]
    s = 2
    a = orc.fill_uniform(S * 2, orc.stream_seed(40 + s, 0))
    b = orc.fill_uniform(S * 2, orc.stream_seed(40, 1))
    mix = slice_of(ctx, d_mix, s * S * 2, S * 2)
    pitch = slice_of(ctx, d_pitch, s * pl.out_len * 2, pl.out_len * 2)
    spec = slice_of(ctx, d_spec, s * F * 2 * 513, F * 2 * 513)
    L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
    assert np.array_equal(mix[:S], L) and np.array_equal(mix[S:], R)
    ref_p = orc.stretch(np.stack([L, R], 1).reshape(-1), 2, 1.0, p)
    assert np.isfinite(pitch).all() and rel_rms(pitch, ref_p) <= 1e-4
    assert np.array_equal(spec.view(np.uint32), orc.spectrum(pitch, 2).reshape(-1).view(np.uint32))
    assert rel_rms(spec, orc.spectrum(ref_p, 2).reshape(-1)) <= 1e-4
    for d in (d_a, d_b, d_mix, d_pitch, d_spec, d_cnt):
        d.free()
