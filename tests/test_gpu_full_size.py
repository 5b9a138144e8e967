"""GPU, BASELINE.json's full C5 size on one GPU: 1024 streams x 10 s through input -> mix(2) -> pitch -> spectrum.
The oracle cannot run 1024 streams in seconds, so the whole batch is pinned through a size-independent property —
streams share no state, hence every stream of the batch must equal the same stream run ALONE, bit for bit — and a few
streams are compared with the oracle directly (the bars of the small tests: mix and spectrum-of-its-input bit-exact,
pitch within 1e-4 relative RMS)."""
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


def run_graph(ctx, nae, d_a, d_b, n_streams, S, p, first_stream=0, want_wsola=False):
    pl = ctx.stretch_plan(1.0, p, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_mix, d_pitch, d_spec = ctx.empty(n_streams * S * 2), ctx.empty(n_streams * pl.out_len * 2), ctx.empty(n_streams * F * 2 * 513)
    g = nae.Graph4()
    g.in_a = nae.Sig.interleaved(d_a.at(first_stream * S * 2), S, 2)
    g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
    g.vol_a = g.vol_b = 0.5
    g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
    g.rate, g.pitch = 1.0, p
    g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
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
    n_streams, S, p = 1024, 480000, 2 ** (3 / 12)
    d_a, d_b = ctx.empty(n_streams * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, 0, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    d_mix, d_pitch, d_spec, d_w, pl, F = run_graph(ctx, nae, d_a, d_b, n_streams, S, p, want_wsola=True)
    assert pl.out_len == S and F == 1872
    b = orc.fill_uniform(S * 2, orc.stream_seed(0, 1))
    for s in (0, 517, 1023):
        mix = slice_of(ctx, d_mix, s * S * 2, S * 2)
        pitch = slice_of(ctx, d_pitch, s * pl.out_len * 2, pl.out_len * 2)
        spec = slice_of(ctx, d_spec, s * F * 2 * 513, F * 2 * 513)
        wsola = slice_of(ctx, d_w, s * S * 2, S * 2)
        # --- the same stream on its own (1-stream launches take other code paths: no stream grouping, other tiling)
        m1, p1, s1, w1, _, _ = run_graph(ctx, nae, d_a, d_b, 1, S, p, first_stream=s, want_wsola=True)
        assert np.array_equal(mix, m1.download())
        assert np.array_equal(pitch.view(np.uint32), p1.download().view(np.uint32))
        assert np.array_equal(spec.view(np.uint32), s1.download().view(np.uint32))
        assert np.array_equal(wsola.view(np.uint32), w1.download().view(np.uint32))
        for d in (m1, p1, s1, w1):
            d.free()
        # --- and against the oracle
        a = orc.fill_uniform(S * 2, orc.stream_seed(s, 0))
        L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
        assert np.array_equal(mix[:S], L) and np.array_equal(mix[S:], R)
        mixed = np.stack([L, R], 1).reshape(-1)
        assert rel_rms(pitch, orc.stretch(mixed, 2, 1.0, p)) <= 1e-4
        assert np.array_equal(spec.view(np.uint32), orc.spectrum(pitch, 2).reshape(-1).view(np.uint32))
        assert np.array_equal(wsola.view(np.uint32), orc.st_process(mixed, 2, 48000, 1.0, p).view(np.uint32))
    for d in (d_a, d_b, d_mix, d_pitch, d_spec, d_w):
        d.free()
