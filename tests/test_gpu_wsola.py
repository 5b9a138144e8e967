"""GPU parity, K7 option A: the SoundTouch-shaped WSOLA + anti-alias FIR + cubic transposer chain,
through the C ABI.  Bar: BIT-EXACT against oracle/orc_wsola.c — samples and every chosen overlap offset — because the
arg-max over candidate offsets makes any difference in floating-point order audible as a different splice.
(Versus SoundTouch 2.3.2 itself the chain is PARITY UNPINNED: see the oracle's header.)"""
import ctypes as C

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def gpu_wsola(ctx, nae, x, ch, sr, rate, pitch, n_streams=1, planar=False, want_offsets=False):
    """x: [n_streams][L][ch] interleaved (or [n_streams][ch][L] planar) -> ([n_streams][out_len*ch], offsets)"""
    L = x.size // (ch * n_streams)
    pl = ctx.wsola_plan(sr, ch, rate, pitch, L)
    d_x = ctx.array(x if x.size else np.zeros(1, np.float32))
    d_y = ctx.empty(max(1, n_streams * pl.out_len * ch))
    n_off = max(int(pl.n_seq) - 1, 0)
    d_off = ctx.empty(max(1, n_streams * n_off), np.int32) if want_offsets else None
    src = nae.Sig.planar(d_x.ptr, L, ch) if planar else nae.Sig.interleaved(d_x.ptr, L, ch)
    dst = nae.Sig.interleaved(d_y.ptr, pl.out_len, ch)
    ctx.wsola_block(sr, rate, pitch, src, L, ch, n_streams, dst, d_off.ptr if d_off else 0)
    y = d_y.download()[: n_streams * pl.out_len * ch].reshape(n_streams, -1)
    offs = d_off.download()[: n_streams * n_off].reshape(n_streams, n_off) if d_off else None
    d_x.free(); d_y.free()
    if d_off:
        d_off.free()
    return (y, offs, pl) if want_offsets else (y, pl)


CASES = [
    (48000, 2, 1.0, 2 ** (3 / 12), 60000),    # pitch up: stretcher, filter, transposer
    (48000, 2, 1.0, 2 ** (-4 / 12), 50000),   # pitch down: transposer, filter, stretcher
    (48000, 2, 1.5, 1 / 1.5, 70000),          # tempo only (rate_eff == 1): filter, transposer, stretcher
    (48000, 2, 0.8, 1 / 0.8, 40000),
    (48000, 2, 1.3, 1.0, 30000),              # plain rate change (tempo 1 still runs the stretcher)
    (48000, 2, 1.0, 1.0, 20000),
    (48000, 1, 1.0, 2 ** (3 / 12), 60000),    # mono: generic FIR (double accumulation), mono cross-fade
    (44100, 1, 0.75, 1.0, 30000),
    (44100, 2, 1.0, 2 ** (5 / 12), 44100),
    (8000, 2, 1.0, 1.25, 16000),              # smallest windows (overlap 64)
    (22050, 1, 1.0, 0.8, 22050),              # overlap 176: group count not a multiple of 4
]


@pytest.mark.parametrize("sr,ch,rate,pitch,L", CASES)
def test_block_is_bit_exact(ctx, nae, sr, ch, rate, pitch, L):
    n_streams = 3
    x = orc.fill_uniform(n_streams * L * ch, 1000 + L)
    y, offs, pl = gpu_wsola(ctx, nae, x, ch, sr, rate, pitch, n_streams, want_offsets=True)
    assert pl.n_seq >= 3
    for s in range(n_streams):
        ref, ref_offs = orc.st_process(x.reshape(n_streams, -1)[s], ch, sr, rate, pitch, want_offsets=True)
        assert ref.size == y[s].size
        assert np.array_equal(offs[s], ref_offs), (s, np.flatnonzero(offs[s] != ref_offs)[:5])
        assert np.array_equal(y[s].view(np.uint32), ref.view(np.uint32)), (s, int(np.count_nonzero(y[s] != ref)))


@pytest.mark.parametrize("nc", [1, 2, 4])
def test_every_search_shape_is_bit_exact(nae, nc):
    """the stretcher kernel has three shapes (4 / 2 / 1 candidate offsets per thread, picked by batch size); each must give the oracle's offsets
    and samples.  The stereo 48-kHz cases of shape 4 run the instantiation with a compile-time LDS row stride, the others the generic one."""
    c = nae.Context(0)
    c.debug_set("td_nc", nc)
    try:
        for sr, ch, rate, pitch, L in ((48000, 2, 1.0, 2 ** (3 / 12), 40000), (48000, 1, 1.0, 0.8, 30000),
                                       (22050, 1, 1.0, 0.8, 22050), (8000, 2, 1.2, 1.0, 12000), (44100, 2, 1.0, 2 ** (-4 / 12), 30000),
                                       (48000, 2, 2.0, 0.5, 40000), (48000, 2, 0.5, 2.0, 40000)):
            x = orc.fill_uniform(2 * L * ch, 50 + nc)
            y, offs, _ = gpu_wsola(c, nae, x, ch, sr, rate, pitch, 2, want_offsets=True)
            for s in range(2):
                ref, ref_offs = orc.st_process(x.reshape(2, -1)[s], ch, sr, rate, pitch, want_offsets=True)
                assert np.array_equal(offs[s], ref_offs), (nc, sr, ch, s)
                assert np.array_equal(y[s].view(np.uint32), ref.view(np.uint32)), (nc, sr, ch, s)
    finally:
        c.close()


def test_tonal_input_and_planar_source(ctx, nae):
    """a tonal signal has many near-equal correlation peaks: the exact evaluation order decides the splice"""
    sr, ch, L = 48000, 2, 96000
    t = np.arange(L) / sr
    x = np.stack([0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1234.5 * t),
                  0.4 * np.sin(2 * np.pi * 660 * t + 1.0)], 1).astype(np.float32)
    for rate, pitch in ((1.0, 2 ** (3 / 12)), (1.25, 0.8)):
        ref, ref_offs = orc.st_process(x.reshape(-1), ch, sr, rate, pitch, want_offsets=True)
        y, offs, _ = gpu_wsola(ctx, nae, x.reshape(-1), ch, sr, rate, pitch, want_offsets=True)
        assert np.array_equal(offs[0], ref_offs)
        assert np.array_equal(y[0], ref)
        yp, _ = gpu_wsola(ctx, nae, np.ascontiguousarray(x.T).reshape(-1), ch, sr, rate, pitch, planar=True)
        assert np.array_equal(yp[0], ref)


def test_edge_inputs(ctx, nae):
    sr, ch = 48000, 2
    # shorter than one sequence: everything comes out of the flush padding
    for L in (0, 1, 500, 4000):
        x = orc.fill_uniform(max(L, 1) * ch, 3)[: L * ch]
        ref = orc.st_process(x, ch, sr, 1.0, 1.2)
        y, pl = gpu_wsola(ctx, nae, x, ch, sr, 1.0, 1.2)
        assert y.shape[1] == ref.size == pl.out_len * ch
        assert np.array_equal(y[0], ref), L
    # silence: the normaliser falls back to 1, every score ties at the weight curve's maximum
    z = np.zeros(30000 * ch, np.float32)
    ref, ref_offs = orc.st_process(z, ch, sr, 1.0, 1.2, want_offsets=True)
    y, offs, _ = gpu_wsola(ctx, nae, z, ch, sr, 1.0, 1.2, want_offsets=True)
    assert np.array_equal(offs[0], ref_offs) and not y.any()
    # non-finite input: a NaN score at offset 0 can never be beaten; elsewhere it never wins
    x = orc.fill_uniform(30000 * ch, 8)
    x[12345] = np.nan
    x[40001] = np.inf
    ref, ref_offs = orc.st_process(x, ch, sr, 1.0, 1.2, want_offsets=True)
    y, offs, _ = gpu_wsola(ctx, nae, x, ch, sr, 1.0, 1.2, want_offsets=True)
    assert np.array_equal(offs[0], ref_offs)
    assert np.array_equal(y[0].view(np.uint32), ref.view(np.uint32))


def stream_wsola(ctx, x, ch, sr, rate, pitch, put_sizes, recv_chunk=3456, device_put=False, pinned=None):
    """drive the handle the way audio-velocity.cpp:344-440 drives SoundTouch: put a frame, drain what is ready"""
    lib = ctx.lib
    L = x.size // ch
    h = C.c_void_p()
    assert lib.nae_wsola_create(ctx.h, sr, ch, rate, pitch, C.byref(h)) == 0
    outs, early, pos, i = [], 0, 0, 0
    d_x = ctx.array(x) if device_put else None

    def drain(limit):
        while lib.nae_wsola_available(h) > limit:
            buf = np.empty(recv_chunk * ch, np.float32)
            got = C.c_size_t()
            assert lib.nae_wsola_receive_host(h, buf.ctypes.data, recv_chunk, C.byref(got)) == 0
            outs.append(buf[: got.value * ch])

    while pos < L:
        n = min(put_sizes[i % len(put_sizes)], L - pos)
        i += 1
        if device_put:
            assert lib.nae_wsola_put(h, d_x.at(pos * ch), n) == 0
        elif pinned is not None:
            # ONE page-locked staging buffer, overwritten for every put: legal because put_host returns only when the
            # copy has left the buffer (with pinned memory hipMemcpyAsync is truly asynchronous)
            pinned[: n * ch] = x[pos * ch:(pos + n) * ch]
            assert lib.nae_wsola_put_host(h, pinned.ctypes.data, n) == 0
            pinned[: n * ch] = np.nan
        else:
            chunk = np.ascontiguousarray(x[pos * ch:(pos + n) * ch])
            assert lib.nae_wsola_put_host(h, chunk.ctypes.data, n) == 0
        pos += n
        early = max(early, lib.nae_wsola_available(h))
        drain(int(1152 / rate))
    assert lib.nae_wsola_flush(h) == 0
    assert lib.nae_wsola_put_host(h, x.ctypes.data, 1) == -5          # NAE_ERR_STATE: put after flush
    drain(0)
    assert lib.nae_wsola_destroy(h) == 0
    if d_x is not None:
        d_x.free()
    return (np.concatenate(outs) if outs else np.zeros(0, np.float32)), early


@pytest.mark.parametrize("ch,rate,pitch", [(2, 1.0, 2 ** (3 / 12)), (2, 1.0, 2 ** (-4 / 12)), (2, 1.5, 1 / 1.5), (1, 0.8, 1.0),
                                           (1, 1.0, 1.3)])
def test_streaming_handle_equals_oracle_stream(ctx, nae, ch, rate, pitch):
    """frame-by-frame puts with interleaved receives give the stream the oracle gives for the same call sequence"""
    sr, L = 48000, 50000
    x = orc.fill_uniform(L * ch, 77)
    for sizes, dev in (([1152], False), ([1152, 4096, 37, 9000, 1, 20000], True), ([L], False)):
        y, early = stream_wsola(ctx, x, ch, sr, rate, pitch, sizes, device_put=dev)
        ref = orc.st_process(x, ch, sr, rate, pitch, chunk=sizes)
        assert y.size == ref.size, (sizes, y.size, ref.size)
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), (sizes, int(np.count_nonzero(y != ref)))
        if sizes == [1152]:
            assert early > 0


def test_put_host_from_one_reused_pinned_buffer(ctx, nae):
    """the caller's buffer is free again when nae_wsola_put_host / nae_stretch_put_host return: feeding every chunk through
    the same page-locked buffer (and trashing it right after each put) gives the oracle's stream"""
    sr, ch, L, p = 48000, 2, 60000, 2 ** (3 / 12)
    x = orc.fill_uniform(L * ch, 78)
    pinned = ctx.pinned(4096 * ch)
    y, _ = stream_wsola(ctx, x, ch, sr, 1.0, p, [1152, 4096, 37], pinned=pinned)
    ref = orc.st_process(x, ch, sr, 1.0, p, chunk=[1152, 4096, 37])
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))
    # the phase-vocoder handle
    lib = ctx.lib
    h = C.c_void_p()
    pf = float(np.float32(p))
    assert lib.nae_stretch_create(ctx.h, sr, ch, 1.0, pf, C.byref(h)) == 0
    outs, pos = [], 0
    while pos < L:
        n = min(4096, L - pos)
        pinned[: n * ch] = x[pos * ch:(pos + n) * ch]
        assert lib.nae_stretch_put_host(h, pinned.ctypes.data, n) == 0
        pinned[: n * ch] = np.nan
        pos += n
    assert lib.nae_stretch_flush(h) == 0
    k = lib.nae_stretch_available(h)
    buf, got = np.empty(k * ch, np.float32), C.c_size_t()
    assert lib.nae_stretch_receive_host(h, buf.ctypes.data, k, C.byref(got)) == 0 and got.value == k
    assert lib.nae_stretch_destroy(h) == 0
    ctx.free_pinned(pinned)
    d_x = ctx.array(x)
    pl = ctx.stretch_plan(1.0, pf, L)
    d_y = ctx.empty(pl.out_len * ch)
    ctx.stretch_block(1.0, pf, nae.Sig.interleaved(d_x.ptr, L, ch), L, ch, 1, nae.Sig.interleaved(d_y.ptr, pl.out_len, ch))
    assert np.array_equal(buf.view(np.uint32), d_y.download().view(np.uint32))
    d_x.free(); d_y.free()


def test_create_rejects_what_the_reference_rejects(ctx, nae):
    h = C.c_void_p()
    assert ctx.lib.nae_wsola_create(ctx.h, 96000, 2, 1.0, 1.0, C.byref(h)) == -2      # audio-velocity.cpp:371-379
    assert ctx.lib.nae_wsola_create(ctx.h, 48000, 3, 1.0, 1.0, C.byref(h)) == -1
    assert ctx.lib.nae_wsola_create(ctx.h, 48000, 2, 0.0, 1.0, C.byref(h)) == -1


def test_full_size_properties(ctx, nae):
    """C5 per-stream size (10 s stereo): length rule, a scaled input gives the scaled output (the splice points depend
    on correlation RATIOS only... up to rounding, so compare against the oracle on a prefix instead), and the first
    second is bit-exact versus the oracle run on the whole stream"""
    sr, ch, L = 48000, 2, 480000
    x = orc.fill_uniform(L * ch, 4242)
    y, offs, pl = gpu_wsola(ctx, nae, x, ch, sr, 1.0, 2 ** (3 / 12), want_offsets=True)
    assert pl.out_len == L and y.shape[1] == L * ch
    ref, ref_offs = orc.st_process(x, ch, sr, 1.0, 2 ** (3 / 12), want_offsets=True)
    assert np.array_equal(offs[0], ref_offs)
    assert np.array_equal(y[0].view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name", ["pitch_up3", "tempo_1p25", "pitch_down4", "mono_22k_down", "mono_8k_rate"])
def test_committed_golden(ctx, nae, golden, name):
    """GPU against the committed vectors of the independent numpy restatement (tests/golden/st_numpy.py), without the C
    oracle in the loop"""
    g = golden["wsola_golden"]
    ch, sr, rate, pitch = g[name + "_params"]
    x = g[str(g[name + "_src"])]
    y, offs, _ = gpu_wsola(ctx, nae, x, int(ch), int(sr), float(rate), float(pitch), want_offsets=True)
    assert np.array_equal(y[0].view(np.uint32), g[name].view(np.uint32))
    assert np.array_equal(offs[0], g[name + "_offsets"][: offs.shape[1]])


def test_odd_layouts_equal_clean_layouts(ctx, nae):
    """misaligned bases, padded frame / stream strides and planar destinations select the scalar access paths of the
    three kernels; results must equal the clean-layout results bit for bit"""
    sr, ch, L, n_streams = 48000, 2, 30000, 5
    x = orc.fill_uniform(n_streams * L * ch, 12)
    for rate, pitch in ((1.0, 2 ** (3 / 12)), (1.0, 0.8), (1.25, 0.8)):
        clean, pl = gpu_wsola(ctx, nae, x, ch, sr, rate, pitch, n_streams)
        clean = clean.reshape(n_streams, pl.out_len, ch)
        for (cs_s, fs_s), (cs_d, fs_d), pe in (((1, 3), (1, 2), 1), ((L + 1, 1), (pl.out_len + 3, 1), 3), ((1, 2), (1, 5), 1)):
            ss_s = (L * fs_s if cs_s == 1 else ch * cs_s) + 7
            ss_d = (pl.out_len * fs_d if cs_d == 1 else ch * cs_d) + 3
            src = np.zeros(pe + n_streams * ss_s + 8, np.float32)
            idx_s = np.arange(n_streams)[:, None, None] * ss_s + np.arange(L)[None, :, None] * fs_s + np.arange(ch)[None, None, :] * cs_s + pe
            src[idx_s] = x.reshape(n_streams, L, ch)
            d_src, d_dst = ctx.array(src), ctx.array(np.full(pe + n_streams * ss_d + 8, 5.0, np.float32))
            ctx.wsola_block(sr, rate, pitch, nae.Sig(d_src.at(pe), ss_s, cs_s, fs_s), L, ch, n_streams, nae.Sig(d_dst.at(pe), ss_d, cs_d, fs_d))
            got = d_dst.download()
            idx_d = np.arange(n_streams)[:, None, None] * ss_d + np.arange(pl.out_len)[None, :, None] * fs_d + np.arange(ch)[None, None, :] * cs_d + pe
            assert np.array_equal(got[idx_d].view(np.uint32), clean.view(np.uint32)), (rate, pitch, cs_s, fs_s, cs_d, fs_d)
            mask = np.ones(got.size, bool)
            mask[idx_d] = False
            assert np.all(got[mask] == 5.0)
            d_src.free(); d_dst.free()
