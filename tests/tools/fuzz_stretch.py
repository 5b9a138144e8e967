#!/usr/bin/env python3
"""Randomised differential run (GPU box): nae_stretch_block_f32 and nae_spectrum_block_f32 against the CPU oracle over random
rates, pitches, lengths, channel counts, layouts and batch sizes — in particular batches around the vocoder's tile-policy
thresholds (stream-channels 256 / 512 / 1024), where a stream is cut into 1, 2 or several time tiles.
    python tests/tools/fuzz_stretch.py [cases=40] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import naeload
import orc


def rel_rms(a, b):
    d = np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))
    return d / max(np.sqrt(np.mean(b.astype(np.float64) ** 2)), 1e-30)


def main(cases=40, seed=1, ctx=None, nae=None):
    rng = np.random.default_rng(seed)
    if nae is None:
        nae = naeload.load()
    if ctx is None:
        ctx = nae.Context(0)
    worst = 0.0
    for k in range(cases):
        ch = int(rng.choice([1, 2]))
        n_streams = int(rng.choice([1, 2, 5, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513]))
        L = int(rng.integers(1100, 60000))
        rate = float(np.exp(rng.uniform(np.log(0.5), np.log(2.0))))
        pitch = float(2.0 ** rng.uniform(-1.0, 1.0))
        if not (0.25 <= rate * pitch <= 4.0):
            continue
        planar_in, planar_out = bool(rng.integers(2)), bool(rng.integers(2))
        x = (0.5 * rng.uniform(-1, 1, (n_streams, L, ch))).astype(np.float32)
        pl = ctx.stretch_plan(rate, pitch, L)
        flat = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(-1) if planar_in else x.reshape(-1)
        d_x, d_o = ctx.array(flat), ctx.empty(max(1, n_streams * pl.out_len * ch))
        src = nae.Sig.planar(d_x.ptr, L, ch) if planar_in else nae.Sig.interleaved(d_x.ptr, L, ch)
        dst = nae.Sig.planar(d_o.ptr, pl.out_len, ch) if planar_out else nae.Sig.interleaved(d_o.ptr, pl.out_len, ch)
        ctx.stretch_block(rate, pitch, src, L, ch, n_streams, dst)
        out = d_o.download()[: n_streams * pl.out_len * ch]
        out = out.reshape(n_streams, ch, pl.out_len).transpose(0, 2, 1) if planar_out else out.reshape(n_streams, pl.out_len, ch)
        d_x.free(); d_o.free()
        errs = []
        for s in sorted({0, n_streams // 2, n_streams - 1}):
            ref = orc.stretch(x[s].reshape(-1), ch, rate, pitch).reshape(-1, ch)
            assert ref.shape == out[s].shape, (ref.shape, out[s].shape)
            errs.append(rel_rms(out[s], ref) if ref.size else 0.0)
        e = max(errs)
        worst = max(worst, e)
        flag = "" if e <= 1e-4 else "   <-- ABOVE TOLERANCE"
        print(f"case {k:3d}: streams {n_streams:4d} ch {ch} L {L:6d} rate {rate:.3f} pitch {pitch:.3f} "
              f"{'P' if planar_in else 'I'}->{'P' if planar_out else 'I'} out {pl.out_len:6d}  rel-RMS {e:.2e}{flag}", flush=True)
        # spectrum of the same input, stream 0 and last, bit-exact
        T = L
        F = ctx.spectrum_frames(T)
        if F:
            flat2 = x.reshape(-1)
            d_x, d_s = ctx.array(flat2), ctx.empty(n_streams * F * ch * 513)
            ctx.spectrum_block(nae.Sig.interleaved(d_x.ptr, T, ch), T, ch, n_streams, d_s.ptr, F * ch * 513)
            sp = d_s.download().reshape(n_streams, F, ch, 513)
            d_x.free(); d_s.free()
            for s in (0, n_streams - 1):
                ref = orc.spectrum(x[s].reshape(-1), ch)
                assert np.array_equal(sp[s].view(np.uint32), ref.view(np.uint32)), f"spectrum differs: case {k} stream {s}"
    print(f"worst rel-RMS {worst:.2e} over {cases} cases (tolerance 1e-4)")
    assert worst <= 1e-4
    return worst


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
