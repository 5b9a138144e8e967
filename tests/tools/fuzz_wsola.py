#!/usr/bin/env python3
"""Randomised differential run (GPU box): nae_wsola_block_f32 against oracle/orc_wsola.c — samples AND chosen overlap offsets,
bit for bit — over random sample rates, rate / pitch settings, lengths, channel counts, layouts and batch sizes (640 streams
is where the search switches from 2 to 4 candidates per thread and starts keeping the next window from the copied frames).
    python tests/tools/fuzz_wsola.py [cases=30] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import naeload
import orc


def main(cases=30, seed=1, ctx=None, nae=None):
    rng = np.random.default_rng(seed)
    if nae is None:
        nae = naeload.load()
    if ctx is None:
        ctx = nae.Context(0)
    done = 0
    for k in range(cases):
        sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        ch = int(rng.choice([1, 2]))
        n_streams = int(rng.choice([1, 2, 3, 7, 639, 640, 700]))
        L = int(rng.integers(sr // 4, 3 * sr if n_streams < 100 else sr))
        rate = float(np.exp(rng.uniform(np.log(0.6), np.log(1.8))))
        pitch = float(2.0 ** rng.uniform(-0.8, 0.8))
        planar = bool(rng.integers(2))
        kind = rng.choice(["noise", "tone", "silence+click"])
        if kind == "noise":
            x = (0.7 * rng.uniform(-1, 1, (n_streams, L, ch))).astype(np.float32)
        elif kind == "tone":
            t = np.arange(L)[None, :, None] / sr
            f = rng.uniform(80, 3000, (n_streams, 1, ch))
            x = (0.6 * np.sin(2 * np.pi * f * t)).astype(np.float32)
        else:
            x = np.zeros((n_streams, L, ch), np.float32)
            x[:, L // 3, :] = 1.0
        try:
            pl = ctx.wsola_plan(sr, ch, rate, pitch, L)
        except Exception as e:                           # settings outside the supported tempo / rate range
            print(f"case {k:3d}: sr {sr} rate {rate:.3f} pitch {pitch:.3f}: refused ({str(e)[:60]})")
            continue
        flat = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(-1) if planar else x.reshape(-1)
        d_x = ctx.array(flat)
        d_y = ctx.empty(max(1, n_streams * pl.out_len * ch))
        n_off = max(int(pl.n_seq) - 1, 0)
        d_off = ctx.empty(max(1, n_streams * n_off), np.int32)
        src = nae.Sig.planar(d_x.ptr, L, ch) if planar else nae.Sig.interleaved(d_x.ptr, L, ch)
        ctx.wsola_block(sr, rate, pitch, src, L, ch, n_streams, nae.Sig.interleaved(d_y.ptr, pl.out_len, ch), d_off.ptr)
        y = d_y.download()[: n_streams * pl.out_len * ch].reshape(n_streams, -1)
        offs = d_off.download()[: n_streams * n_off].reshape(n_streams, n_off)
        d_x.free(); d_y.free(); d_off.free()
        for s in sorted({0, n_streams // 2, n_streams - 1}):
            ref, ref_offs = orc.st_process(x[s].reshape(-1), ch, sr, rate, pitch, want_offsets=True)
            assert ref.size == y[s].size, (k, s, ref.size, y[s].size)
            assert np.array_equal(offs[s], ref_offs), f"case {k} stream {s}: offsets differ at {np.flatnonzero(offs[s] != ref_offs)[:5]}"
            assert np.array_equal(y[s].view(np.uint32), ref.view(np.uint32)), f"case {k} stream {s}: {int(np.count_nonzero(y[s] != ref))} samples differ"
        done += 1
        print(f"case {k:3d}: sr {sr:5d} streams {n_streams:3d} ch {ch} L {L:6d} rate {rate:.3f} pitch {pitch:.3f} {kind:13s} "
              f"{'planar' if planar else 'interl'}  sequences {int(pl.n_seq):3d}  bit-exact", flush=True)
    print(f"{done} cases bit-exact (samples and offsets)")
    return done


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
