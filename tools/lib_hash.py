#!/usr/bin/env python3
"""SHA-256 of every result buffer of the 4-node graph on a few shapes (vocoder: one frame per step, frame-interleaved, tiled; the WSOLA chain): run it
once per library build (NAE_GPU_LIB=...) and compare the lines — equal lines = equal bits.   python tools/lib_hash.py > a.txt"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload

nae = naeload.load()
for n, S, semis in ((6, 48000, 3.0), (5, 30011, 7.0), (9, 12345, -4.0), (1300, 6000, 3.0), (300, 20000, 3.0), (1, 600000, 3.0), (40, 100000, 5.0)):
    p = 2 ** (semis / 12)
    with nae.Context(0) as ctx:
        pl = ctx.stretch_plan(1.0, p, S)
        F = ctx.spectrum_frames(pl.out_len)
        d_a, d_b = ctx.empty(n * S * 2), ctx.empty(S * 2)
        ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n, 0, 0)
        ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
        d_mix, d_pitch, d_spec = ctx.empty(n * S * 2), ctx.empty(n * pl.out_len * 2), ctx.empty(n * F * 2 * 513)
        d_pitch.zero(); d_spec.zero()
        g = nae.Graph4()
        g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
        g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
        g.vol_a = g.vol_b = 0.5
        g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
        g.rate, g.pitch = 1.0, p
        g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
        g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
        g.S, g.n_streams = S, n
        ctx.graph4(g)
        h = [hashlib.sha256(d.download().tobytes()).hexdigest()[:16] for d in (d_mix, d_pitch, d_spec)]
        print(f"n={n} S={S} st={semis:+g} mix {h[0]} pitch {h[1]} spectrum {h[2]}", flush=True)
