#!/usr/bin/env python3
"""r05: per-kernel times of BASELINE configs[2] (one stereo stream, one hour, pitch +3 semitones) and of other few-long-stream shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import naeload
nae = naeload.load()
ctx = nae.Context(0)
for n_streams, seconds in ((1, 3600), (4, 900), (8, 450), (16, 225)):
    L, ch, p = seconds * 48000, 2, 2 ** (3 / 12)
    pl = ctx.stretch_plan(1.0, p, L)
    d_x, d_o = ctx.empty(n_streams * L * ch), ctx.empty(n_streams * pl.out_len * ch)
    ctx.fill_uniform(d_x.ptr, L * ch, L * ch, n_streams, 0, 0)
    src, dst = nae.Sig.interleaved(d_x.ptr, L, ch), nae.Sig.interleaved(d_o.ptr, pl.out_len, ch)
    for _ in range(3):
        ctx.stretch_block(1.0, p, src, L, ch, n_streams, dst)
    ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(5):
        ctx.stretch_block(1.0, p, src, L, ch, n_streams, dst)
    ctx.sync(); ctx.prof_enable(False)
    rep = {k: round(v[0] / max(v[1], 1), 3) for k, v in ctx.prof_report().items()}
    print(f"{n_streams} x {seconds} s:", rep, "sum", round(sum(rep.values()), 3), flush=True)
    d_x.free(); d_o.free()
