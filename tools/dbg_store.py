import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload
nae = naeload.load()
ctx = nae.Context(0)
n, S = 1024, 480000
rate = 2 ** (3 / 12); pitch = 1 / rate
pl = ctx.stretch_plan(rate, pitch, S)
d_in = ctx.empty(n * S * 2); d_out = ctx.empty(n * pl.out_len * 2)
ctx.fill_uniform(d_in.ptr, S * 2, S * 2, n, 0, 0)
src = nae.Sig.planar(d_in.ptr, S, 2)
for name, dst in (("planar out", nae.Sig.planar(d_out.ptr, pl.out_len, 2)), ("interleaved out", nae.Sig.interleaved(d_out.ptr, pl.out_len, 2))):
    for _ in range(2): ctx.stretch_block(rate, pitch, src, S, 2, n, dst)
    ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3): ctx.stretch_block(rate, pitch, src, S, 2, n, dst)
    ctx.prof_enable(False)
    print(name, pl.frames, {k: round(v[0] / v[1], 3) for k, v in ctx.prof_report().items()})
