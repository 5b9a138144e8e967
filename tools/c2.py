#!/usr/bin/env python3
"""C2 at a chip-filling size: fused gain on 25000 x 4096 stereo frames (819 MB in, 819 MB out) and the 3-launch chain"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload
nae = naeload.load()
ctx = nae.Context(0)
n3, S2, vol = 25000, 4096, 0.7071
d_x, d_y, d_p = ctx.empty(n3 * S2 * 2), ctx.empty(n3 * S2 * 2), ctx.empty(n3 * S2 * 2)
ctx.fill_uniform(d_x.ptr, n3 * S2 * 2, 0, 1, 0, 0)
def timed(fn, reps=10):
    for _ in range(2): fn()
    ctx.sync(); a, b = ctx.event(), ctx.event(); ctx.record(a)
    for _ in range(reps): fn()
    ctx.record(b); return ctx.elapsed_ms(a, b) / reps
inter = lambda t: nae.Sig.interleaved(t.ptr, S2, 2)
planar = lambda t: nae.Sig.planar(t.ptr, S2, 2)
ms = timed(lambda: ctx.gain_sig(inter(d_x), inter(d_y), S2, 2, n3, vol))
print("fused gain: %.3f ms  %.0f GB/s (16 B per sample-frame)" % (ms, n3 * S2 * 16 / ms / 1e6))
import numpy as np
ms = timed(lambda: ctx.gain(np.float32, [d_x.ptr], [d_y.ptr], n3 * S2 * 2, vol))
print("K1 gain, one 819 MB plane: %.3f ms  %.0f GB/s" % (ms, n3 * S2 * 16 / ms / 1e6))
ms = timed(lambda: ctx.copy_sig(inter(d_x), planar(d_p), S2, 2, n3))
print("split (i2p): %.3f ms  %.0f GB/s" % (ms, n3 * S2 * 16 / ms / 1e6))
ms = timed(lambda: ctx.copy_sig(planar(d_p), inter(d_y), S2, 2, n3))
print("merge (p2i): %.3f ms  %.0f GB/s" % (ms, n3 * S2 * 16 / ms / 1e6))
