#!/usr/bin/env python3
"""End-to-end rate of the C5 graph when the boundary hands over HOST buffers (what the per-frame adapter does):
H2D of the inputs + graph + D2H of the pitch and spectrum outputs, pageable numpy memory through nae_memcpy_*.
Never the headline `value` (bench.py times HBM-resident data); reported in DESIGN.md §9.
   python tools/bench_pcie.py [--streams 256]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import naeload

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
nae = naeload.load()
ctx = nae.Context(0)
n, S, p = a.streams, 480000, 2 ** (3 / 12)
pl = ctx.stretch_plan(1.0, p, S)
F = ctx.spectrum_frames(pl.out_len)
h_a = np.random.default_rng(1).uniform(-1, 1, n * S * 2).astype(np.float32)
h_b = np.random.default_rng(2).uniform(-1, 1, S * 2).astype(np.float32)
h_pitch = np.empty(n * pl.out_len * 2, np.float32)
h_spec = np.empty(n * F * 2 * 513, np.float32)
d_a, d_b = ctx.empty(n * S * 2), ctx.empty(S * 2)
d_mix, d_pitch, d_spec = ctx.empty(n * S * 2), ctx.empty(n * pl.out_len * 2), ctx.empty(n * F * 2 * 513)
g = nae.Graph4()
g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
g.vol_a = g.vol_b = 0.5
g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
g.rate, g.pitch = 1.0, p
g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
g.S, g.n_streams = S, n


def once():
    t = [time.perf_counter()]
    ctx._ck(ctx.lib.nae_memcpy_h2d(ctx.h, d_a.ptr, h_a.ctypes.data, h_a.nbytes))
    ctx._ck(ctx.lib.nae_memcpy_h2d(ctx.h, d_b.ptr, h_b.ctypes.data, h_b.nbytes))
    ctx.sync(); t.append(time.perf_counter())
    ctx.graph4(g)
    ctx.sync(); t.append(time.perf_counter())
    ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, h_pitch.ctypes.data, d_pitch.ptr, h_pitch.nbytes))
    ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, h_spec.ctypes.data, d_spec.ptr, h_spec.nbytes))
    ctx.sync(); t.append(time.perf_counter())
    return [t[i + 1] - t[i] for i in range(3)]


once()
best = min((once() for _ in range(a.reps)), key=sum)
tot = sum(best)
print(json.dumps({"streams": n, "sample_frames": n * S, "h2d_s": best[0], "graph_s": best[1], "d2h_s": best[2],
                  "h2d_GBps": (h_a.nbytes + h_b.nbytes) / best[0] / 1e9, "d2h_GBps": (h_pitch.nbytes + h_spec.nbytes) / best[2] / 1e9,
                  "sample_frames_per_s_incl_pcie": n * S / tot, "sample_frames_per_s_graph_only": n * S / best[1]}))
