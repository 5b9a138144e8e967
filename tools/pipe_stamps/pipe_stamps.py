#!/usr/bin/env python3
"""Diagnostic (variant build, never shipped): per-wave s_memtime stamps around the two barriers of eight steps of
pv_pipe_kernel, workgroup 0, while the C5 graph runs.  Build: tools/pipe_stamps/make_variant.sh (a stamped COPY of the kernel source)
Run:   NAE_GPU_LIB=nodey-audio-editor_amd/variants/libnae_gpu_stamps.so python tools/pipe_stamps/pipe_stamps.py [n_streams]   (512: one workgroup per CU)"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import naeload

nae = naeload.load()
ctx = nae.Context(0)
n_streams, S, p = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 480000, 2 ** (3 / 12)
pl = ctx.stretch_plan(1.0, p, S)
F = ctx.spectrum_frames(pl.out_len)
d_a, d_b = ctx.empty(n_streams * S * 2), ctx.empty(S * 2)
ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, 0, 0)
ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
d_mix, d_pitch, d_spec = ctx.empty(n_streams * S * 2), ctx.empty(n_streams * pl.out_len * 2), ctx.empty(n_streams * F * 2 * 513)
g = nae.Graph4()
g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
g.vol_a = g.vol_b = 0.5
g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
g.rate, g.pitch = 1.0, p
g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
g.S, g.n_streams = S, n_streams
for _ in range(int(os.environ.get("STAMP_ITERS", "3"))):     # STAMP_ITERS=400: the clock has settled where bench.py measures
    ctx.graph4(g)
ctx.sync()
st = np.zeros(16 * 8 * 4, np.uint64)
rc = ctx.lib.nae_debug_read_pipe_stamps(st.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
st = st.reshape(16, 8, 4).astype(np.int64)
t0 = st[:, 0, 0].min()
st -= t0
names = ["R1 "] * 4 + ["R2a"] * 4 + ["R2b"] * 4 + ["R3 "] * 4
print("cycles relative to the first stamp; per wave and step: arrive A, leave A, arrive B, leave B")
for step in range(8):
    print(f"-- step {200 + step}")
    for w in range(16):
        a0, a1, b0, b1 = st[w, step]
        nxt = st[w, step + 1, 0] if step < 7 else -1
        print(f"  wave {w:2d} {names[w]} slot{w % 4}: A {a0:7d} -> {a1:7d} (wait {a1 - a0:5d}) | alpha {b0 - a1:5d} | B {b0:7d} -> {b1:7d} (wait {b1 - b0:5d}) | beta {nxt - b1 if nxt >= 0 else -1:5d}")
per_step = (st[:, 7, 0] - st[:, 0, 0]) / 7.0
print("cycles per step per wave:", np.round(per_step).astype(int))
tot = np.zeros(4 * 64, np.uint64)
assert ctx.lib.nae_debug_read_pipe_total(tot.ctypes.data_as(C.c_void_p)) == 0
tot = tot.reshape(64, 4)
tot = tot[tot[:, 2] > 0]                 # fewer than 64 workgroups on the sampled XCD (small batches)
cyc, real, steps = tot[:, 0].astype(np.float64), tot[:, 1].astype(np.float64), tot[:, 2].astype(np.float64)
print("whole loop, R1 wave of 64 sampled workgroups: cycles per step min/median/max =", int((cyc / steps).min()), int(np.median(cyc / steps)), int((cyc / steps).max()),
      "| clock GHz min/median/max = %.2f %.2f %.2f" % tuple(np.percentile(cyc / real * 0.1, [0, 50, 100])), "| loop ms median = %.3f" % (np.median(real) * 1e-5))
xcc = (tot[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
hw = tot[:, 3].astype(np.int64) & 0xffffffff
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"xcc {x}: {m.sum():2d} sampled workgroups, cycles/step " + " ".join(str(int(v)) for v in sorted((cyc / steps)[m])) + "  GHz %.2f" % np.median((cyc / real * 0.1)[m]))
# which workgroups share a CU, and who is the fast one: HW_ID (gfx9) bits 8-11 = CU, 12 = SH, 13-14 = SE
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x3
place = collections.defaultdict(list)
for j in range(len(tot)):
    place[(int(xcc[j]), int(se[j]), int(sh[j]), int(cu[j]))].append((j, int(cyc[j] / steps[j])))
print("workgroups by (xcc, se, sh, cu): [(dispatch index within the XCD, cycles per step), ...]")
for k in sorted(place):
    print("  ", k, place[k])
