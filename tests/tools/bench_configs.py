#!/usr/bin/env python3
"""Time every BASELINE.json config on one MI355X and print a markdown table (profiles/r0N_configs.md is its output).
Lives under tests/ because it checks parity against, and times, the CPU oracle (only tests/, smoke() and bench.py may).
bench.py remains the contract benchmark (C5); this tool covers C1-C4 and the per-node kernels."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import naeload
import orc

nae = naeload.load()
ctx = nae.Context(0)
PEAK = 8000.0
rows = []


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ctx.sync()
    a, b = ctx.event(), ctx.event()
    ctx.record(a)
    for _ in range(reps):
        fn()
    ctx.record(b)
    return ctx.elapsed_ms(a, b) / reps


def rr(x, y):
    return float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)) / max(np.sqrt(np.mean(y.astype(np.float64) ** 2)), 1e-30))


def add(cfg, what, frames, ms, alg_bytes_per_frame, parity, cpu=None):
    gbs = frames * alg_bytes_per_frame / (ms * 1e-3) / 1e9
    rows.append(f"| {cfg} | {what} | {frames:.4g} | {ms:.3f} | {frames / (ms * 1e-3):.3e} | {alg_bytes_per_frame:g} | {gbs:.0f} | {gbs / PEAK:.3f} | {parity} | {cpu or ''} |")


# ---------------------------------------------------------------- C1: 2-input mix, 10 s stereo
S = 480000
a, b = orc.fill_uniform(2 * S, orc.stream_seed(0, 0)), orc.fill_uniform(2 * S, orc.stream_seed(0, 1))
d_a, d_b, d_o = ctx.array(a), ctx.array(b), ctx.empty(2 * S)
ins = [nae.Sig.interleaved(d_a.ptr, S, 2), nae.Sig.interleaved(d_b.ptr, S, 2)]
out = nae.Sig.planar(d_o.ptr, S, 2)
ms = timed(lambda: ctx.amix_sig(ins, [0.5, 0.5], out, S, 1), reps=50)
t0 = time.perf_counter()
for _ in range(5):
    L, R = orc.amix([a[0::2], b[0::2]], [a[1::2], b[1::2]], [0.5, 0.5])
cpu_s = (time.perf_counter() - t0) / 5
got = d_o.download()
ok = np.array_equal(got[:S], L) and np.array_equal(got[S:], R)
add("C1", "amix(2), 1 stream x 10 s (single launch; launch-bound at this size)", S, ms, 24, "bit-exact" if ok else "MISMATCH",
    f"oracle 1 thread: {S / cpu_s:.3e} sf/s (incl. numpy deinterleave)")
for x in (d_a, d_b, d_o):
    x.free()

# ---------------------------------------------------------------- C2: split -> gain -> merge, 1000 x 4096
n, S2, vol = 1000, 4096, 0.7071
x = orc.fill_uniform(n * S2 * 2, orc.stream_seed(0))
d_x, d_pl, d_g, d_y = ctx.array(x), ctx.empty(x.size), ctx.empty(x.size), ctx.empty(x.size)
inter = lambda t: nae.Sig.interleaved(t.ptr, S2, 2)
planar = lambda t: nae.Sig.planar(t.ptr, S2, 2)


def chain():
    ctx.copy_sig(inter(d_x), planar(d_pl), S2, 2, n)
    ctx.gain(np.float32, [d_pl.ptr], [d_g.ptr], x.size, vol)
    ctx.copy_sig(planar(d_g), inter(d_y), S2, 2, n)


ms = timed(chain, reps=50)
ref = orc.change_volume([x], vol)[0]
ok = np.array_equal(d_y.download(), ref)
t0 = time.perf_counter()
for _ in range(5):
    pl = orc.deinterleave(x, 2)
    gg = orc.change_volume(pl, vol)
    orc.interleave(gg)
cpu_s = (time.perf_counter() - t0) / 5
add("C2", "split -> gain -> merge (3 launches), 1000 x 4096", n * S2, ms, 48, "bit-exact" if ok else "MISMATCH", f"oracle 1 thread: {n * S2 / cpu_s:.3e} sf/s")
ms = timed(lambda: ctx.gain_sig(inter(d_x), inter(d_y), S2, 2, n, vol), reps=50)
ok = np.array_equal(d_y.download(), ref)
add("C2", "same, fused into one launch", n * S2, ms, 16, "bit-exact" if ok else "MISMATCH")
for t in (d_x, d_pl, d_g, d_y):
    t.free()
# the same chain at a size that fills the chip (HBM-bound regime): 1000 x 409600
n3 = 25000
d_x, d_y = ctx.empty(n3 * S2 * 2), ctx.empty(n3 * S2 * 2)
ctx.fill_uniform(d_x.ptr, n3 * S2 * 2, 0, 1, 0, 0)
ms = timed(lambda: ctx.gain_sig(nae.Sig.interleaved(d_x.ptr, S2, 2), nae.Sig.interleaved(d_y.ptr, S2, 2), S2, 2, n3, vol), reps=10)
add("C2 x25", "fused gain, 25000 x 4096 (819 MB in)", n3 * S2, ms, 16, "n/a")
d_x.free(); d_y.free()

# ---------------------------------------------------------------- C3: pitch node on 1 h of stereo (one stream)
L3 = 3600 * 48000
p = 2 ** (3 / 12)
pl3 = ctx.stretch_plan(1.0, p, L3)
d_x, d_y = ctx.empty(L3 * 2), ctx.empty(pl3.out_len * 2)
ctx.fill_uniform(d_x.ptr, L3 * 2, 0, 1, 7, 0)
src, dst = nae.Sig.interleaved(d_x.ptr, L3, 2), nae.Sig.interleaved(d_y.ptr, pl3.out_len, 2)
ms = timed(lambda: ctx.stretch_block(1.0, p, src, L3, 2, 1, dst), reps=3, warm=1)
# parity on the first 20 s against the oracle (the vocoder is causal up to its 1024-sample window)
Lp = 20 * 48000
xin = np.empty(Lp * 2, np.float32)
ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, xin.ctypes.data, d_x.ptr, xin.nbytes)); ctx.sync()
t0 = time.perf_counter()
ref = orc.stretch(xin, 2, 1.0, p)
cpu_s = time.perf_counter() - t0
gy = np.empty(Lp * 2, np.float32)
ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, gy.ctypes.data, d_y.ptr, gy.nbytes)); ctx.sync()
keep = (Lp - 4096) * 2
err = rr(gy[:keep], ref[:keep])
add("C3", "pitch +3 st, 1 stream x 1 h stereo (phase vocoder + transposer)", L3, ms, 16, f"rel-RMS {err:.1e} on first 20 s",
    f"oracle 1 thread: {Lp / cpu_s:.3e} sf/s (20 s sample)")
# the same hour through the SoundTouch-shaped chain (K7 option A): one stream = one workgroup walking ~46 000 WSOLA
# sequences in order (the sequential dependency of the algorithm), so this is the latency-bound corner of that path
wpl = ctx.wsola_plan(48000, 2, 1.0, p, L3)
ms_w = timed(lambda: ctx.wsola_block(48000, 1.0, p, src, L3, 2, 1, dst), reps=1, warm=1)
Lw = 5 * 48000
t0 = time.perf_counter()
ref_w = orc.st_process(xin[: Lw * 2], 2, 48000, 1.0, p)
cpu_w = time.perf_counter() - t0
gw = np.empty((Lw - 48000) * 2, np.float32)
ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, gw.ctypes.data, d_y.ptr, gw.nbytes)); ctx.sync()
okw = np.array_equal(gw, ref_w[: gw.size])       # a prefix: output frame n depends on input up to ~n + one sequence only
add("C3 (SoundTouch-shaped)", f"pitch +3 st, 1 stream x 1 h stereo, WSOLA + FIR + cubic, {wpl.n_seq} sequences in series", L3, ms_w, 16,
    "bit-exact on first 4 s" if okw else "MISMATCH", f"oracle 1 thread: {Lw / cpu_w:.3e} sf/s (5 s sample)")
d_x.free(); d_y.free()

# ---------------------------------------------------------------- C4: 4-node graph on 8 ch x 96 kHz = 4 stereo pairs, 60 s
ns, S4 = 4, 60 * 96000
pl4 = ctx.stretch_plan(1.0, p, S4)
F4 = ctx.spectrum_frames(pl4.out_len)
d_a, d_b = ctx.empty(ns * S4 * 2), ctx.empty(S4 * 2)
ctx.fill_uniform(d_a.ptr, S4 * 2, S4 * 2, ns, 0, 0)
ctx.fill_uniform(d_b.ptr, S4 * 2, 0, 1, 0, 1)
d_mix, d_pitch, d_spec = ctx.empty(ns * S4 * 2), ctx.empty(ns * pl4.out_len * 2), ctx.empty(ns * F4 * 2 * 513)
g = nae.Graph4()
g.in_a = nae.Sig.interleaved(d_a.ptr, S4, 2)
g.in_b = nae.Sig.interleaved(d_b.ptr, S4, 2, shared=True)
g.vol_a = g.vol_b = 0.5
g.mix_out = nae.Sig.planar(d_mix.ptr, S4, 2)
g.rate, g.pitch = 1.0, p
g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl4.out_len, 2)
g.spec_out, g.spec_stream_stride = d_spec.ptr, F4 * 2 * 513
g.S, g.n_streams = S4, ns
ms = timed(lambda: ctx.graph4(g), reps=5)
add("C4", "4-node graph, 4 stereo pairs (8 ch) x 96 kHz x 60 s (outside the reference's 2-ch / 48 kHz envelope)", ns * S4, ms, 64.03, "as C5")
for t in (d_a, d_b, d_mix, d_pitch, d_spec):
    t.free()

print("| config | what | sample-frames | ms | sample-frames/s | alg B/sf | alg GB/s | frac of 8 TB/s | parity | CPU oracle |")
print("|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
