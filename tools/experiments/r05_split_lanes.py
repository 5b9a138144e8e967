#!/usr/bin/env python3
"""r05 experiment: the C5 step with the batch cut into P parts issued on L contexts (= L HIP streams) of one device, so that the
memory-bound kernels of one part can run beside the ALU-bound vocoder of another.  Same kernels, same results; only the issue order
and the queues differ.  Prints ms per step for every (P, L, order) asked for, interleaved over --rounds rounds.

  python tools/experiments/r05_split_lanes.py --shapes 1x1,2x2,4x2,4x4,8x2 --steps 10 --rounds 3
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import naeload  # noqa: E402

BINS = 513


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--shapes", default="1x1,2x2,4x2,4x4")
    ap.add_argument("--orders", default="part")          # part: whole graph per part; stage: all mixes, all vocoders, all spectra
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    nae = naeload.load()
    S = int(round(a.seconds * 48000))
    n = a.streams
    pitch = 2.0 ** (3 / 12.0)
    ctxs = [nae.Context(0) for _ in range(8)]
    c0 = ctxs[0]
    pl = c0.stretch_plan(1.0, pitch, S)
    F = c0.spectrum_frames(pl.out_len)
    d_a = c0.empty(n * S * 2)
    d_b = c0.empty(S * 2)
    d_mix = c0.empty(n * S * 2)
    d_pitch = c0.empty(n * pl.out_len * 2)
    d_spec = c0.empty(n * F * 2 * BINS)
    c0.fill_uniform(d_a.ptr, S * 2, S * 2, n, 0, 0)
    c0.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    c0.sync()

    def graph(first, count):
        g = nae.Graph4()
        g.in_a = nae.Sig.interleaved(d_a.ptr + first * S * 2 * 4, S, 2)
        g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
        g.vol_a = g.vol_b = 0.5
        g.mix_out = nae.Sig.planar(d_mix.ptr + first * S * 2 * 4, S, 2)
        g.rate, g.pitch = 1.0, pitch
        g.pitch_out = nae.Sig.interleaved(d_pitch.ptr + first * pl.out_len * 2 * 4, pl.out_len, 2)
        g.spec_out, g.spec_stream_stride = d_spec.ptr + first * F * 2 * BINS * 4, F * 2 * BINS
        g.S, g.n_streams = S, count
        return g

    def peek(ptr, count):
        import numpy as np
        out = np.empty(count, np.uint32)
        c0._ck(c0.lib.nae_memcpy_d2h(c0.h, out.ctypes.data, ptr, count * 4))
        c0.sync()
        return out

    def snapshot():
        return (peek(d_pitch.ptr, 1 << 22), peek(d_pitch.ptr + (n * pl.out_len * 2 - (1 << 22)) * 4, 1 << 22),
                peek(d_spec.ptr + ((n - 1) * F * 2 * BINS) * 4, 1 << 20), peek(d_spec.ptr + ((n // 2) * F * 2 * BINS) * 4, 1 << 20))

    def make(P, L):
        per = n // P
        return [(ctxs[i % L], graph(i * per, per if i < P - 1 else n - per * (P - 1))) for i in range(P)]

    def run_steps(parts, L, order, steps):
        for _ in range(steps):
            if order == "part":
                for c, g in parts:
                    c.graph4(g)
            else:
                for m in (1, 2, 4):
                    for c, g in parts:
                        c.graph4_stages(g, m)
        for c in ctxs[:L]:
            c.sync()

    shapes = [tuple(int(v) for v in s.split("x")) for s in a.shapes.split(",")]
    orders = a.orders.split(",")
    ref = None
    if a.check:
        import numpy as np
        run_steps(make(1, 1), 1, "part", 1)
        ref = snapshot()
    for r in range(a.rounds):
        for (P, L) in shapes:
            for order in orders:
                parts = make(P, L)
                if ref is not None:
                    for arr in (d_pitch, d_spec):
                        arr.zero()
                    c0.sync()
                run_steps(parts, L, order, a.warmup)
                t0 = time.perf_counter()
                run_steps(parts, L, order, a.steps)
                ms = (time.perf_counter() - t0) / a.steps * 1e3
                ok = ""
                if ref is not None:
                    import numpy as np
                    ok = " same bits" if all(np.array_equal(x, y) for x, y in zip(ref, snapshot())) else " DIFFERENT"
                print(f"round {r} parts {P} lanes {L} order {order:5s}: {ms:7.3f} ms per step  clock {c0.clock_ghz():.2f} GHz{ok}", flush=True)


if __name__ == "__main__":
    main()
