#!/usr/bin/env python3
"""Experiment: the C5 step as k independent slices of the batch on k contexts (= k HIP streams), launched together so that one
slice's vector-bound vocoder can overlap another slice's memory-bound mix / spectrum kernels.
    python tools/overlap_probe.py [--streams 1024] [--slices 1 2 4] [--steps 10] [--stagger 0|1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload


def build(nae, n, S, p):
    ctx = nae.Context(0)
    pl = ctx.stretch_plan(1.0, p, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_a, d_b = ctx.empty(n * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n, 0, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
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
    ctx.sync()
    return ctx, g, (d_a, d_b, d_mix, d_pitch, d_spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--slices", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--stagger-ms", type=float, nargs="+", default=[0.0])
    a = ap.parse_args()
    nae = naeload.load()
    S, p = 480000, 2 ** (3 / 12)
    for k in a.slices:
        lanes = [build(nae, a.streams // k, S, p) for _ in range(k)]
        for mode in ["one after the other"] + [f"together, slice i started {st} ms after slice i-1" for st in a.stagger_ms]:
            for rep in range(2):
                t0 = time.perf_counter()
                for it in range(a.steps):
                    for i, (ctx, g, _) in enumerate(lanes):
                        if it == 0 and i > 0 and mode.startswith("together"):
                            time.sleep(float(mode.split()[-5]) * 1e-3)
                        ctx.graph4(g)
                        if not mode.startswith("together"):
                            ctx.sync()
                for ctx, _, _ in lanes:
                    ctx.sync()
                dt = (time.perf_counter() - t0) / a.steps
            print(f"{k} slice(s) of {a.streams // k} streams, {mode}: {dt * 1e3:.2f} ms per step of {a.streams} streams", flush=True)
        for ctx, _, bufs in lanes:
            ctx.close()


if __name__ == "__main__":
    main()
