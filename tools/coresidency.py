#!/usr/bin/env python3
"""Experiment (round 4): the C5 step as TWO half batches on two contexts (= two HIP streams) with device-side dependencies that
decide which kernels may share the CUs — the vector/LDS-bound vocoder of one half beside the memory-bound mix + transposer and
the spectrum of the other half.  Stop rule of the round: keep a schedule only if the step drops by >= 0.3 ms.

    python tools/coresidency.py [--streams 1024] [--steps 30] [--modes one free stagger pipelined split]

modes
  one        one context, whole batch (the product's default path)
  free       two contexts, both halves enqueued back to back, no dependency (the hardware interleaves as it likes)
  stagger    front(B) waits for front(A): A's vocoder runs beside B's mix + transposer, B's vocoder beside A's spectrum
  pipelined  the two vocoders never overlap each other: vocoder(B, n) waits for vocoder(A, n), vocoder(A, n+1) for vocoder(B, n);
             each vocoder runs beside the other half's spectrum and next front stage
  joined     stagger, and stream A waits for B's last kernel at the end of every step (what a split inside nae_graph4_run would
             have to do so that whatever the caller enqueues next sees all results)
Each two-context mode is run with the vocoder in its 64-VGPR shape (NAE_DEBUG=pv_lean=1, leaves registers and LDS to a co-runner) and in
the shape nae_graph4_run picks by itself for 512 streams (128 VGPRs, tables in registers, fills the register file alone)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload


def build(nae, n, S, p, first_stream=0):
    ctx = nae.Context(0)
    pl = ctx.stretch_plan(1.0, p, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_a, d_b = ctx.empty(n * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n, first_stream, 0)
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


def run(mode, lanes, steps):
    (A, gA, _), (B, gB, _) = lanes
    evAF, evVA, evVB = A.event(), A.event(), B.event()
    for n in range(steps):
        if mode == "free":
            A.graph4_stages(gA, 7)
            B.graph4_stages(gB, 7)
        elif mode in ("stagger", "joined"):
            A.graph4_stages(gA, 1)
            A.record(evAF)
            A.graph4_stages(gA, 2)
            B.wait_event(evAF)
            B.graph4_stages(gB, 1)
            A.graph4_stages(gA, 4)
            B.graph4_stages(gB, 6)
            if mode == "joined":
                B.record(evVB)
                A.wait_event(evVB)
        elif mode == "pipelined":
            A.graph4_stages(gA, 1)
            if n > 0:
                A.wait_event(evVB)
            A.graph4_stages(gA, 2)
            A.record(evVA)
            A.graph4_stages(gA, 4)
            B.graph4_stages(gB, 1)
            B.wait_event(evVA)
            B.graph4_stages(gB, 2)
            B.record(evVB)
            B.graph4_stages(gB, 4)
        else:
            raise SystemExit(mode)
    A.sync()
    B.sync()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--modes", nargs="+", default=["one", "free", "stagger", "joined", "pipelined"])
    a = ap.parse_args()
    nae = naeload.load()
    S, p = 480000, 2 ** (3 / 12)
    half = a.streams // 2

    def timed(fn):
        fn(8)                                    # settle the clock
        t0 = time.perf_counter()
        fn(a.steps)
        return (time.perf_counter() - t0) / a.steps * 1e3

    for mode in a.modes:
        if mode == "one":
            os.environ.pop("NAE_DEBUG", None)
            ctx, g, bufs = build(nae, a.streams, S, p)

            def fn(k):
                for _ in range(k):
                    ctx.graph4(g)
                ctx.sync()
            ms = timed(fn)
            print(f"{mode:10s} {'':22s} {ms:7.3f} ms per step of {a.streams} streams, clock {ctx.clock_ghz():.2f} GHz", flush=True)
            ctx.close()
            continue
        for lean in (True, False):
            if lean:
                os.environ["NAE_DEBUG"] = "pv_lean=1"
            else:
                os.environ.pop("NAE_DEBUG", None)
            lanes = [build(nae, half, S, p, 0), build(nae, a.streams - half, S, p, half)]
            ms = timed(lambda k: run(mode, lanes, k))
            print(f"{mode:10s} {'vocoder 64 VGPRs' if lean else 'vocoder 128 VGPRs':22s} {ms:7.3f} ms per step of {a.streams} streams, "
                  f"clock {lanes[0][0].clock_ghz():.2f} GHz", flush=True)
            for ctx, _, _ in lanes:
                ctx.close()
        os.environ.pop("NAE_DEBUG", None)


if __name__ == "__main__":
    main()
