#!/usr/bin/env python3
"""A/B helper: the C5 graph on a few streams under each environment setting given on the command line; every result buffer is
compared bit for bit with the first setting's.     python tools/ab_check.py "" "NAE_DEBUG=spec_narrow=1" "NAE_DEBUG=pv_flow=2" """
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload


def run(nae, env, n, S, semis):
    for kv in env.split():
        k, v = kv.split("=")
        os.environ[k] = v
    try:
        p = 2 ** (semis / 12)
        with nae.Context(0) as ctx:
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
            ctx.graph4(g)
            return [d_mix.download().view(np.uint32), d_pitch.download().view(np.uint32), d_spec.download().view(np.uint32)]
    finally:
        for kv in env.split():
            os.environ.pop(kv.split("=")[0], None)


def main():
    nae = naeload.load()
    envs = sys.argv[1:] or [""]
    bad = 0
    for n, S, semis in ((6, 48000, 3.0), (5, 30011, 7.0), (9, 12345, 1.0), (1300, 6000, 3.0)):
        base = run(nae, envs[0], n, S, semis)
        for e in envs[1:]:
            got = run(nae, e, n, S, semis)
            same = [bool(np.array_equal(a, b)) for a, b in zip(base, got)]
            print(f"n={n} S={S} st={semis:+g} [{e}] mix/pitch/spectrum identical: {same}", flush=True)
            bad += same.count(False)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
