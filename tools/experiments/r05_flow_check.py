#!/usr/bin/env python3
"""r05: the C5 graph at a given stream count under two NAE_PV_FLOW settings — every pitch sample and every spectrum value bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import naeload
nae = naeload.load()
n, S, pitch = int(sys.argv[1]), int(float(sys.argv[2]) * 48000), 2 ** (3 / 12)
res = []
for flow in sys.argv[3:]:
    os.environ["NAE_PV_FLOW"] = flow
    ctx = nae.Context(0)
    pl = ctx.stretch_plan(1.0, pitch, S)
    F = ctx.spectrum_frames(pl.out_len)
    d_a, d_b = ctx.empty(n * S * 2), ctx.empty(S * 2)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n, 0, 0)
    ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
    d_mix, d_pitch, d_spec = ctx.empty(n * S * 2), ctx.empty(n * pl.out_len * 2), ctx.empty(n * F * 2 * 513)
    d_pitch.zero(); d_spec.zero()
    g = nae.Graph4()
    g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
    g.in_b = nae.Sig.interleaved(d_b.ptr, S, 2, shared=True)
    g.vol_a = g.vol_b = 0.5
    g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
    g.rate, g.pitch = 1.0, pitch
    g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
    g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * 513
    g.S, g.n_streams = S, n
    ctx.prof_reset(); ctx.prof_enable(True)
    ctx.graph4(g); ctx.sync()
    ctx.prof_enable(False)
    print("NAE_PV_FLOW=" + flow, {k: round(v[0] / max(v[1], 1), 3) for k, v in ctx.prof_report().items()}, flush=True)
    res.append((d_pitch.download().view(np.uint32), d_spec.download().view(np.uint32)))
    ctx.close()
for i in range(1, len(res)):
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[i]))
    print("setting", sys.argv[3 + i], "vs", sys.argv[3], ":", "same bits" if same else "DIFFERENT")
    if not same:
        d = np.nonzero(res[0][0] != res[i][0])[0]
        print("pitch mismatches:", d.size, d[:10])
        sys.exit(1)
