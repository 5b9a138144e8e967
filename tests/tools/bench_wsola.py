#!/usr/bin/env python3
"""Time the SoundTouch-shaped WSOLA chain (K7 option A) at the C5 per-GPU size: N streams x 10 s stereo.
Lives under tests/ because --check compares with, and times, the CPU oracle (only tests/, smoke() and bench.py may).
   python tests/tools/bench_wsola.py [--streams 1024] [--seconds 10] [--rate 1.0] [--semitones 3] [--check]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import naeload

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--rate", type=float, default=1.0)
ap.add_argument("--semitones", type=float, default=3.0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--check", action="store_true", help="compare stream 0 and the last stream with the CPU oracle")
args = ap.parse_args()

nae = naeload.load()
ctx = nae.Context(0)
sr, ch = 48000, 2
S = int(args.seconds * sr)
pitch = 2.0 ** (args.semitones / 12.0)
pl = ctx.wsola_plan(sr, ch, args.rate, pitch, S)
n = args.streams
d_x = ctx.empty(n * S * ch)
d_y = ctx.empty(n * pl.out_len * ch)
ctx.fill_uniform(d_x.ptr, S * ch, S * ch, n)
src, dst = nae.Sig.interleaved(d_x.ptr, S, ch), nae.Sig.interleaved(d_y.ptr, pl.out_len, ch)
t0 = time.perf_counter()
ctx.wsola_block(sr, args.rate, pitch, src, S, ch, n, dst)      # first call: plan + cubic table + workspaces
ctx.sync()
first_ms = (time.perf_counter() - t0) * 1e3
ctx.prof_enable(True)
ctx.prof_reset()
a, b = ctx.event(), ctx.event()
ctx.record(a)
for _ in range(args.steps):
    ctx.wsola_block(sr, args.rate, pitch, src, S, ch, n, dst)
ctx.record(b)
ms = ctx.elapsed_ms(a, b) / args.steps
rep = {k: v[0] / v[1] for k, v in ctx.prof_report().items() if v[1]}
res = {"streams": n, "frames_per_stream": S, "rate": args.rate, "pitch": pitch, "order": pl.order, "n_seq": pl.n_seq,
       "overlap": pl.overlap_len, "sequence": pl.seq_len, "seek": pl.seek_len, "ms_per_step": ms,
       "sample_frames_per_s": n * S / (ms * 1e-3), "first_call_ms": first_ms, "kernels_ms": rep}
if args.check:
    import orc
    y = d_y.download().reshape(n, -1)
    for s in (0, n - 1):
        x = orc.fill_uniform(S * ch, orc.stream_seed(s))
        t0 = time.perf_counter()
        ref = orc.st_process(x, ch, sr, args.rate, pitch)
        cpu_s = time.perf_counter() - t0
        res[f"bit_exact_stream_{s}"] = bool(np.array_equal(y[s].view(np.uint32), ref.view(np.uint32)))
    res["cpu_oracle_sample_frames_per_s_1thread"] = S / cpu_s
print(json.dumps(res))
