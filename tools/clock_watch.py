#!/usr/bin/env python3
"""Sample the GPU's core clock and power (rocm-smi) while the C5 graph runs in a loop: is the step clock- or power-limited?
   python tools/clock_watch.py [seconds [streams [stage mask: 7 = whole graph, 2 = the vocoder alone]]]
   (NAE_DEBUG=pv_flow=0 | pv_flow=2 in the environment: two-barrier / one-barrier vocoder pipeline; profiles/r05_flow.md)"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import naeload

nae = naeload.load()
ctx = nae.Context(0)
n, S, p = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 480000, 2 ** (3 / 12)
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 7
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
samples, stop = [], False


def watch():
    while not stop:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True)
        try:
            samples.append((time.perf_counter(), json.loads(r.stdout)))
        except Exception:
            samples.append((time.perf_counter(), r.stdout[-300:] + r.stderr[-300:]))


secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ctx.graph4(g); ctx.sync()
t = threading.Thread(target=watch); t.start()
t0 = time.perf_counter(); steps = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        if mask == 7:
            ctx.graph4(g)
        else:
            ctx.graph4_stages(g, mask)
    ctx.sync(); steps += 10
t1 = time.perf_counter(); stop = True; t.join()
print("ms per step over %.1f s: %.3f" % (t1 - t0, (t1 - t0) / steps * 1e3))
for ts, s in samples[:: max(1, len(samples) // 12)]:
    print("%.2f s" % (ts - t0), json.dumps(s)[:600])
# averages over the second half of the run (settled)
half = [s for ts, s in samples if ts - t0 > secs / 2 and isinstance(s, dict)]
def num(v):
    try:
        return float(str(v).strip("()MhzW ").split()[0].replace("Mhz", ""))
    except Exception:
        return None
acc = {}
for s in half:
    for card, d in s.items():
        if isinstance(d, dict):
            for k, v in d.items():
                x = num(v)
                if x is not None and ("sclk" in k.lower() or "power" in k.lower()):
                    acc.setdefault(k, []).append(x)
print("settled averages (%d samples):" % len(half), {k: round(sum(v) / len(v), 1) for k, v in acc.items()})
