"""Randomised differential run (GPU box) of the vocoder on TINY inputs (1 .. 5000 sample-frames: fewer frames than a pipeline step,
lengths around the FFT size) for every shape of the pipeline (pv_fps 1 / 2 / 4 through nae_debug_set, with and without 64-frame tiles) against the oracle.
Outputs of a few samples sit on the window's edge, where they are rounding noise (1e-11 for a 0.5-amplitude input), so the error is
measured against max(RMS of the reference, 1e-4): python tests/tools/fuzz_tiny.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, naeload, orc
nae = naeload.load()
rng = np.random.default_rng(5)
worst = 0.0
n_cases = 0
for fps in (0, 1, 2, 4):
    for tile in (0, 64):
        with nae.Context(0) as ctx:
            ctx.debug_set("pv_fps", fps).debug_set("pv_tile", tile)
            for k in range(14):
                ch = int(rng.choice([1, 2])); n_streams = int(rng.choice([1, 2, 3, 9]))
                L = int(rng.choice([1, 2, 17, 200, 255, 256, 257, 511, 700, 1023, 1024, 1025, 1500, 2600, 5000]))
                rate = float(np.exp(rng.uniform(np.log(0.5), np.log(2.0)))); pitch = float(2.0 ** rng.uniform(-1.0, 1.0))
                if not (0.25 <= rate * pitch <= 4.0): continue
                x = (0.5 * rng.uniform(-1, 1, (n_streams, L, ch))).astype(np.float32)
                pl = ctx.stretch_plan(rate, pitch, L)
                d_x, d_o = ctx.array(x.reshape(-1)), ctx.empty(max(1, n_streams * pl.out_len * ch))
                ctx.stretch_block(rate, pitch, nae.Sig.interleaved(d_x.ptr, L, ch), L, ch, n_streams, nae.Sig.interleaved(d_o.ptr, pl.out_len, ch))
                out = d_o.download()[: n_streams * pl.out_len * ch].reshape(n_streams, pl.out_len, ch)
                d_x.free(); d_o.free()
                for s in range(n_streams):
                    ref = orc.stretch(x[s].reshape(-1), ch, rate, pitch).reshape(-1, ch)
                    assert ref.shape == out[s].shape, (fps, tile, L, ref.shape, out[s].shape)
                    if ref.size:
                        den = max(np.sqrt(np.mean(ref.astype(np.float64) ** 2)), 1e-4)
                        e = np.sqrt(np.mean((out[s].astype(np.float64) - ref) ** 2)) / den
                        if e > 2e-6: print(f"  fps {fps} tile {tile} L {L} ch {ch} streams {n_streams} rate {rate:.3f} pitch {pitch:.3f}: err {e:.2e} (rms ref {den:.2e})")
                        worst = max(worst, e)
                        assert e <= 1e-4, (fps, tile, L, ch, n_streams, rate, pitch, e)
                n_cases += 1
print("cases", n_cases, "worst", worst)
