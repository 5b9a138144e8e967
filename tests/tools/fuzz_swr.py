#!/usr/bin/env python3
"""Randomised differential run (GPU box): the N2 input converter (nae_swr: swr-default polyphase resampler, format conversion,
mono -> stereo) against oracle/orc_swr.c, bit for bit, over random input rates, formats, lengths and cuts into convert calls —
through nae_swr_convert_host and through the queued nae_swr_convert.
    python tests/tools/fuzz_swr.py [cases=24] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import naeload
import orc
from test_gpu_nodes import swr_drive


def main(cases=24, seed=1, ctx=None, nae=None):
    rng = np.random.default_rng(seed)
    if nae is None:
        nae = naeload.load()
    if ctx is None:
        ctx = nae.Context(0)
    done = 0
    for k in range(cases):
        in_rate = int(rng.choice([8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, int(rng.integers(7000, 200000))]))
        n = int(rng.integers(200, 30000))
        chunks = [int(c) for c in rng.integers(1, 4000, int(rng.integers(1, 6)))]
        max_out = int(rng.choice([1152, 4096, 20000]))
        x = (0.8 * rng.uniform(-1, 1, n)).astype(np.float32)
        st = np.stack([x, -x], 1).reshape(-1).astype(np.float32)
        queued = bool(rng.integers(2))
        try:
            L, R = swr_drive(ctx, nae, nae.FMT_FLT, in_rate, 2, lambda p, m: [np.ascontiguousarray(st[2 * p: 2 * (p + m)])], n, chunks,
                             max_out=max_out, queued=queued)
        except AssertionError:
            print(f"case {k:3d}: in_rate {in_rate}: refused by nae_swr_create")
            continue
        # equal rates: libswresample sets up no resampler at all (a wire); the oracle's resampler is only defined for a real ratio
        ident = in_rate == 48000
        ref = x if ident else orc.swr_resample(x, in_rate, 48000)
        assert L.size == ref.size, (k, in_rate, n, L.size, ref.size, chunks, max_out, queued)
        assert np.array_equal(L.view(np.uint32), ref.view(np.uint32)), f"case {k}: L differs"
        assert np.array_equal(R.view(np.uint32), (-x if ident else orc.swr_resample(-x, in_rate, 48000)).view(np.uint32)), f"case {k}: R differs"
        done += 1
        print(f"case {k:3d}: in_rate {in_rate:6d} frames {n:6d} -> {L.size:6d} chunks {chunks[:3]} max_out {max_out} {'queued' if queued else 'host'}  bit-exact", flush=True)
    print(f"{done} cases bit-exact")
    return done


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
