#!/usr/bin/env python3
"""Golden vectors of the N2 input resampler (libswresample's default polyphase resampler as specified in
include/nae_dsp_spec.h and oracle/orc_swr.c), authored independently of the oracle: float64 throughout (the filter bank
is NOT rounded to float32, the taps are summed with numpy's pairwise float64 sum), positions from exact rational
arithmetic with Python integers.  Oracle and GPU must reproduce these within 1e-4 relative RMS.

    python tests/golden/swr_numpy.py        # rewrites tests/golden/swr_golden.npz

PARITY UNPINNED versus FFmpeg: neither libswresample nor a fixture of it exists under /root/reference."""
import os
from fractions import Fraction
from math import ceil

import numpy as np

from math import gcd

P_MAX, FILTER_SIZE, BETA, CUTOFF = 1024, 32, 9.0, 0.97


def phase_count(in_rate, out_rate):
    """exact_rational (libswresample's default): the reduced numerator of out/in when it fits the table, else 1024"""
    exact = out_rate // gcd(out_rate, in_rate)
    return exact if exact <= P_MAX else P_MAX


def bank64(in_rate, out_rate):
    P = phase_count(in_rate, out_rate)
    factor = min(out_rate * CUTOFF / in_rate, 1.0)
    L = max(int(ceil(FILTER_SIZE / factor)), 1)
    if L > 1:
        L = (L + 1) & ~1                      # resample_init: FFALIGN(filter_length, 2) for filters longer than one tap
    center = (L - 1) // 2
    i = np.arange(L)[None, :]
    ph = np.arange(P)[:, None]
    x = np.pi * ((i - center) - ph / P) * factor
    y = np.sinc(x / np.pi)
    w = 2.0 * x / (factor * L * np.pi)
    y = y * np.i0(BETA * np.sqrt(np.maximum(1.0 - w * w, 0.0)))
    return y / y.sum(axis=1, keepdims=True), L, center


def resample64(x, in_rate, out_rate):
    x = np.asarray(x, np.float64)
    N = x.size
    bank, L, center = bank64(in_rate, out_rate)
    P = phase_count(in_rate, out_rate)
    step = Fraction(in_rate * P, out_rate)                  # input position advance per output, in 1/P samples
    R = (min(N, L) + 1) // 2
    assert N > L
    ext = np.concatenate([x[1:L + 1][::-1], x, x[::-1][:R]])          # x[-k] = x[k];  x[N + j] = x[N - 1 - j]
    lead = L
    out = []
    n = 0
    while True:
        pos = -P * center + (n * step.numerator) // step.denominator
        s, ph = pos // P, pos % P
        if s + L > N + R:
            break
        out.append(float(np.sum(ext[lead + s: lead + s + L] * bank[ph])))
        n += 1
    return np.array(out)


def signals():
    rng = np.random.default_rng(20261004)
    t = np.arange(6000)
    return {"tones": 0.5 * np.sin(2 * np.pi * 1000 * t / 44100) + 0.25 * np.sin(2 * np.pi * 3300 * t / 44100),
            "noise": rng.uniform(-1, 1, 6000), "impulse": (t == 1000).astype(np.float64)}


def main():
    out = {}
    for name, x in signals().items():
        x32 = x.astype(np.float32)
        out[f"{name}_in"] = x32
        for in_rate, out_rate in ((44100, 48000), (22050, 48000), (96000, 48000), (8000, 48000), (88200, 48000)):   # 88.2 kHz: 61 -> 62 taps
            out[f"{name}_{in_rate}_{out_rate}"] = resample64(x32, in_rate, out_rate)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "swr_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
