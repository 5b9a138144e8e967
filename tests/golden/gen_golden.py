#!/usr/bin/env python3
"""Golden-vector generator.  Run:  python tests/golden/gen_golden.py   (writes tests/golden/*.npz)

The reference (Stehsaer/nodey-audio-editor) ships no tests, fixtures or golden files and cannot be
built or imported here (C++23 + FFmpeg/SoundTouch/Boost), so these vectors come from an
INDEPENDENT restatement of the reference's arithmetic in numpy float32 — one IEEE-rounded numpy op per
reference op, no fused multiply-add anywhere (numpy has none) — written from the reference source text:

  K1 gain        src/processor/audio-vol.cpp:75-100       K4 bimix v1  src/processor/audio-bimix.cpp:310-317
  K2 split/merge src/processor/audio-velocity.cpp:169-180 K5 bimix v2  src/processor/audio-bimix.cpp:624-627,833-850
  K3 amix        src/processor/audio-amix.cpp:293-307     K6 to-f32    src/processor/audio-velocity.cpp:160-229

K8 (FFT spectrum) has no reference code; its golden is scipy.fft.rfft in float64 of the Hann-windowed frames
(FFTW r2c convention: un-normalised forward DFT), compared under the 1e-4 relative-RMS tolerance.
K7 (tempo/pitch) has no independent implementation to pin against (SoundTouch absent: PARITY UNPINNED); the
file k7_regression.npz stores the ORACLE's own output for a tone so later rounds notice unintended drift.
k7_golden.npz pins the same node to its SPECIFICATION: authored by pv_numpy.py, a float64 restatement
with numpy's FFT, compared under the 1e-4 relative-RMS tolerance.
The SoundTouch-shaped chain (K7 option A) is authored by st_numpy.py, a separate numpy-float32 block restatement.
"""
import os
import sys

import numpy as np
import scipy.fft

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def splitmix_uniform(n, seed):
    """SURVEY.md §8d generator: splitmix64 -> high u32 -> float(u>>8) * 2^-23 - 1"""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(32)).astype(np.uint32)
    return ((u >> np.uint32(8)).astype(f32) * f32(1.0 / 8388608.0) - f32(1.0)).astype(f32)


def x86_trunc_i32(f):
    """cvttss2si: truncate; 0x80000000 when out of int32 range or NaN"""
    f = f.astype(f32)
    ok = (f >= f32(-2147483648.0)) & (f < f32(2147483648.0))
    t = np.where(ok, np.trunc(np.where(ok, f, 0)).astype(np.float64), -2147483648.0)
    return t.astype(np.int64).astype(np.int32)


def gain(x, vol):
    vol = f32(vol)
    if x.dtype == np.float32:
        return (x * vol).astype(f32)
    r = x86_trunc_i32(x.astype(f32) * vol)
    if x.dtype == np.int16:
        return (r.astype(np.int64) & 0xFFFF).astype(np.uint16).view(np.int16)
    return r


def amix(inL, inR, vol):
    l = np.zeros_like(inL[0], dtype=f32)
    r = np.zeros_like(inR[0], dtype=f32)
    for a, b, v in zip(inL, inR, vol):
        l = (l + (a * f32(v)).astype(f32)).astype(f32)
        r = (r + (b * f32(v)).astype(f32)).astype(f32)
    return l, r


def bimix(ll, lr, rl, rr, bias):
    bias = f32(bias)
    bm, bp = f32(f32(1) - bias), f32(f32(1) + bias)
    two = f32(2)
    outL = (((ll / two).astype(f32) + (lr / two).astype(f32)).astype(f32) * bm).astype(f32)
    outR = (((rl / two).astype(f32) + (rr / two).astype(f32)).astype(f32) * bp).astype(f32)
    return outL, outR


def to_f32(fmt, planes, S, ch):
    if fmt == "flt":
        return planes[0].astype(f32)
    if fmt == "fltp":
        return np.stack(planes, 1).reshape(-1).astype(f32)
    if fmt == "s16":
        return (planes[0].astype(f32) / f32(32768.0)).astype(f32)
    if fmt == "s16p":
        return np.stack([(p.astype(f32) / f32(32767)).astype(f32) for p in planes], 1).reshape(-1)
    if fmt == "s32":
        return (planes[0].astype(f32) / f32(2147483648.0)).astype(f32)
    if fmt == "s32p":
        return np.stack([(p.astype(np.float64) / 2147483647.0).astype(f32) for p in planes], 1).reshape(-1)
    raise ValueError(fmt)


def main():
    rng = np.random.default_rng(20261004)
    out = {}

    # ---- synthetic generator KAT
    out["splitmix_seed"] = np.array([0x9E3779B97F4A7C15], np.uint64)
    out["splitmix_out"] = splitmix_uniform(64, 0x9E3779B97F4A7C15)

    # ---- K1
    S = 4099  # odd: exercises the scalar tail
    xf = splitmix_uniform(S * 2, 1)
    xf[:8] = [0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40, 3.4e38, np.inf]  # zeros, denormals, overflow
    for vol in (0.0, 0.70710678, 1.0, 10.0):
        out[f"k1_f32_v{vol}"] = gain(xf, vol)
    out["k1_f32_in"] = xf
    xi16 = rng.integers(-32768, 32768, S * 2, dtype=np.int16)
    xi16[:4] = [32767, -32768, 0, -1]
    out["k1_s16_in"] = xi16
    for vol in (0.5, 0.70710678, 1.0, 3.0, 10.0):  # 3.0 and 10.0 leave the int16 range: modular wrap
        out[f"k1_s16_v{vol}"] = gain(xi16, vol)
    xi32 = rng.integers(-2**31, 2**31, S * 2, dtype=np.int64).astype(np.int32)
    xi32[:4] = [2**31 - 1, -2**31, 0, -1]
    out["k1_s32_in"] = xi32
    for vol in (0.5, 0.70710678, 1.0, 3.0):  # 3.0 overflows int32: x86 "integer indefinite"
        out[f"k1_s32_v{vol}"] = gain(xi32, vol)

    # ---- K3 (n = 1, 2, 16)
    S3 = 1153
    ins = [splitmix_uniform(S3 * 2, 100 + i) for i in range(16)]
    out["k3_in"] = np.stack(ins)
    for n in (1, 2, 16):
        vol = (np.arange(1, n + 1, dtype=f32) / f32(n * (n + 1) / 2)).astype(f32)
        L, R = amix([a[0::2] for a in ins[:n]], [a[1::2] for a in ins[:n]], vol)
        out[f"k3_vol_n{n}"] = vol
        out[f"k3_L_n{n}"], out[f"k3_R_n{n}"] = L, R

    # ---- K4
    q = [splitmix_uniform(S3, 200 + i) for i in range(4)]
    out["k4_in"] = np.stack(q)
    for bias in (-1.0, -0.3, 0.0, 0.25, 1.0):
        L, R = bimix(q[0], q[1], q[2], q[3], bias)
        out[f"k4_L_b{bias}"], out[f"k4_R_b{bias}"] = L, R

    # ---- K5
    out["k5_mono"] = ((q[0] + q[1]).astype(f32).astype(np.float64) * 0.5).astype(f32)
    un, al = 37, 500
    inter = np.zeros(2 * (un + al), f32)
    inter[0:2 * un:2] = q[2][:un]
    inter[2 * un::2] = q[2][un:un + al]
    inter[2 * un + 1::2] = q[3][:al]
    out["k5_inter_e0"] = inter
    inter1 = np.zeros(2 * (un + al), f32)
    inter1[1:2 * un:2] = q[2][:un]
    inter1[2 * un + 1::2] = q[2][un:un + al]
    inter1[2 * un::2] = q[3][:al]
    out["k5_inter_e1"] = inter1

    # ---- K6
    S6 = 777
    a16 = rng.integers(-32768, 32768, (2, S6), dtype=np.int16)
    a16[:, :2] = [[32767, -32768], [-32768, 32767]]
    a32 = rng.integers(-2**31, 2**31, (2, S6), dtype=np.int64).astype(np.int32)
    a32[:, :2] = [[2**31 - 1, -2**31], [-2**31, 2**31 - 1]]
    af = splitmix_uniform(2 * S6, 300).reshape(2, S6)
    out["k6_s16_planes"], out["k6_s32_planes"], out["k6_f32_planes"] = a16, a32, af
    out["k6_s16"] = to_f32("s16", [a16.T.reshape(-1).copy()], S6, 2)      # packed input = interleaved planes
    out["k6_s16p"] = to_f32("s16p", [a16[0], a16[1]], S6, 2)
    out["k6_s32"] = to_f32("s32", [a32.T.reshape(-1).copy()], S6, 2)
    out["k6_s32p"] = to_f32("s32p", [a32[0], a32[1]], S6, 2)
    out["k6_fltp"] = to_f32("fltp", [af[0], af[1]], S6, 2)

    np.savez_compressed(os.path.join(HERE, "nodes.npz"), **out)

    # ---- K8: float64 DFT of Hann-windowed frames (noise + the KAT tone + an impulse)
    T = 1024 + 256 * 5 + 100
    n = np.arange(T)
    tone = (0.5 * np.sin(2 * np.pi * 1000 * n / 48000) + 0.25 * np.sin(2 * np.pi * 3300 * n / 48000)).astype(f32)
    noise = splitmix_uniform(T, 400)
    imp = np.zeros(T, f32)
    imp[1000] = 1.0
    hann = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)
    k8 = {}
    for name, sig in (("tone", tone), ("noise", noise), ("impulse", imp)):
        F = (T - 1024) // 256 + 1
        frames = np.stack([sig[f * 256:f * 256 + 1024].astype(np.float64) * hann for f in range(F)])
        k8[f"{name}_in"] = sig
        k8[f"{name}_mag"] = np.abs(scipy.fft.rfft(frames, axis=1))  # float64
    np.savez_compressed(os.path.join(HERE, "spectrum.npz"), **k8)

    # ---- K7 regression (ORACLE output, not an independent pin)
    sys.path.insert(0, os.path.dirname(HERE))
    import orc
    L = 8192
    n = np.arange(L)
    tone = (0.5 * np.sin(2 * np.pi * 1000 * n / 48000) + 0.25 * np.sin(2 * np.pi * 3300 * n / 48000)).astype(f32)
    k7 = {"in": tone}
    for name, (rate, pitch) in {"pitch_up3": (1.0, 2 ** (3 / 12)), "tempo_1p5": (1.5, 1 / 1.5), "rate_0p8": (0.8, 1.0)}.items():
        k7[name] = orc.stretch(tone, 1, rate, pitch)
        k7[name + "_params"] = np.array([rate, pitch])
    np.savez_compressed(os.path.join(HERE, "k7_regression.npz"), **k7)

    # ---- K7 vocoder + transposer against the SPECIFICATION: pv_numpy.py is a float64 restatement of DESIGN.md §3.3 with
    # numpy's FFT (nothing shared with the oracle's canonical FFT / polynomial atan2).  Faded two-tone and chirp inputs:
    # no phase-wrap decision of an energetic bin sits within float32 rounding of half a turn.
    import pv_numpy
    Lp = 12000
    n = np.arange(Lp)
    fade = np.ones(Lp)
    fade[:2048] = 0.5 - 0.5 * np.cos(np.pi * np.arange(2048) / 2048)
    fade[-2048:] = fade[:2048][::-1]
    tone2 = ((0.5 * np.sin(2 * np.pi * 1000 * n / 48000) + 0.25 * np.sin(2 * np.pi * 3300 * n / 48000)) * fade).astype(f32)
    chirp = (0.4 * np.sin(2 * np.pi * (300 * n / 48000 + 4000 * (n / 48000) ** 2)) * fade).astype(f32)
    kg = {"mono": tone2, "stereo": np.stack([tone2, chirp], 1).reshape(-1)}
    for name, (src, ch, rate, pitch) in {"up3": ("stereo", 2, 1.0, 2 ** (3 / 12)), "down4": ("stereo", 2, 1.0, 2 ** (-4 / 12)),
                                         "tempo_1p5": ("stereo", 2, 1.5, 1 / 1.5), "tempo_0p7": ("stereo", 2, 0.7, 1 / 0.7),
                                         "rate2_up5": ("mono", 1, 2.0, 2 ** (5 / 12)), "rate_0p8": ("mono", 1, 0.8, 1.0)}.items():
        kg[name] = pv_numpy.stretch(kg[src], ch, rate, pitch).astype(f32)
        kg[name + "_params"] = np.array([ch, rate, pitch])
        kg[name + "_src"] = np.array(src)
    np.savez_compressed(os.path.join(HERE, "k7_golden.npz"), **kg)

    # ---- SoundTouch-shaped chain (K7 option A): authored by the independent numpy block restatement st_numpy.py, so
    # the C oracle (streaming, FIFO by FIFO) is pinned by construction as K1-K6 are; versus SoundTouch itself the chain
    # stays PARITY UNPINNED (library absent).  A two-tone stereo signal and a mono noise signal, every stage order.
    import st_numpy
    Lw = 16000
    n = np.arange(Lw)
    st_in = np.stack([0.5 * np.sin(2 * np.pi * 440 * n / 48000) + 0.2 * np.sin(2 * np.pi * 1234.5 * n / 48000),
                      0.4 * np.sin(2 * np.pi * 660 * n / 48000 + 1.0)], 1).astype(f32).reshape(-1)
    ws = {"in": st_in, "mono_in": splitmix_uniform(12000, 77)}
    cases = {"pitch_up3": ("in", 2, 48000, 1.0, 2 ** (3 / 12)), "tempo_1p25": ("in", 2, 48000, 1.25, 0.8),
             "pitch_down4": ("in", 2, 48000, 1.0, 2 ** (-4 / 12)), "mono_22k_down": ("mono_in", 1, 22050, 1.0, 0.8),
             "mono_8k_rate": ("mono_in", 1, 8000, 1.3, 1.0)}
    for name, (src, ch, sr, rate, pitch) in cases.items():
        y, offs = st_numpy.process(ws[src], ch, sr, rate, pitch)
        ws[name] = y
        ws[name + "_offsets"] = offs          # the restatement runs through all 200 flush blocks: a superset of the oracle's
        ws[name + "_params"] = np.array([ch, sr, rate, pitch])
        ws[name + "_src"] = np.array(src)
    np.savez_compressed(os.path.join(HERE, "wsola_golden.npz"), **ws)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
