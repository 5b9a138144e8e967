"""Independent float64 restatement of the builder-specified K7 node (phase vocoder N = 1024, synthesis
hop 256, Hann/Hann, gain 2/3, exact Q0.32 phase accumulation; 16-tap Kaiser-sinc rate transposer, 128 phases, linear
phase interpolation; transposer first when rho > 1), written from the specification text with numpy's own FFT — none of
the oracle's code paths (canonical radix-8 FFT, polynomial atan2) are shared.  gen_golden.py uses it to author
tests/golden/k7_golden.npz; the C oracle must reproduce it within the float tolerance of BASELINE.json (1e-4 relative
RMS) on tonal material, where no phase-wrap decision sits within float32 rounding of half a turn for a bin that carries
energy.  It pins the oracle to the specification, not to SoundTouch (PARITY UNPINNED there)."""
import numpy as np

N, HOP, BINS = 1024, 256, 513
TAPS, PHASES, BETA, CUTOFF = 16, 128, 8.0, 0.94


def plan(rate, pitch, L):
    tempo, rho = 1.0 / pitch, rate * pitch
    if abs(tempo - 1.0) < 1e-6:
        tempo = 1.0
    if abs(rho - 1.0) < 1e-6:
        rho = 1.0
    pv_on, rs_on = tempo != 1.0, rho != 1.0
    ha = int(np.floor(HOP * tempo * 2 ** 24 + 0.5))
    out_len = int(np.floor(L / (tempo * rho) + 0.5))
    rs_first = pv_on and rs_on and rho > 1.0
    step = int(np.floor(rho * 2 ** 32 + 0.5))
    if rs_first:
        mid = int(np.floor(L / rho + 0.5))
        pv_out = out_len
    elif rs_on:
        mid = (((out_len - 1) * step) >> 32) + TAPS // 2 + 1 if out_len else 0
        pv_out = mid
    else:
        mid = pv_out = out_len
    frames = (pv_out + N // 2 + HOP - 1) // HOP + 1 if pv_on else 0
    return dict(tempo=tempo, rho=rho, pv_on=pv_on, rs_on=rs_on, ha=ha, out_len=out_len, mid=mid, frames=frames,
                rs_first=rs_first, step=step, pv_out=pv_out)


def vocoder(x, pl, M):
    """one channel, float64 in/out"""
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
    k = np.arange(BINS)
    v = np.zeros(M + N + HOP)
    d0 = pl["ha"] >> 24
    qs = qa_prev = None
    s_prev = 0
    for f in range(pl["frames"]):
        s = (((f - 1) * pl["ha"] + (1 << 23)) >> 24) - N // 2
        idx = s + np.arange(N)
        ok = (idx >= 0) & (idx < x.size)
        fr = np.where(ok, x[np.clip(idx, 0, max(x.size - 1, 0))] if x.size else 0.0, 0.0)
        X = np.fft.rfft(fr * w)
        qa = np.round(np.angle(X) / (2 * np.pi) * 2 ** 32).astype(np.int64) & 0xFFFFFFFF
        if f == 0:
            qs = qa.copy()
        else:
            d = s - s_prev
            R = ((HOP << 24) + d // 2) // d
            e = ((k * d) & (N - 1)) << 22
            dw = (qa - qa_prev - e) & 0xFFFFFFFF
            dw = np.where(dw >= 2 ** 31, dw - 2 ** 32, dw)                     # int32 view of the wrapped difference
            adv = ((k * HOP) & (N - 1)) << 22
            scaled = (dw * R + (1 << 23)) >> 24
            qs = (qs + adv + scaled) & 0xFFFFFFFF
            assert d in (d0, d0 + 1)
        qa_prev, s_prev = qa, s
        ph = np.where(qs >= 2 ** 31, qs - 2 ** 32, qs) / 2.0 ** 32 * 2 * np.pi
        Y = np.abs(X) * np.exp(1j * ph)
        Y[0] = Y[0].real
        Y[-1] = Y[-1].real
        y = np.fft.irfft(Y, N)
        o = (f - 1) * HOP - N // 2
        lo, hi = max(o, 0), min(o + N, M)
        if hi > lo:
            v[lo:hi] += (w * y)[lo - o:hi - o]
    return v[:M] * (2.0 / 3.0)


def rs_table(rho):
    c = CUTOFF * (1.0 / rho if rho > 1.0 else 1.0)
    tab = np.zeros((PHASES + 1, TAPS))
    for p in range(PHASES + 1):
        xx = (np.arange(TAPS) - (TAPS // 2 - 1)) - p / PHASES
        a = xx / (TAPS / 2.0)
        wk = np.where(np.abs(a) < 1.0, np.i0(BETA * np.sqrt(np.maximum(1.0 - a * a, 0.0))) / np.i0(BETA),
                      np.where(np.abs(a) == 1.0, 1.0 / np.i0(BETA), 0.0))
        row = c * np.sinc(c * xx) * wk
        tab[p] = row / row.sum()
    return tab.astype(np.float32).astype(np.float64)          # the table is stored in float32 by specification


def transposer(v, pl, n_out):
    tab = rs_table(pl["rho"])
    out = np.zeros(n_out)
    for j in range(n_out):
        pos = j * pl["step"]
        idx, frac = pos >> 32, pos & 0xFFFFFFFF
        ph, alpha = frac >> 25, (frac & 0x1FFFFFF) / 33554432.0
        coef = tab[ph] + alpha * (tab[ph + 1] - tab[ph])
        m = idx - (TAPS // 2 - 1) + np.arange(TAPS)
        ok = (m >= 0) & (m < v.size)
        out[j] = np.dot(coef[ok], v[m[ok]])
    return out


def stretch(x, ch, rate, pitch):
    """interleaved [L*ch] -> interleaved [out_len*ch], float64"""
    x = np.asarray(x, np.float64).reshape(-1, ch)
    pl = plan(rate, pitch, x.shape[0])
    out = np.zeros((pl["out_len"], ch))
    for c in range(ch):
        s = x[:, c]
        if not pl["pv_on"] and not pl["rs_on"]:
            out[:, c] = s
        elif pl["rs_first"]:
            out[:, c] = vocoder(transposer(s, pl, pl["mid"]), pl, pl["out_len"])
        elif pl["pv_on"] and pl["rs_on"]:
            out[:, c] = transposer(vocoder(s, pl, pl["mid"]), pl, pl["out_len"])
        elif pl["pv_on"]:
            out[:, c] = vocoder(s, pl, pl["out_len"])
        else:
            out[:, c] = transposer(s, pl, pl["out_len"])
    return out.reshape(-1)
