"""Independent numpy-float32 restatement of the SoundTouch-shaped chain (WSOLA stretcher, 64-tap anti-alias FIR, cubic
transposer) in BLOCK form: whole arrays and index arithmetic instead of the FIFOs of oracle/orc_wsola.c.  Used by
gen_golden.py to author tests/golden/wsola_golden.npz, which pins the C oracle "by construction" the way nodes.npz pins
K1-K6: two restatements written separately must agree bit for bit.  It pins nothing against SoundTouch itself (the
library is not available here: PARITY UNPINNED, see the oracle's header).

Float discipline: every product and every partial sum is a float32 operation in the order the x86 SSE build performs it
(np.cumsum accumulates sequentially in its dtype; np.sum does not, so it is never used on samples)."""
import numpy as np

f32 = np.float32


def _seq_sum(a, axis):
    """sequential float32 sum along `axis` (first element + second + ...), as a scalar loop would do it"""
    return np.take(np.cumsum(a.astype(f32), axis=axis, dtype=f32), -1, axis=axis)


def td_params(sr, tempo):
    seq = 90.0 + (40.0 - 90.0) / 1.5 * (tempo - 0.5)
    seek = 20.0 + (15.0 - 20.0) / 1.5 * (tempo - 0.5)
    seq_ms = int(min(max(seq, 40.0), 90.0) + 0.5)
    seek_ms = int(min(max(seek, 15.0), 20.0) + 0.5)
    ovl = max(sr * 8 // 1000, 16)
    ovl -= ovl % 8
    swl = max(sr * seq_ms // 1000, 2 * ovl)
    seekl = sr * seek_ms // 1000
    nominal = tempo * (swl - ovl)
    req = max(int(nominal + 0.5) + ovl, swl) + seekl
    return ovl, swl, seekl, nominal, req


def td_block(x, ch, sr, tempo):
    """x: [n][ch] float32 -> (stretched [m][ch], offsets)"""
    ovl, swl, seekl, nominal, req = td_params(sr, tempo)
    body = swl - 2 * ovl
    n = x.shape[0]
    out, offs = [], []
    mid = np.zeros((ovl, ch), f32)
    pos, skip, first = 0, 0.0, True
    idx = np.arange(seekl)[:, None] + np.arange(ovl)[None, :]                  # candidate i, frame j -> window frame
    u = (2.0 * np.arange(seekl) - seekl) / float(seekl)
    weight = 1.0 - 0.25 * u * u                                               # float64, as the library computes it
    while n - pos >= req:
        offset = 0
        if not first:
            win = x[pos:pos + seekl + ovl]
            cand = win[idx].reshape(seekl, ovl * ch)                          # [seekl][ovl*ch] floats in memory order
            prod = (cand * mid.reshape(-1)[None, :]).astype(f32)
            sq = (cand * cand).astype(f32)
            lanes_c = _seq_sum(prod.reshape(seekl, -1, 4), 1)                 # four partial sums over (index mod 4)
            lanes_n = _seq_sum(sq.reshape(seekl, -1, 4), 1)
            corr = ((lanes_c[:, 0] + lanes_c[:, 1]).astype(f32) + lanes_c[:, 2]).astype(f32) + lanes_c[:, 3]
            norm = ((lanes_n[:, 0] + lanes_n[:, 1]).astype(f32) + lanes_n[:, 2]).astype(f32) + lanes_n[:, 3]
            corr, norm = corr.astype(f32), norm.astype(f32)
            score = corr.astype(np.float64) / np.sqrt(np.where(norm < 1e-9, 1.0, norm.astype(np.float64)))
            score = (score + 0.1) * weight
            best = 0
            bv = score[0]
            for i in range(1, seekl):                                          # first maximum wins; NaN never wins
                if score[i] > bv:
                    bv, best = score[i], i
            offs.append(best)
            seg = win[best:best + ovl]
            if ch == 2:
                step = f32(1.0) / f32(ovl)
                f1 = np.concatenate([[f32(0)], np.cumsum(np.full(ovl - 1, step, f32), dtype=f32)])
                f2 = np.empty(ovl, f32)
                acc = f32(1.0)
                for i in range(ovl):
                    f2[i] = acc
                    acc = f32(acc - step)
                mixed = ((seg * f1[:, None]).astype(f32) + (mid * f2[:, None]).astype(f32)).astype(f32)
            else:
                m1 = np.arange(ovl, dtype=f32)[:, None]
                m2 = (f32(ovl) - np.arange(ovl, dtype=f32))[:, None]
                mixed = (((seg * m1).astype(f32) + (mid * m2).astype(f32)).astype(f32) / f32(ovl)).astype(f32)
            out.append(mixed)
            offset = best + ovl
        else:
            first = False
            skip -= float(int(tempo * ovl + 0.5 * seekl + 0.5))
            if skip <= -nominal:
                skip = -nominal
        out.append(x[pos + offset:pos + offset + body])
        mid = x[pos + offset + body:pos + offset + body + ovl].copy()
        skip += nominal
        adv = int(skip)
        skip -= adv
        pos += adv
    y = np.concatenate(out) if out else np.zeros((0, ch), f32)
    return y.astype(f32), np.array(offs, np.int32)


def aa_coefficients(rate):
    cutoff = 0.5 / rate if rate > 1.0 else 0.5 * rate
    t = np.arange(64, dtype=np.float64) - 32.0
    arg = t * (2.0 * np.pi * cutoff)
    with np.errstate(invalid="ignore", divide="ignore"):
        h = np.where(arg != 0.0, np.sin(arg) / arg, 1.0)
    work = (0.54 + 0.46 * np.cos((2.0 * np.pi / 64.0) * t)) * h
    total = 0.0
    for v in work:                                                             # running double sum, in order
        total += v
    v = work * (16384.0 / total)
    v = v + np.where(v >= 0.0, 0.5, -0.5)
    return (v.astype(f32) / f32(16384.0)).astype(f32)


def aa_block(x, ch, coef):
    """64-tap FIR: out[j] = sum_k x[j+k] c[k]; stereo: even and odd taps summed separately (an even count of outputs);
    mono: float products accumulated in float, in tap order"""
    n = x.shape[0]
    if n < 64:
        return np.zeros((0, ch), f32)
    count = n - 64
    if ch == 2:
        count &= ~1
        if count < 2:
            return np.zeros((0, ch), f32)
    win = np.lib.stride_tricks.sliding_window_view(x, 64, axis=0)[:count]       # [count][ch][64]
    prod = (win * coef[None, None, :]).astype(f32)
    if ch == 2:
        ev = _seq_sum(prod[:, :, 0::2], 2)
        od = _seq_sum(prod[:, :, 1::2], 2)
        return (od + ev).astype(f32)
    return _seq_sum(prod, 2).astype(f32)            # LONG_SAMPLETYPE is float in SoundTouch >= 2.1 float builds


def cu_block(x, ch, rate):
    """4-point cubic at positions advanced by `rate` in double precision; consumes while pos < n - 4"""
    k = np.array([[-0.5, 1.0, -0.5, 0.0], [1.5, -2.5, 0.0, 1.0], [-1.5, 2.0, 0.5, 0.0], [0.5, -0.5, 0.0, 0.0]], f32)
    n = x.shape[0]
    pos, fract, out = 0, 0.0, []
    while pos < n - 4:
        x2 = f32(fract)
        x1 = f32(x2 * x2)
        x0 = f32(x1 * x2)
        y = [f32(f32(f32(k[m, 0] * x0) + f32(k[m, 1] * x1)) + f32(k[m, 2] * x2)) + f32(k[m, 3] * f32(1.0)) for m in range(4)]
        y = [f32(v) for v in y]
        p = x[pos:pos + 4]
        o = (p[0] * y[0]).astype(f32)
        o = (o + (p[1] * y[1]).astype(f32)).astype(f32)
        o = (o + (p[2] * y[2]).astype(f32)).astype(f32)
        o = (o + (p[3] * y[3]).astype(f32)).astype(f32)
        out.append(o)
        fract += rate
        whole = int(fract)
        fract -= whole
        pos += whole
    return np.array(out, f32).reshape(-1, ch)


def process(x, ch, sr, rate_in, pitch):
    """one put of the whole signal, flush, receive all: [L*ch] interleaved -> ([out*ch], offsets)"""
    x = np.asarray(x, f32).reshape(-1, ch)
    L = x.shape[0]
    tempo, rate = 1.0 / pitch, pitch * rate_in
    expected = int(L / (rate * tempo) + 0.5)
    xp = np.concatenate([x, np.zeros((128 * 200, ch), f32)])                   # the flush rule's zero blocks, all of them
    coef = aa_coefficients(rate)
    if rate > 1.0:
        a, offs = td_block(xp, ch, sr, tempo)
        y = cu_block(aa_block(a, ch, coef), ch, rate)
    elif rate < 1.0:
        y, offs = td_block(aa_block(cu_block(xp, ch, rate), ch, coef), ch, sr, tempo)
    else:
        y, offs = td_block(cu_block(aa_block(xp, ch, coef), ch, rate), ch, sr, tempo)
    assert y.shape[0] >= expected, "flush padding too short for this case"
    return y[:expected].reshape(-1), offs
