"""CPU: the SoundTouch-shaped oracle (oracle/orc_wsola.c, K7 option A / SURVEY.md §8f N1) behaves like a WSOLA
time-stretcher + transposer, is invariant to how the stream is chunked, and the product's host-side planner
(nae_wsola_plan_make: pure scalar bookkeeping, no GPU) predicts exactly the counts the oracle produces.

PARITY UNPINNED versus SoundTouch 2.3.2 itself: the library is not in the reference tree and the reference holds no
fixtures for it; these tests pin signal-level behaviour and the oracle <-> product agreement only."""
import numpy as np
import pytest

import orc

SR = 48000


def tone(L, f, sr=SR, ch=2, amp=0.5):
    t = np.arange(L) / sr
    x = amp * np.sin(2 * np.pi * f * t)
    return np.repeat(x[:, None], ch, 1).astype(np.float32).reshape(-1)


def peak_hz(y, sr=SR):
    seg = y[y.size // 4: y.size // 4 + 32768]
    sp = np.abs(np.fft.rfft(seg * np.hanning(seg.size)))
    return np.fft.rfftfreq(seg.size, 1 / sr)[int(np.argmax(sp))]


@pytest.mark.parametrize("rate,pitch", [(1.0, 2 ** (3 / 12)), (1.0, 2 ** (-5 / 12)), (1.5, 1 / 1.5), (0.8, 1 / 0.8),
                                        (1.25, 1.0), (0.8, 1.0), (1.0, 1.0)])
def test_length_pitch_and_tempo(rate, pitch):
    L = SR * 2
    x = tone(L, 440.0)
    y = orc.st_process(x, 2, SR, rate, pitch)
    n = y.size // 2
    assert n == int(L / rate + 0.5)                              # flush rule: round(in / (rate * tempo))
    assert abs(peak_hz(y[0::2]) - 440.0 * pitch * rate) < 3.0    # what SoundTouch's setRate / setPitch mean
    body = y[0::2][n // 8: -n // 8]
    assert 0.30 < np.sqrt(np.mean(body ** 2)) < 0.40             # a 0.5-amplitude sine stays one


@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("rate,pitch", [(1.0, 2 ** (3 / 12)), (0.8, 1.0), (1.0, 1.0)])
def test_chunking_does_not_change_the_stream(ch, rate, pitch):
    L = 40000
    x = orc.fill_uniform(L * ch, 5)
    whole = orc.st_process(x, ch, SR, rate, pitch)
    for chunk in (1152, 4096, 997):
        assert np.array_equal(orc.st_process(x, ch, SR, rate, pitch, chunk=chunk), whole), chunk


def test_parameters_follow_the_tempo():
    st = orc.SoundTouchChain(SR, 2, 1.0, 1.0)
    assert st.params() == dict(overlap=384, sequence=3504, seek=864, required=3504 + 864)   # 73 ms, 18 ms at tempo 1
    st.close()
    st = orc.SoundTouchChain(SR, 2, 1.0, 0.5)         # tempo 2: shortest sequence / seek window
    assert st.params()["sequence"] == 1920 and st.params()["seek"] == 720
    st.close()
    st = orc.SoundTouchChain(44100, 1, 1.0, 2.0)      # tempo 0.5: longest
    p = st.params()
    assert p["overlap"] == 352 and p["sequence"] == 3969 and p["seek"] == 882
    st.close()
    for sr in (7999, 48001):
        with pytest.raises(ValueError):
            orc.SoundTouchChain(sr, 2, 1.0, 1.0)


def test_anti_alias_filter_shape():
    st = orc.SoundTouchChain(SR, 2, 2.0, 1.0)         # rate 2: cutoff 0.25
    h = st.aa_coef().astype(np.float64)
    st.close()
    assert abs(h.sum() - 1.0) < 3e-3                  # unity DC gain up to the library's +-0.5/16384 per tap
    H = np.abs(np.fft.rfft(h, 4096))
    f = np.fft.rfftfreq(4096)
    assert H[f < 0.15].min() > 0.95 and H[f > 0.35].max() < 0.02


def test_cubic_weights_interpolate():
    w = np.zeros(4, np.float32)
    for x in (0.0, 0.25, 0.5, 0.999):
        orc.lib().orc_st_cubic_weights(np.float32(x), w.ctypes.data)
        assert abs(w.sum() - 1.0) < 1e-6
        # reproduces a straight line through the 4 points
        assert abs(np.dot(w, np.array([-1.0, 0.0, 1.0, 2.0])) - x) < 1e-6


@pytest.mark.parametrize("sr,ch,rate,pitch,L", [
    (48000, 2, 1.0, 2 ** (3 / 12), 100000), (48000, 2, 1.0, 2 ** (-4 / 12), 50000), (48000, 1, 1.5, 1 / 1.5, 30000),
    (44100, 2, 0.8, 1.0, 44100), (8000, 1, 1.0, 1.0, 20000), (48000, 2, 2.0, 1.0, 9000), (22050, 2, 1.0, 1.3, 500),
    (48000, 2, 1.0, 2 ** (3 / 12), 0)])
def test_host_planner_matches_oracle_counts(nae, sr, ch, rate, pitch, L):
    pl = nae.Context.wsola_plan(sr, ch, rate, pitch, L)
    x = orc.fill_uniform(max(L, 1) * ch, 9)[: L * ch]
    y, offs = orc.st_process(x, ch, sr, rate, pitch, want_offsets=True)
    st = orc.SoundTouchChain(sr, ch, rate, pitch)
    prm = st.params()
    st.close()
    assert (pl.overlap_len, pl.seq_len, pl.seek_len, pl.sample_req) == (prm["overlap"], prm["sequence"], prm["seek"], prm["required"])
    assert pl.out_len == y.size // ch
    assert pl.n_seq == (offs.size + 1 if pl.n_seq else 0)
    assert pl.order == (0 if rate * pitch > 1 else (1 if rate * pitch == 1 else 2))


def test_host_planner_rejects_what_the_reference_rejects(nae):
    for sr, ch in ((7000, 2), (96000, 2)):
        with pytest.raises(nae.NaeError):
            nae.Context.wsola_plan(sr, ch, 1.0, 1.0, 1000)
    with pytest.raises(nae.NaeError):
        nae.Context.wsola_plan(48000, 3, 1.0, 1.0, 1000)
    with pytest.raises(nae.NaeError):
        nae.Context.wsola_plan(48000, 2, 0.0, 1.0, 1000)


WSOLA_CASES = ("pitch_up3", "tempo_1p25", "pitch_down4", "mono_22k_down", "mono_8k_rate")


@pytest.mark.parametrize("name", WSOLA_CASES)
def test_oracle_matches_the_independent_numpy_restatement(golden, name):
    """tests/golden/wsola_golden.npz is authored by tests/golden/st_numpy.py (block form, numpy float32) — a second
    restatement written separately from the C oracle (streaming form).  Samples and offsets must agree bit for bit."""
    g = golden["wsola_golden"]
    ch, sr, rate, pitch = g[name + "_params"]
    x = g[str(g[name + "_src"])]
    y, offs = orc.st_process(x, int(ch), int(sr), float(rate), float(pitch), want_offsets=True)
    assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32))
    assert offs.size >= 2 and np.array_equal(offs, g[name + "_offsets"][: offs.size])
