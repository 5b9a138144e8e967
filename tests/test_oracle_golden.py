"""CPU: the oracle (oracle/*.c) against the golden vectors of tests/golden (independent numpy restatement of the
reference loops; scipy float64 DFT for the spectrum).  Bit-exact for K1..K6, <= 1e-4 relative RMS for K8."""
import numpy as np
import pytest

import orc
from conftest import rel_rms


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 2: np.uint16}[a.dtype.itemsize])


def assert_bits(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(bits(a), bits(b))


def test_splitmix_generator(golden):
    g = golden["nodes"]
    assert_bits(orc.fill_uniform(64, int(g["splitmix_seed"][0])), g["splitmix_out"])
    x = orc.fill_uniform(100000, 7)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(float(x.mean())) < 0.01


@pytest.mark.parametrize("vol", [0.0, 0.70710678, 1.0, 10.0])
def test_k1_gain_f32(golden, vol):
    g = golden["nodes"]
    x = g["k1_f32_in"]
    packed = orc.change_volume([x], vol)[0]                      # FLT: one plane of S*ch
    assert_bits(packed, g[f"k1_f32_v{vol}"])
    planar = orc.change_volume([x[0::2].copy(), x[1::2].copy()], vol)  # FLTP: ch planes of S
    assert_bits(planar[0], g[f"k1_f32_v{vol}"][0::2].copy())
    assert_bits(planar[1], g[f"k1_f32_v{vol}"][1::2].copy())


@pytest.mark.parametrize("vol", [0.5, 0.70710678, 1.0, 3.0, 10.0])
def test_k1_gain_s16(golden, vol):
    g = golden["nodes"]
    assert_bits(orc.change_volume([g["k1_s16_in"]], vol)[0], g[f"k1_s16_v{vol}"])


@pytest.mark.parametrize("vol", [0.5, 0.70710678, 1.0, 3.0])
def test_k1_gain_s32(golden, vol):
    g = golden["nodes"]
    assert_bits(orc.change_volume([g["k1_s32_in"]], vol)[0], g[f"k1_s32_v{vol}"])


def test_k2_split_merge_roundtrip(golden):
    x = golden["nodes"]["k1_f32_in"]
    L, R = orc.deinterleave(x, 2)
    assert_bits(L, x[0::2].copy())
    assert_bits(R, x[1::2].copy())
    assert_bits(orc.interleave([L, R]), x)
    assert_bits(orc.interleave([x]), x)  # mono


@pytest.mark.parametrize("n", [1, 2, 16])
def test_k3_amix(golden, n):
    g = golden["nodes"]
    ins = g["k3_in"]
    L, R = orc.amix([ins[i][0::2] for i in range(n)], [ins[i][1::2] for i in range(n)], g[f"k3_vol_n{n}"])
    assert_bits(L, g[f"k3_L_n{n}"])
    assert_bits(R, g[f"k3_R_n{n}"])


def test_k3_volume_normalisation():
    v = orc.amix_normalise([0.2, 0.3, 0.5, 0.4], [0, 1, 0, 0])
    assert v[1] == np.float32(0.3)
    s = np.float32(0.2) + np.float32(0.5) + np.float32(0.4)
    assert v[0] == np.float32(0.2) / s and v[3] == np.float32(0.4) / s
    # all locked: divisor clamps to 0.001 and nothing changes
    assert np.array_equal(orc.amix_normalise([0.5, 0.5], [1, 1]), np.array([0.5, 0.5], np.float32))


@pytest.mark.parametrize("bias", [-1.0, -0.3, 0.0, 0.25, 1.0])
def test_k4_bimix(golden, bias):
    g = golden["nodes"]
    q = g["k4_in"]
    L, R = orc.bimix(q[0], q[1], q[2], q[3], bias)
    assert_bits(L, g[f"k4_L_b{bias}"])
    assert_bits(R, g[f"k4_R_b{bias}"])


def test_k5_bimix2(golden):
    g = golden["nodes"]
    q = g["k4_in"]
    assert_bits(orc.bimix2_downmix(q[0], q[1]), g["k5_mono"])
    assert_bits(orc.bimix2_interleave(q[2], q[3], 37, 500, 0), g["k5_inter_e0"])
    assert_bits(orc.bimix2_interleave(q[2], q[3], 37, 500, 1), g["k5_inter_e1"])
    tail = orc.bimix2_interleave(q[2][:100], None, 100, 0, 0)   # single-sided tail (audio-bimix.cpp:736-742)
    assert np.array_equal(tail[0::2], q[2][:100]) and not tail[1::2].any()


def test_k6_to_f32(golden):
    g = golden["nodes"]
    a16, a32, af = g["k6_s16_planes"], g["k6_s32_planes"], g["k6_f32_planes"]
    S = a16.shape[1]
    cases = [
        (orc.FMT_S16, [a16.T.reshape(-1).copy()], "k6_s16"), (orc.FMT_S16P, [a16[0].copy(), a16[1].copy()], "k6_s16p"),
        (orc.FMT_S32, [a32.T.reshape(-1).copy()], "k6_s32"), (orc.FMT_S32P, [a32[0].copy(), a32[1].copy()], "k6_s32p"),
        (orc.FMT_FLTP, [af[0].copy(), af[1].copy()], "k6_fltp"),
    ]
    for fmt, planes, key in cases:
        rc, out = orc.to_f32_interleaved(fmt, planes, S, 2)
        assert rc == 0
        assert_bits(out, g[key])
    rc, out = orc.to_f32_interleaved(orc.FMT_FLT, [g["k6_fltp"].copy()], S, 2)
    assert rc == 0
    assert_bits(out, g["k6_fltp"])
    assert orc.to_f32_interleaved(4, [af[0].copy()], S, 1)[0] == -1      # AV_SAMPLE_FMT_DBL: unsupported
    # the planar divisors really differ from the packed ones (32767 vs 32768)
    assert not np.array_equal(g["k6_s16"], g["k6_s16p"])


def test_clamp():
    x = np.array([-2.0, -1.0, -0.5, 0.0, 0.99, 1.0, 1.5, np.inf, -np.inf], np.float32)
    assert np.array_equal(orc.clamp(x), np.clip(x, -1, 1))
    assert np.isnan(orc.clamp(np.array([np.nan], np.float32))[0])  # std::clamp passes NaN through


def test_fft_against_numpy():
    rng = np.random.default_rng(1)
    for _ in range(4):
        x = rng.standard_normal(1024).astype(np.float32)
        X = orc.rfft1024(x)
        ref = np.fft.rfft(x.astype(np.float64))
        assert np.abs(X - ref).max() / np.abs(ref).max() < 5e-7
    hann = orc.hann()
    assert np.allclose(hann, 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024), atol=1e-7)


@pytest.mark.parametrize("name", ["tone", "noise", "impulse"])
def test_k8_spectrum_vs_float64_dft(golden, name):
    g = golden["spectrum"]
    x = g[f"{name}_in"]
    ref = g[f"{name}_mag"]                               # [F, 513] float64
    got = orc.spectrum(x, 1)[:, 0, :]
    assert got.shape == ref.shape
    assert rel_rms(got, ref) <= 1e-4                     # tolerance stated by BASELINE.json north_star
    assert rel_rms(got, ref) <= 2e-6                     # what f32 actually achieves
    stereo = np.stack([x, -0.5 * x], 1).reshape(-1)
    got2 = orc.spectrum(stereo, 2)
    assert rel_rms(got2[:, 0, :], ref) <= 2e-6 and rel_rms(got2[:, 1, :], 0.5 * ref) <= 2e-6


def test_k8_edge_lengths():
    assert orc.spectrum(np.zeros(1023, np.float32), 1).shape == (0, 1, 513)
    assert orc.spectrum(np.zeros(1024, np.float32), 1).shape == (1, 1, 513)
    assert orc.spectrum(np.zeros(1279, np.float32), 1).shape == (1, 1, 513)
    assert orc.spectrum(np.zeros(1280, np.float32), 1).shape == (2, 1, 513)
    assert orc.spectrum(np.zeros(0, np.float32), 2).shape == (0, 2, 513)


def test_atan2_q32():
    L = orc.lib()
    rng = np.random.default_rng(2)
    worst = 0.0
    for a, b in rng.standard_normal((5000, 2)).astype(np.float32):
        q = L.orc_atan2_q32(a, b) / 2.0 ** 32
        r = np.arctan2(np.float64(a), np.float64(b)) / (2 * np.pi)
        worst = max(worst, abs((q - r + 0.5) % 1.0 - 0.5))
    assert worst < 2e-7
    assert L.orc_atan2_q32(0.0, 0.0) == 0
    assert L.orc_atan2_q32(0.0, 1.0) == 0
    assert L.orc_atan2_q32(1.0, 0.0) == 2 ** 30           # quarter turn
    # revision 2 (include/nae_dsp_spec.h): the octants follow the sign bits and half a turn wraps (+1/2 == -1/2 turn)
    assert L.orc_atan2_q32(0.0, -1.0) == -2 ** 31
    assert L.orc_atan2_q32(-0.0, -1.0) == -2 ** 31
    assert L.orc_atan2_q32(0.0, -0.0) == -2 ** 31         # as C's atan2(+0, -0) = pi
    assert L.orc_atan2_q32(-1.0, 0.0) == -2 ** 30
    assert L.orc_atan2_q32(1.0, 1.0) in range(2 ** 29 - 400, 2 ** 29 + 400)
    assert L.orc_atan2_q32(1e-35, 1e-35) in range(0, 20000)  # a vanishing bin (below 2^-100) has a near-zero phase, not 1/8 turn
    # bins of 2^100 or more have phase 0 by specification (the integer reciprocal seed would wrap beyond 1.6e38): finite huge values,
    # Inf and NaN alike — and just below the cut the result is an ordinary phase
    for im, re in ((1.0, 3e38), (3e38, 1.0), (-3e38, -3e38), (2.0 ** 100, 0.0), (0.0, -(2.0 ** 100)), (np.inf, 1.0), (1.0, np.nan), (2e38, 2e38)):
        assert L.orc_atan2_q32(im, re) == 0, (im, re)
    assert L.orc_atan2_q32(2.0 ** 99, 0.0) == 2 ** 30 and L.orc_atan2_q32(0.0, -(2.0 ** 99)) == -2 ** 31
    assert abs(L.orc_atan2_q32(2.0 ** 98, 2.0 ** 98) - 2 ** 29) < 400


def test_k7_properties_and_regression(golden):
    g = golden["k7_regression"]
    x = g["in"]
    for name in ("pitch_up3", "tempo_1p5", "rate_0p8"):
        rate, pitch = g[name + "_params"]
        out = orc.stretch(x, 1, rate, pitch)
        assert out.size == int(np.floor(x.size / rate + 0.5))
        assert rel_rms(out, g[name]) < 1e-6               # oracle self-regression (not a reference pin)
    # identity parameters are a bit copy
    assert np.array_equal(orc.stretch(x, 1, 1.0, 1.0), x)


K7_GOLDEN = ("up3", "down4", "tempo_1p5", "tempo_0p7", "rate2_up5", "rate_0p8")


@pytest.mark.parametrize("name", K7_GOLDEN)
def test_k7_matches_the_specification_restatement(golden, name):
    """tests/golden/k7_golden.npz: float64 numpy restatement of DESIGN.md §3.3 (tests/golden/pv_numpy.py) — every stage
    order of vocoder and transposer, within BASELINE.json's float tolerance"""
    g = golden["k7_golden"]
    ch, rate, pitch = g[name + "_params"]
    out = orc.stretch(g[str(g[name + "_src"])], int(ch), float(rate), float(pitch))
    assert out.size == g[name].size
    assert rel_rms(out, g[name]) <= 1e-4


def test_k7_pitch_moves_a_tone():
    sr, n = 48000, 48000
    t = np.arange(n) / sr
    x = (0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    for semis in (-5, 3, 7):
        p = 2.0 ** (semis / 12)
        y = orc.stretch(x, 1, 1.0, p)
        assert y.size == n
        seg = y[8192:-8192].astype(np.float64)
        sp = np.abs(np.fft.rfft(seg * np.hanning(seg.size)))
        f = np.argmax(sp) * sr / seg.size
        assert abs(f - 1000 * p) < 3.0, (semis, f)
    # keep-pitch tempo change: length /v, tone stays
    y = orc.stretch(x, 1, 1.5, 1 / 1.5)
    assert y.size == 32000
    seg = y[8192:-8192].astype(np.float64)
    assert abs(np.argmax(np.abs(np.fft.rfft(seg * np.hanning(seg.size)))) * sr / seg.size - 1000) < 3.0
    # plain rate change: length /v, tone * v
    y = orc.stretch(x, 1, 1.5, 1.0)
    seg = y[4096:-4096].astype(np.float64)
    assert abs(np.argmax(np.abs(np.fft.rfft(seg * np.hanning(seg.size)))) * sr / seg.size - 1500) < 3.0


def test_k7_parameter_envelope():
    assert orc.plan(0.0, 1.0, 100)[0] != 0
    assert orc.plan(1.0, -1.0, 100)[0] != 0
    assert orc.plan(1.0, 1000.0, 100)[0] != 0
    rc, pl = orc.plan(1.0, 1.0, 12345)
    assert rc == 0 and not pl.pv_on and not pl.rs_on and pl.out_len == 12345
    rc, pl = orc.plan(2.0, 0.5, 1000)     # velocity 2, keep pitch: stretch only
    assert rc == 0 and pl.pv_on and not pl.rs_on and pl.out_len == 500
    rc, pl = orc.plan(1.0, 1.0, 0)
    assert rc == 0 and pl.out_len == 0


@pytest.mark.parametrize("in_rate", [44100, 22050, 96000, 8000, 88200])
def test_n2_swr_oracle_matches_float64_golden(golden, in_rate):
    """oracle/orc_swr.c (libswresample's default resampler, restated; UNPINNED versus FFmpeg) against the independent float64
    restatement of the same specification (tests/golden/swr_numpy.py): same output count, <= 1e-4 relative RMS"""
    g = golden["swr_golden"]
    for name in ("tones", "noise", "impulse"):
        y = orc.swr_resample(g[name + "_in"], in_rate, 48000)
        ref = g[f"{name}_{in_rate}_48000"]
        assert y.size == ref.size
        assert rel_rms(y, ref) <= 1e-4


def test_n2_swr_plan_and_edges():
    rc, pl = orc.swr_plan(44100, 48000)
    # exact_rational (libswresample's default): 48000 / 44100 = 160 / 147 -> 160 phases, every output advances 147 phase steps
    assert rc == 0 and (pl.filter_length, pl.phase_count, pl.src_incr, pl.dst_incr_div, pl.dst_incr_mod) == (32, 160, 1, 147, 0)
    rc, pl = orc.swr_plan(96000, 48000)                     # down-conversion: the filter stretches by 1 / (0.5 * 0.97); 1 / 2 -> one phase
    assert rc == 0 and pl.filter_length == 66 and pl.phase_count == 1
    rc, pl = orc.swr_plan(88200, 48000)                     # ceil(32 / 0.5279) = 61 taps, rounded up to an EVEN 62 as resample_init does
    assert rc == 0 and pl.filter_length == 62 and pl.phase_count == 80
    assert [orc.swr_plan(r, 48000)[1].phase_count for r in (22050, 8000, 32000, 11025)] == [320, 6, 3, 640]
    rc, pl = orc.swr_plan(44101, 48000)                     # a ratio that does not reduce keeps 2^phase_shift phases (nearest phase)
    assert rc == 0 and pl.phase_count == 1024
    assert orc.swr_plan(48000, 1000)[0] == -2                # 48x down: beyond NAE_SWR_MAX_TAPS
    assert orc.swr_resample(np.zeros(0, np.float32), 44100, 48000).size == 0
    # unit DC gain of every phase; a constant stays that constant away from the ends and, thanks to the reflection, at them
    y = orc.swr_resample(np.full(4000, 0.25, np.float32), 44100, 48000)
    assert y.size == 4354 and np.abs(y - 0.25).max() < 1e-6
    # one second in -> one second out
    assert orc.swr_resample(np.zeros(44100, np.float32), 44100, 48000).size == 48000
