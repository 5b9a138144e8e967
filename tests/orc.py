"""ctypes binding of the CPU oracle (oracle/libnae_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libnae_oracle.so")

FMT_S16, FMT_S32, FMT_FLT, FMT_S16P, FMT_S32P, FMT_FLTP = 1, 2, 3, 6, 7, 8
BINS = 513


class Plan(C.Structure):
    _fields_ = [("pv_on", C.c_int), ("rs_on", C.c_int), ("tempo_eff", C.c_double), ("rate_eff", C.c_double),
                ("ha_q24", C.c_int64), ("d0", C.c_int32), ("r_q24", C.c_uint32 * 2), ("step_q32", C.c_uint64),
                ("out_len", C.c_size_t), ("mid_len", C.c_size_t), ("frames", C.c_size_t), ("rs_first", C.c_int)]


class SwrPlan(C.Structure):
    _fields_ = [("in_rate", C.c_int), ("out_rate", C.c_int), ("filter_length", C.c_int), ("filter_alloc", C.c_int),
                ("phase_count", C.c_int), ("src_incr", C.c_int), ("dst_incr_div", C.c_int), ("dst_incr_mod", C.c_int),
                ("index0", C.c_longlong), ("factor", C.c_double)]


_lib = None


def build():
    r = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_atan2_q32.restype = C.c_int32
        L.orc_atan2_q32.argtypes = [C.c_float, C.c_float]
        L.orc_spectrum_frames.restype = C.c_size_t
        L.orc_spectrum_frames.argtypes = [C.c_size_t]
        L.orc_stretch_plan_make.argtypes = [C.c_double, C.c_double, C.c_size_t, C.POINTER(Plan)]
        L.orc_stretch_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_pv_synth_phase.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Plan), C.c_void_p]
        L.orc_hann1024.restype = C.POINTER(C.c_float)
        L.orc_rs_table.restype = C.POINTER(C.c_float)
        L.orc_rs_table.argtypes = [C.c_double]
        L.orc_fill_uniform.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.orc_st_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        L.orc_st_destroy.argtypes = [C.c_void_p]
        L.orc_st_put.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_st_available.restype = C.c_size_t
        L.orc_st_available.argtypes = [C.c_void_p]
        L.orc_st_receive.restype = C.c_size_t
        L.orc_st_receive.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_st_flush.argtypes = [C.c_void_p]
        L.orc_st_offsets.restype = C.c_size_t
        L.orc_st_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_st_params.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_st_aa_coef.restype = C.POINTER(C.c_float)
        L.orc_st_aa_coef.argtypes = [C.c_void_p]
        L.orc_st_cubic_weights.argtypes = [C.c_float, C.c_void_p]
        L.orc_swr_plan_make.argtypes = [C.c_int, C.c_int, C.POINTER(SwrPlan)]
        L.orc_swr_build_filter.argtypes = [C.POINTER(SwrPlan), C.c_void_p]
        L.orc_swr_out_len.restype = C.c_size_t
        L.orc_swr_out_len.argtypes = [C.POINTER(SwrPlan), C.c_size_t]
        L.orc_swr_outputs_upto.restype = C.c_size_t
        L.orc_swr_outputs_upto.argtypes = [C.POINTER(SwrPlan), C.c_size_t]
        L.orc_swr_resample_f32.restype = C.c_size_t
        L.orc_swr_resample_f32.argtypes = [C.POINTER(SwrPlan), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pp(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def change_volume(planes, volume):
    """planes: list of 1-D arrays of one dtype (f32 / i16 / i32).  Returns new planes."""
    dt = planes[0].dtype
    fn = {np.dtype(np.float32): lib().orc_change_volume_f32, np.dtype(np.int16): lib().orc_change_volume_s16,
          np.dtype(np.int32): lib().orc_change_volume_s32}[dt]
    src = [np.ascontiguousarray(p) for p in planes]
    dst = [np.empty_like(p) for p in src]
    fn(_pp(dst), _pp(src), C.c_int(len(src)), C.c_int(src[0].size), C.c_float(volume))
    return dst


def interleave(planes):
    S, ch = planes[0].size, len(planes)
    src = [np.ascontiguousarray(p, np.float32) for p in planes]
    dst = np.empty(S * ch, np.float32)
    lib().orc_interleave_f32(_pp(src), _p(dst), C.c_size_t(S), C.c_int(ch))
    return dst


def deinterleave(x, ch):
    x = np.ascontiguousarray(x, np.float32)
    S = x.size // ch
    dst = [np.empty(S, np.float32) for _ in range(ch)]
    lib().orc_deinterleave_f32(_p(x), _pp(dst), C.c_size_t(S), C.c_int(ch))
    return dst


def amix(inL, inR, vol):
    n, S = len(inL), inL[0].size
    inL = [np.ascontiguousarray(a, np.float32) for a in inL]
    inR = [np.ascontiguousarray(a, np.float32) for a in inR]
    v = np.asarray(vol, np.float32)
    oL, oR = np.empty(S, np.float32), np.empty(S, np.float32)
    lib().orc_amix_f32(_pp(inL), _pp(inR), _p(v), C.c_int(n), _p(oL), _p(oR), C.c_size_t(S))
    return oL, oR


def amix_normalise(volumes, locks):
    v = np.array(volumes, np.float32)
    l = np.array(locks, np.uint8)
    lib().orc_amix_normalise_volumes(_p(v), _p(l), C.c_int(v.size))
    return v


def bimix(ll, lr, rl, rr, bias):
    a = [np.ascontiguousarray(x, np.float32) for x in (ll, lr, rl, rr)]
    S = a[0].size
    oL, oR = np.empty(S, np.float32), np.empty(S, np.float32)
    lib().orc_bimix_f32(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), C.c_float(bias), _p(oL), _p(oR), C.c_size_t(S))
    return oL, oR


def bimix2_downmix(l, r):
    l, r = np.ascontiguousarray(l, np.float32), np.ascontiguousarray(r, np.float32)
    m = np.empty_like(l)
    lib().orc_bimix2_downmix_f32(_p(l), _p(r), _p(m), C.c_size_t(l.size))
    return m


def bimix2_interleave(earlier, later, unaligned, aligned, earlier_offset):
    e = np.ascontiguousarray(earlier, np.float32)
    l = np.ascontiguousarray(later if later is not None else np.zeros(1, np.float32), np.float32)
    dst = np.empty(2 * (unaligned + aligned), np.float32)
    lib().orc_bimix2_interleave_f32(_p(dst), _p(e), _p(l), C.c_size_t(unaligned), C.c_size_t(aligned),
                                    C.c_int(earlier_offset))
    return dst


def to_f32_interleaved(fmt, planes, S, ch):
    planes = [np.ascontiguousarray(p) for p in planes]
    dst = np.empty(S * ch, np.float32)
    rc = lib().orc_to_f32_interleaved(C.c_int(fmt), _pp(planes), C.c_size_t(S), C.c_int(ch), _p(dst))
    return rc, dst


def clamp(x):
    y = np.array(x, np.float32)
    lib().orc_clamp_f32(_p(y), C.c_size_t(y.size))
    return y


def rfft1024(xw):
    xw = np.ascontiguousarray(xw, np.float32)
    X = np.empty(2 * BINS, np.float32)
    lib().orc_rfft1024(_p(xw), _p(X))
    return X[0::2] + 1j * X[1::2]


def hann():
    return np.ctypeslib.as_array(lib().orc_hann1024(), shape=(1024,)).copy()


def spectrum(x, ch):
    """x: interleaved [T*ch] f32 -> [frames, ch, 513]"""
    x = np.ascontiguousarray(x, np.float32)
    T = x.size // ch
    F = lib().orc_spectrum_frames(T)
    out = np.empty((F, ch, BINS), np.float32)
    if F:
        lib().orc_spectrum_f32(_p(x), C.c_size_t(T), C.c_int(ch), _p(out))
    return out


def plan(rate, pitch, L):
    pl = Plan()
    rc = lib().orc_stretch_plan_make(rate, pitch, L, C.byref(pl))
    return rc, pl


def stretch(x, ch, rate, pitch):
    """x: interleaved [L*ch] -> interleaved [out_len*ch]"""
    x = np.ascontiguousarray(x, np.float32)
    L = x.size // ch
    rc, pl = plan(rate, pitch, L)
    assert rc == 0, rc
    out = np.empty(max(pl.out_len, 1) * ch, np.float32)
    rc = lib().orc_stretch_f32(_p(x), L, ch, rate, pitch, _p(out))
    assert rc == 0, rc
    return out[: pl.out_len * ch]


def pv_synth_phase(x, ch, rate, pitch):
    x = np.ascontiguousarray(x, np.float32)
    L = x.size // ch
    rc, pl = plan(rate, pitch, L)
    assert rc == 0 and pl.pv_on
    qs = np.empty((pl.frames, ch, BINS), np.int32)
    rc = lib().orc_pv_synth_phase(_p(x), L, ch, C.byref(pl), _p(qs))
    assert rc == 0
    return qs


class SoundTouchChain:
    """streaming handle of the SoundTouch-shaped oracle (orc_wsola.c)"""

    def __init__(self, sample_rate, ch, rate, pitch):
        self.h = C.c_void_p()
        self.ch = ch
        rc = lib().orc_st_create(sample_rate, ch, rate, pitch, C.byref(self.h))
        if rc:
            raise ValueError("orc_st_create rc=%d" % rc)

    def close(self):
        if self.h:
            lib().orc_st_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def put(self, x):
        x = np.ascontiguousarray(x, np.float32)
        lib().orc_st_put(self.h, _p(x), x.size // self.ch)

    def available(self):
        return lib().orc_st_available(self.h)

    def receive(self, n=None):
        n = self.available() if n is None else n
        out = np.empty(max(n, 1) * self.ch, np.float32)
        got = lib().orc_st_receive(self.h, _p(out), n)
        return out[: got * self.ch]

    def flush(self):
        lib().orc_st_flush(self.h)

    def offsets(self):
        n = lib().orc_st_offsets(self.h, None, 0)
        a = np.empty(max(n, 1), np.int32)
        lib().orc_st_offsets(self.h, _p(a), n)
        return a[:n]

    def params(self):
        v = np.zeros(4, np.int32)
        lib().orc_st_params(self.h, _p(v))
        return dict(overlap=int(v[0]), sequence=int(v[1]), seek=int(v[2]), required=int(v[3]))

    def aa_coef(self):
        return np.ctypeslib.as_array(lib().orc_st_aa_coef(self.h), shape=(64,)).copy()


def st_process(x, ch, sample_rate, rate, pitch, chunk=None, want_offsets=False):
    """whole buffer through the SoundTouch-shaped chain: put (optionally in chunks: one size or a repeating list of
    sizes, receiving what is ready after every put), flush, receive all"""
    x = np.ascontiguousarray(x, np.float32)
    st = SoundTouchChain(sample_rate, ch, rate, pitch)
    L = x.size // ch
    sizes = [L] if not chunk else (list(chunk) if isinstance(chunk, (list, tuple)) else [chunk])
    outs = []
    a = i = 0
    while a < L:
        step = max(min(sizes[i % len(sizes)], L - a), 1)
        i += 1
        st.put(x[a * ch:(a + step) * ch])
        a += step
        if chunk:
            outs.append(st.receive())
    st.flush()
    outs.append(st.receive())
    offs = st.offsets()
    st.close()
    y = np.concatenate(outs) if outs else np.zeros(0, np.float32)
    return (y, offs) if want_offsets else y


def swr_plan(in_rate, out_rate):
    pl = SwrPlan()
    rc = lib().orc_swr_plan_make(in_rate, out_rate, C.byref(pl))
    return rc, pl


def swr_bank(pl):
    bank = np.empty((pl.phase_count, pl.filter_alloc), np.float32)
    lib().orc_swr_build_filter(C.byref(pl), _p(bank))
    return bank


def swr_resample(x, in_rate, out_rate):
    """one channel, whole signal including what a drain delivers (N2 spec: oracle/orc_swr.c)"""
    x = np.ascontiguousarray(x, np.float32)
    rc, pl = swr_plan(in_rate, out_rate)
    assert rc == 0, rc
    bank = swr_bank(pl)
    n_out = lib().orc_swr_out_len(C.byref(pl), x.size)
    out = np.empty(max(n_out, 1), np.float32)
    got = lib().orc_swr_resample_f32(C.byref(pl), _p(bank), _p(x), x.size, 1, _p(out), 1)
    assert got == n_out
    return out[:n_out]


def fill_uniform(n, seed):
    a = np.empty(n, np.float32)
    lib().orc_fill_uniform(_p(a), C.c_size_t(n), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF))
    return a


def stream_seed(s, k=0):
    """stream s, input k -> seed"""
    return (0x9E3779B97F4A7C15 * (1 + s) + k) & 0xFFFFFFFFFFFFFFFF
