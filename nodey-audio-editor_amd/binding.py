"""ctypes binding of include/nae_gpu.h.  No CPU compute lives here — every method enqueues HIP work."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FMT_S16, FMT_S32, FMT_FLT, FMT_S16P, FMT_S32P, FMT_FLTP = 1, 2, 3, 6, 7, 8
FFT_N, HOP, BINS = 1024, 256, 513

# every entry point include/nae_gpu.h declares (tests check the library exports exactly these)
EXPORTED_SYMBOLS = [
    "nae_abi_version", "nae_device_count", "nae_ctx_create", "nae_ctx_destroy", "nae_ctx_set_stream",
    "nae_ctx_stream", "nae_sync", "nae_poll", "nae_last_error", "nae_device_name", "nae_malloc", "nae_free",
    "nae_memcpy_h2d", "nae_memcpy_d2h", "nae_memcpy_d2d", "nae_memset", "nae_event_create", "nae_event_record",
    "nae_event_query", "nae_ctx_wait_event", "nae_debug_graph4_stages",
    "nae_event_elapsed_ms", "nae_event_destroy", "nae_prof_enable", "nae_prof_reset", "nae_prof_get",
    "nae_malloc_host", "nae_free_host", "nae_debug_clock_ghz", "nae_debug_set", "nae_debug_diff_u32", "nae_fill_uniform_f32", "nae_gain_f32", "nae_gain_s16", "nae_gain_s32", "nae_gain_frame",
    "nae_deinterleave_f32", "nae_interleave_f32", "nae_copy_sig_f32", "nae_gain_sig_f32", "nae_amix_f32",
    "nae_amix_sig_f32", "nae_bimix_f32", "nae_bimix2_downmix_f32", "nae_bimix2_interleave_f32",
    "nae_to_f32_interleaved", "nae_clamp_f32", "nae_stretch_plan_make", "nae_stretch_block_f32",
    "nae_debug_pv_tile_phase", "nae_stretch_create", "nae_stretch_put", "nae_stretch_put_host", "nae_stretch_flush",
    "nae_stretch_available", "nae_stretch_receive", "nae_stretch_receive_host", "nae_stretch_destroy",
    "nae_swr_create", "nae_swr_convert_host", "nae_swr_convert", "nae_swr_buffered", "nae_swr_destroy", "nae_mono_to_stereo_f32",
    "nae_spectrum_frames", "nae_spectrum_block_f32", "nae_spectrum_create", "nae_spectrum_put",
    "nae_spectrum_available", "nae_spectrum_receive", "nae_spectrum_destroy", "nae_graph4_run",
    "nae_wsola_plan_make", "nae_wsola_block_f32", "nae_wsola_create", "nae_wsola_put", "nae_wsola_put_host",
    "nae_wsola_flush", "nae_wsola_available", "nae_wsola_receive", "nae_wsola_receive_host", "nae_wsola_destroy",
]


class NaeError(RuntimeError):
    pass


def lib_path() -> str:
    # NAE_GPU_LIB selects another build of the same ABI (A/B timing of kernel variants on one GPU box)
    return os.environ.get("NAE_GPU_LIB") or os.path.join(_HERE, "libnae_gpu.so")


def build_library(verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libnae_gpu.so (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _HERE, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise NaeError("building libnae_gpu.so failed")
    return lib_path()


class Sig(C.Structure):
    """nae_sig: element (s, c, i) at base[s*stream_stride + c*chan_stride + i*frame_stride]"""
    _fields_ = [("base", C.c_void_p), ("stream_stride", C.c_size_t), ("chan_stride", C.c_size_t),
                ("frame_stride", C.c_size_t)]

    @staticmethod
    def interleaved(ptr: int, S: int, ch: int, shared: bool = False, stream_stride: Optional[int] = None) -> "Sig":
        ss = 0 if shared else (S * ch if stream_stride is None else stream_stride)
        return Sig(ptr, ss, 1, ch)

    @staticmethod
    def planar(ptr: int, S: int, ch: int, shared: bool = False, plane_stride: Optional[int] = None,
               stream_stride: Optional[int] = None) -> "Sig":
        ps = S if plane_stride is None else plane_stride
        ss = 0 if shared else (ps * ch if stream_stride is None else stream_stride)
        return Sig(ptr, ss, ps, 1)


class StretchPlan(C.Structure):
    _fields_ = [("pv_on", C.c_int), ("rs_on", C.c_int), ("tempo_eff", C.c_double), ("rate_eff", C.c_double),
                ("ha_q24", C.c_int64), ("d0", C.c_int32), ("r_q24", C.c_uint32 * 2), ("step_q32", C.c_uint64),
                ("out_len", C.c_size_t), ("mid_len", C.c_size_t), ("frames", C.c_size_t), ("rs_first", C.c_int)]


class WsolaPlan(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("rate_eff", C.c_double), ("tempo_eff", C.c_double),
                ("order", C.c_int), ("overlap_len", C.c_int), ("seq_len", C.c_int), ("seek_len", C.c_int),
                ("sample_req", C.c_int), ("nominal_skip", C.c_double), ("in_len", C.c_size_t),
                ("flush_zeros", C.c_size_t), ("n_seq", C.c_size_t), ("td_out_len", C.c_size_t),
                ("aa_out_len", C.c_size_t), ("cu_out_len", C.c_size_t), ("out_len", C.c_size_t)]


class Graph4(C.Structure):
    _fields_ = [("in_a", Sig), ("in_b", Sig), ("vol_a", C.c_float), ("vol_b", C.c_float), ("mix_out", Sig),
                ("rate", C.c_double), ("pitch", C.c_double), ("pitch_out", Sig), ("spec_out", C.c_void_p),
                ("spec_stream_stride", C.c_size_t), ("S", C.c_size_t), ("n_streams", C.c_size_t)]


_lib = None


def load_library() -> C.CDLL:
    """Load libnae_gpu.so.  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise NaeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    lib = C.CDLL(p)
    vp, sz, i, f, d = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
    P = C.POINTER
    sigs = {
        "nae_abi_version": (i, []), "nae_device_count": (i, []),
        "nae_ctx_create": (i, [i, P(vp)]), "nae_ctx_destroy": (i, [vp]), "nae_ctx_set_stream": (i, [vp, vp]),
        "nae_ctx_stream": (vp, [vp]), "nae_sync": (i, [vp]), "nae_poll": (i, [vp]),
        "nae_last_error": (C.c_char_p, [vp]), "nae_device_name": (C.c_char_p, [vp]),
        "nae_malloc": (i, [vp, sz, P(vp)]), "nae_free": (i, [vp, vp]),
        "nae_memcpy_h2d": (i, [vp, vp, vp, sz]), "nae_memcpy_d2h": (i, [vp, vp, vp, sz]),
        "nae_memcpy_d2d": (i, [vp, vp, vp, sz]), "nae_memset": (i, [vp, vp, i, sz]),
        "nae_event_create": (i, [vp, P(vp)]), "nae_event_record": (i, [vp, vp]),
        "nae_event_query": (i, [vp]), "nae_ctx_wait_event": (i, [vp, vp]), "nae_debug_graph4_stages": (i, [vp, P(Graph4), i]),
        "nae_event_elapsed_ms": (i, [vp, vp, P(f)]), "nae_event_destroy": (i, [vp]),
        "nae_prof_enable": (i, [vp, i]), "nae_prof_reset": (i, [vp]),
        "nae_prof_get": (i, [vp, i, C.c_char_p, sz, P(d), P(C.c_uint64)]),
        "nae_malloc_host": (i, [vp, sz, P(vp)]), "nae_free_host": (i, [vp, vp]),
        "nae_debug_clock_ghz": (i, [vp, P(C.c_double)]), "nae_debug_set": (i, [vp, C.c_char_p, C.c_longlong]),
        "nae_debug_diff_u32": (i, [vp, vp, vp, sz, vp]),
        "nae_fill_uniform_f32": (i, [vp, vp, sz, sz, sz, C.c_uint64, C.c_uint64]),
        "nae_gain_f32": (i, [vp, P(vp), P(vp), i, sz, f]), "nae_gain_s16": (i, [vp, P(vp), P(vp), i, sz, f]),
        "nae_gain_s32": (i, [vp, P(vp), P(vp), i, sz, f]),
        "nae_gain_frame": (i, [vp, i, P(vp), P(vp), sz, i, f]),
        "nae_deinterleave_f32": (i, [vp, vp, P(vp), sz, i]), "nae_interleave_f32": (i, [vp, P(vp), vp, sz, i]),
        "nae_copy_sig_f32": (i, [vp, P(Sig), P(Sig), sz, i, sz]),
        "nae_gain_sig_f32": (i, [vp, P(Sig), P(Sig), sz, i, sz, f]),
        "nae_amix_f32": (i, [vp, P(vp), P(vp), P(f), i, vp, vp, sz]),
        "nae_amix_sig_f32": (i, [vp, P(Sig), P(f), i, P(Sig), sz, sz]),
        "nae_bimix_f32": (i, [vp, vp, vp, vp, vp, f, vp, vp, sz]),
        "nae_bimix2_downmix_f32": (i, [vp, vp, vp, vp, sz]),
        "nae_bimix2_interleave_f32": (i, [vp, vp, vp, vp, sz, sz, i]),
        "nae_to_f32_interleaved": (i, [vp, i, P(vp), sz, i, vp]), "nae_clamp_f32": (i, [vp, vp, sz]),
        "nae_stretch_plan_make": (i, [d, d, sz, P(StretchPlan)]),
        "nae_stretch_block_f32": (i, [vp, d, d, P(Sig), sz, i, sz, P(Sig)]),
        "nae_debug_pv_tile_phase": (i, [vp, d, d, P(Sig), sz, i, sz, vp, sz, P(sz), P(sz)]),
        "nae_stretch_create": (i, [vp, i, i, f, f, P(vp)]), "nae_stretch_put": (i, [vp, vp, sz]),
        "nae_stretch_put_host": (i, [vp, vp, sz]), "nae_stretch_flush": (i, [vp]),
        "nae_stretch_available": (sz, [vp]), "nae_stretch_receive": (i, [vp, vp, sz, P(sz)]),
        "nae_stretch_receive_host": (i, [vp, vp, sz, P(sz)]), "nae_stretch_destroy": (i, [vp]),
        "nae_swr_create": (i, [vp, i, i, i, i, P(vp)]),
        "nae_swr_convert_host": (i, [vp, P(vp), sz, vp, vp, sz, P(sz)]), "nae_swr_convert": (i, [vp, P(vp), sz, vp, vp, sz, P(sz)]),
        "nae_swr_buffered": (sz, [vp]),
        "nae_swr_destroy": (i, [vp]), "nae_mono_to_stereo_f32": (i, [vp, vp, vp, sz, f]),
        "nae_spectrum_frames": (sz, [sz]), "nae_spectrum_block_f32": (i, [vp, P(Sig), sz, i, sz, vp, sz]),
        "nae_spectrum_create": (i, [vp, i, i, i, P(vp)]), "nae_spectrum_put": (i, [vp, vp, sz]),
        "nae_spectrum_available": (sz, [vp]), "nae_spectrum_receive": (i, [vp, vp, sz, P(sz)]),
        "nae_spectrum_destroy": (i, [vp]), "nae_graph4_run": (i, [vp, P(Graph4)]),
        "nae_wsola_plan_make": (i, [i, i, d, d, sz, P(WsolaPlan)]),
        "nae_wsola_block_f32": (i, [vp, i, d, d, P(Sig), sz, i, sz, P(Sig), vp]),
        "nae_wsola_create": (i, [vp, i, i, d, d, P(vp)]), "nae_wsola_put": (i, [vp, vp, sz]),
        "nae_wsola_put_host": (i, [vp, vp, sz]), "nae_wsola_flush": (i, [vp]), "nae_wsola_available": (sz, [vp]),
        "nae_wsola_receive": (i, [vp, vp, sz, P(sz)]), "nae_wsola_receive_host": (i, [vp, vp, sz, P(sz)]),
        "nae_wsola_destroy": (i, [vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class DeviceArray:
    """A typed device allocation owned by a Context (hipMalloc through nae_malloc)."""

    def __init__(self, ctx: "Context", shape: Sequence[int], dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        ctx._ck(ctx.lib.nae_malloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value
        ctx._allocs.add(self)

    @property
    def size(self) -> int:
        return self.nbytes // self.dtype.itemsize

    def at(self, elem_offset: int) -> int:
        return self.ptr + int(elem_offset) * self.dtype.itemsize

    def upload(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.nbytes, self.nbytes)
        self.ctx._ck(self.ctx.lib.nae_memcpy_h2d(self.ctx.h, self.ptr, host.ctypes.data, self.nbytes))
        self.ctx.sync()
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        self.ctx._ck(self.ctx.lib.nae_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes))
        self.ctx.sync()
        return out

    def zero(self) -> "DeviceArray":
        self.ctx._ck(self.ctx.lib.nae_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def free(self) -> None:
        if self.ptr:
            self.ctx.sync()
            self.ctx.lib.nae_free(self.ctx.h, self.ptr)
            self.ptr = 0
            self.ctx._allocs.discard(self)


def _ptr_array(ptrs: Iterable[int]):
    ptrs = list(ptrs)
    return (C.c_void_p * len(ptrs))(*ptrs)


class Context:
    """nae_ctx wrapper.  One per GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.nae_ctx_create(device, C.byref(h))
        if rc != 0:
            why = ("no such device, or a bad assignment in $NAE_DEBUG" if rc == -1 else "no usable HIP device")
            raise NaeError(f"nae_ctx_create(device={device}) failed with {rc}: {why} — this library has no CPU fallback")
        self.h = h
        self.device = device
        self._allocs = set()
        self._pinned = {}

    # -- plumbing
    def _ck(self, rc: int) -> None:
        if rc != 0:
            raise NaeError(f"nae error {rc}: {self.lib.nae_last_error(self.h).decode(errors='replace')}")

    def name(self) -> str:
        return self.lib.nae_device_name(self.h).decode()

    def sync(self) -> None:
        self._ck(self.lib.nae_sync(self.h))

    def poll(self) -> int:
        return self.lib.nae_poll(self.h)

    def set_stream(self, hip_stream: int) -> None:
        self._ck(self.lib.nae_ctx_set_stream(self.h, C.c_void_p(hip_stream)))

    def empty(self, shape, dtype=np.float32) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def array(self, host: np.ndarray) -> DeviceArray:
        host = np.ascontiguousarray(host)
        return DeviceArray(self, host.shape, host.dtype).upload(host)

    def close(self) -> None:
        if self.h:
            for a in list(self._allocs):
                a.free()
            self.lib.nae_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def debug_set(self, key: str, value: int) -> "Context":
        """tuning / A-B switch of this context (include/nae_gpu.h lists the keys); returns self so that calls chain"""
        self._ck(self.lib.nae_debug_set(self.h, key.encode(), int(value)))
        return self

    # -- events
    def event(self) -> C.c_void_p:
        e = C.c_void_p()
        self._ck(self.lib.nae_event_create(self.h, C.byref(e)))
        return e

    def destroy_event(self, ev) -> None:
        """events are destroyed BEFORE the context they were created from (include/nae_gpu.h)"""
        rc = self.lib.nae_event_destroy(ev)
        if rc:
            raise NaeError(f"nae_event_destroy failed: {rc}")

    def record(self, ev) -> None:
        self._ck(self.lib.nae_event_record(self.h, ev))

    def query(self, ev) -> int:
        """1 = the work in front of the event's last record is done, 0 = pending (never blocks)"""
        rc = self.lib.nae_event_query(ev)
        if rc < 0:
            raise NaeError(f"nae_event_query failed: {rc}")
        return rc

    def wait_event(self, ev) -> None:
        """work enqueued on this context from now on waits (on the device) for the event's last record"""
        self._ck(self.lib.nae_ctx_wait_event(self.h, ev))

    def elapsed_ms(self, a, b) -> float:
        ms = C.c_float()
        rc = self.lib.nae_event_elapsed_ms(a, b, C.byref(ms))
        if rc:
            raise NaeError(f"nae_event_elapsed_ms failed: {rc}")
        return float(ms.value)

    # -- per-kernel timing / synthetic input
    def prof_enable(self, on: bool = True) -> None:
        self._ck(self.lib.nae_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self) -> None:
        self._ck(self.lib.nae_prof_reset(self.h))

    def prof_report(self) -> dict:
        """{kernel: (total_ms, launches)} accumulated since the last reset"""
        out = {}
        n = self.lib.nae_prof_get(self.h, -1, None, 0, None, None)
        for k in range(n):
            name = C.create_string_buffer(128)
            ms, cnt = C.c_double(), C.c_uint64()
            self.lib.nae_prof_get(self.h, k, name, 128, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def pinned(self, shape, dtype=np.float32) -> np.ndarray:
        """numpy view of page-locked host memory (nae_malloc_host); release with free_pinned(array)"""
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._ck(self.lib.nae_malloc_host(self.h, max(nbytes, 16), C.byref(p)))
        buf = (C.c_char * max(nbytes, 16)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def free_pinned(self, arr: np.ndarray) -> None:
        p = self._pinned.pop(arr.ctypes.data, None)
        if p:
            self.sync()
            self.lib.nae_free_host(self.h, p)

    def diff_words(self, a: int, b: int, n_words: int, d_count: int) -> None:
        """adds the number of differing 32-bit words of device buffers a, b to the device uint64 at d_count (asynchronous)"""
        self._ck(self.lib.nae_debug_diff_u32(self.h, a, b, n_words, d_count))

    def clock_ghz(self) -> float:
        """shader clock the GPU holds right now (probe kernel on the context's stream)"""
        g = C.c_double()
        self._ck(self.lib.nae_debug_clock_ghz(self.h, C.byref(g)))
        return g.value

    def fill_uniform(self, dst: int, n_per_stream: int, stream_stride: int, n_streams: int, first_stream: int = 0,
                     input_index: int = 0) -> None:
        self._ck(self.lib.nae_fill_uniform_f32(self.h, dst, n_per_stream, stream_stride, n_streams, first_stream,
                                               input_index))

    # -- K1
    def gain(self, fmt_dtype, src_planes: Sequence[int], dst_planes: Sequence[int], elems: int, volume: float):
        fn = {np.dtype(np.float32): self.lib.nae_gain_f32, np.dtype(np.int16): self.lib.nae_gain_s16,
              np.dtype(np.int32): self.lib.nae_gain_s32}[np.dtype(fmt_dtype)]
        self._ck(fn(self.h, _ptr_array(src_planes), _ptr_array(dst_planes), len(src_planes), elems, volume))

    def gain_frame(self, fmt: int, src_planes, dst_planes, S: int, ch: int, volume: float) -> int:
        return self.lib.nae_gain_frame(self.h, fmt, _ptr_array(src_planes), _ptr_array(dst_planes), S, ch, volume)

    # -- K2
    def deinterleave(self, src: int, dst_planes: Sequence[int], S: int, ch: int):
        self._ck(self.lib.nae_deinterleave_f32(self.h, src, _ptr_array(dst_planes), S, ch))

    def interleave(self, src_planes: Sequence[int], dst: int, S: int, ch: int):
        self._ck(self.lib.nae_interleave_f32(self.h, _ptr_array(src_planes), dst, S, ch))

    def copy_sig(self, src: Sig, dst: Sig, S: int, ch: int, n_streams: int):
        self._ck(self.lib.nae_copy_sig_f32(self.h, C.byref(src), C.byref(dst), S, ch, n_streams))

    def gain_sig(self, src: Sig, dst: Sig, S: int, ch: int, n_streams: int, volume: float):
        self._ck(self.lib.nae_gain_sig_f32(self.h, C.byref(src), C.byref(dst), S, ch, n_streams, volume))

    # -- K3
    def amix(self, inL: Sequence[int], inR: Sequence[int], vol: Sequence[float], outL: int, outR: int, S: int):
        n = len(inL)
        v = (C.c_float * n)(*[float(x) for x in vol])
        self._ck(self.lib.nae_amix_f32(self.h, _ptr_array(inL), _ptr_array(inR), v, n, outL, outR, S))

    def amix_sig(self, inputs: Sequence[Sig], vol: Sequence[float], out: Sig, S: int, n_streams: int):
        n = len(inputs)
        arr = (Sig * n)(*inputs)
        v = (C.c_float * n)(*[float(x) for x in vol])
        self._ck(self.lib.nae_amix_sig_f32(self.h, arr, v, n, C.byref(out), S, n_streams))

    # -- K4 / K5 / K6
    def bimix(self, ll, lr, rl, rr, bias: float, outL, outR, S: int):
        self._ck(self.lib.nae_bimix_f32(self.h, ll, lr, rl, rr, bias, outL, outR, S))

    def bimix2_downmix(self, l, r, mono, S: int):
        self._ck(self.lib.nae_bimix2_downmix_f32(self.h, l, r, mono, S))

    def bimix2_interleave(self, dst, earlier, later, unaligned: int, aligned: int, earlier_offset: int):
        self._ck(self.lib.nae_bimix2_interleave_f32(self.h, dst, earlier, later, unaligned, aligned, earlier_offset))

    def to_f32_interleaved(self, fmt: int, planes: Sequence[int], S: int, ch: int, dst: int) -> int:
        return self.lib.nae_to_f32_interleaved(self.h, fmt, _ptr_array(planes), S, ch, dst)

    def clamp(self, data: int, n: int):
        self._ck(self.lib.nae_clamp_f32(self.h, data, n))

    # -- K7
    @staticmethod
    def stretch_plan(rate: float, pitch: float, in_len: int) -> StretchPlan:
        pl = StretchPlan()
        rc = load_library().nae_stretch_plan_make(rate, pitch, in_len, C.byref(pl))
        if rc:
            raise NaeError(f"nae_stretch_plan_make({rate}, {pitch}) failed: {rc}")
        return pl

    def stretch_block(self, rate: float, pitch: float, src: Sig, in_len: int, ch: int, n_streams: int, dst: Sig):
        self._ck(self.lib.nae_stretch_block_f32(self.h, rate, pitch, C.byref(src), in_len, ch, n_streams, C.byref(dst)))

    def debug_pv_tile_phase(self, rate: float, pitch: float, src: Sig, in_len: int, ch: int, n_streams: int):
        pl = self.stretch_plan(rate, pitch, in_len)
        cap = n_streams * ch * (pl.frames + 1) * BINS
        out = np.zeros(cap, np.int32)
        nt, tf = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.nae_debug_pv_tile_phase(self.h, rate, pitch, C.byref(src), in_len, ch, n_streams,
                                                  out.ctypes.data, cap, C.byref(nt), C.byref(tf)))
        return out[: n_streams * ch * nt.value * BINS].reshape(n_streams, ch, nt.value, BINS), tf.value

    # -- K7 option A: SoundTouch-shaped WSOLA chain
    @staticmethod
    def wsola_plan(sample_rate: int, ch: int, rate: float, pitch: float, in_len: int) -> WsolaPlan:
        pl = WsolaPlan()
        rc = load_library().nae_wsola_plan_make(sample_rate, ch, rate, pitch, in_len, C.byref(pl))
        if rc:
            raise NaeError(f"nae_wsola_plan_make({sample_rate}, {ch}, {rate}, {pitch}) failed: {rc}")
        return pl

    def wsola_block(self, sample_rate: int, rate: float, pitch: float, src: Sig, in_len: int, ch: int, n_streams: int,
                    dst: Sig, offsets_dbg: int = 0):
        self._ck(self.lib.nae_wsola_block_f32(self.h, sample_rate, rate, pitch, C.byref(src), in_len, ch, n_streams,
                                              C.byref(dst), offsets_dbg))

    # -- K8
    def spectrum_frames(self, T: int) -> int:
        return int(self.lib.nae_spectrum_frames(T))

    def spectrum_block(self, src: Sig, T: int, ch: int, n_streams: int, dst: int, dst_stream_stride: int):
        self._ck(self.lib.nae_spectrum_block_f32(self.h, C.byref(src), T, ch, n_streams, dst, dst_stream_stride))

    # -- graph
    def graph4(self, g: Graph4):
        self._ck(self.lib.nae_graph4_run(self.h, C.byref(g)))

    def graph4_stages(self, g: Graph4, mask: int):
        """stages of the graph: 1 = mix (+ transposer when first), 2 = rest of the pitch node, 4 = spectrum"""
        self._ck(self.lib.nae_debug_graph4_stages(self.h, C.byref(g), mask))
