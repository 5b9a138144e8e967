"""nodey-audio-editor_amd — MI355X-native per-node audio DSP (gain, split/merge, mix, tempo/pitch, FFT spectrum).

The product is ``libnae_gpu.so`` (HIP kernels for gfx950 behind the C ABI of ``include/nae_gpu.h``) plus the
C++ adapter classes in ``host/`` that mirror the reference's ``infra::Processor`` plugin interface.  This Python
package is only the ctypes binding the tests and ``bench.py`` drive the C ABI with.  It never computes on the
CPU: if the shared library is missing or HIP is unusable, it raises.

The directory name carries a hyphen (it mirrors the upstream repository name), so import it through
``naeload.load()`` at the repo root, which registers it as ``nodey_audio_editor_amd``.
"""
from .binding import (  # noqa: F401
    NaeError, Context, DeviceArray, Sig, StretchPlan, WsolaPlan, Graph4, lib_path, load_library, build_library,
    FMT_S16, FMT_S32, FMT_FLT, FMT_S16P, FMT_S32P, FMT_FLTP, FFT_N, HOP, BINS, EXPORTED_SYMBOLS,
)
