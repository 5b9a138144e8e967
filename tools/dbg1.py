import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload, orc
nae = naeload.load()
ctx = nae.Context(0)
from test_gpu_stft import gpu_stretch
L, ch = 20000, 2
x = orc.fill_uniform(L * ch, 61)
p = 2 ** (3 / 12)
ref = orc.stretch(x, ch, 1.0, p)
def rr(a, b): return float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / np.sqrt(np.mean(b.astype(np.float64) ** 2)))
runs = [gpu_stretch(ctx, nae, x, ch, 1.0, p)[0] for _ in range(4)]
for i, r in enumerate(runs):
    print("block run", i, "vs oracle", rr(r, ref), "equal to run0:", np.array_equal(r, runs[0]))
lib = ctx.lib
h = C.c_void_p()
lib.nae_stretch_create(ctx.h, 48000, ch, 1.0, p, C.byref(h))
pos = 0
for n in (1152, 4096, 37, 9000, L):
    n = min(n, L - pos)
    chunk = np.ascontiguousarray(x[pos * ch:(pos + n) * ch])
    lib.nae_stretch_put_host(h, chunk.ctypes.data, n); pos += n
lib.nae_stretch_flush(h)
buf = np.empty(L * ch, np.float32); got = C.c_size_t()
lib.nae_stretch_receive_host(h, buf.ctypes.data, L, C.byref(got))
print("stream got", got.value, "vs oracle", rr(buf, ref), "equal run0", np.array_equal(buf, runs[0]))
d = np.flatnonzero(buf != runs[0]); print("ndiff", d.size, d[:10], d[-5:] if d.size else None)
# single put
h2 = C.c_void_p(); lib.nae_stretch_create(ctx.h, 48000, ch, 1.0, p, C.byref(h2))
lib.nae_stretch_put_host(h2, x.ctypes.data, L); lib.nae_stretch_flush(h2)
buf2 = np.empty(L * ch, np.float32)
lib.nae_stretch_receive_host(h2, buf2.ctypes.data, L, C.byref(got))
print("single put: vs oracle", rr(buf2, ref), "equal run0", np.array_equal(buf2, runs[0]))
