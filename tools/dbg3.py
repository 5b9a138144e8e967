import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload, orc
nae = naeload.load()
ctx = nae.Context(0)
from test_gpu_stft import gpu_stretch, stream_stretch
L, ch = 8000, 2
x = orc.fill_uniform(L * ch, 61)
rate, pitch = 1.0, float(np.float32(2 ** (3 / 12)))
blk, pl = gpu_stretch(ctx, nae, x, ch, rate, pitch)
y, early = stream_stretch(ctx, x, ch, rate, pitch, [1152])
d = np.flatnonzero(y != blk)
print("ndiff", d.size, "first", d[:6] // ch if d.size else None)
