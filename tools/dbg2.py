import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload, orc
nae = naeload.load()
ctx = nae.Context(0)
from test_gpu_stft import gpu_stretch, stream_stretch
L, ch = 60000, 2
x = orc.fill_uniform(L * ch, 61)
for rate, pitch in ((1.0, 2 ** (3 / 12)), (1.5, 1 / 1.5), (1.5, 1.0)):
    rate, pitch = float(np.float32(rate)), float(np.float32(pitch))
    blk, pl = gpu_stretch(ctx, nae, x, ch, rate, pitch)
    for sizes in ([L], [30000], [1152]):
        y, early = stream_stretch(ctx, x, ch, rate, pitch, sizes)
        d = np.flatnonzero(y[:min(y.size, blk.size)] != blk[:min(y.size, blk.size)])
        print(rate, round(pitch, 4), sizes, "sizes", y.size, blk.size, "ndiff", d.size, "first", (d[:3] // ch) if d.size else None, "early", early,
              "maxabs", float(np.abs(y[:blk.size] - blk[:y.size]).max()) if y.size == blk.size else None)
