import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload, orc
nae = naeload.load()
ctx = nae.Context(0)
from test_gpu_stft import gpu_stretch
for rate in (1.5, 2 ** (3 / 12), 0.8, 3.7):
    x = orc.fill_uniform(50000 * 2, 5)
    got, pl = gpu_stretch(ctx, nae, x, 2, rate, 1.0)
    ref = orc.stretch(x, 2, rate, 1.0)
    print(rate, "bit-equal:", np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "ndiff", int(np.count_nonzero(got != ref)), "maxabs", float(np.abs(got - ref).max()))
