import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, naeload
nae = naeload.load()
ctx = nae.Context(0)
L3 = 3600 * 48000
p = 2 ** (3 / 12)
pl3 = ctx.stretch_plan(1.0, p, L3)
d_x, d_y = ctx.empty(L3 * 2), ctx.empty(pl3.out_len * 2)
ctx.fill_uniform(d_x.ptr, L3 * 2, 0, 1, 7, 0)
src, dst = nae.Sig.interleaved(d_x.ptr, L3, 2), nae.Sig.interleaved(d_y.ptr, pl3.out_len, 2)
for _ in range(2): ctx.stretch_block(1.0, p, src, L3, 2, 1, dst)
ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(3): ctx.stretch_block(1.0, p, src, L3, 2, 1, dst)
ctx.prof_enable(False)
r = ctx.prof_report()
print(os.environ.get("NAE_PV_TILE", "auto"), "frames", pl3.frames, "total %.2f ms" % sum(v[0] / v[1] for v in r.values()), {k: round(v[0] / v[1], 2) for k, v in r.items()})
