"""round 5: read the per-wave section stamps of the patched spectrum kernel (tools/experiments/r05_spec_stamps.sh) at the C5 shape."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["NAE_GPU_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nodey-audio-editor_amd", "variants", "libnae_gpu_specstamps.so")
import naeload
nae = naeload.load()
n_streams, T, chunk = int(os.environ.get("N", 1024)), 480000, 32
with nae.Context(0) as ctx:
    F = ctx.spectrum_frames(T)
    d_x, d_o = ctx.empty(n_streams * T * 2), ctx.empty(n_streams * F * 2 * 513)
    ctx.fill_uniform(d_x.ptr, T * 2, T * 2, n_streams, 0, 0)
    sig = nae.Sig.interleaved(d_x.ptr, T, 2)
    for _ in range(3):
        ctx.spectrum_block(sig, T, 2, n_streams, d_o.ptr, F * 2 * 513)
    ctx.sync()
    out = d_o.download().reshape(n_streams, F, 1026)
names = ["staged reads -> stores issued", "window (Hann reads, multiplies) + loads issued", "channel 0 (FFT, split, magnitudes)",
         "channel 1", "staging writes", "wait for loads(f+1) / stores(f-1)"]
chunks = (F + chunk - 1) // chunk
raw = out[:, ::chunk, :24].copy().view(np.uint64).reshape(n_streams * chunks, 12)
st = raw[:, :6].astype(np.float64)
pro, loop, epi = raw[:, 6].astype(np.float64), raw[:, 7].astype(np.float64), raw[:, 8].astype(np.float64)
frames = np.minimum(chunk, F - np.arange(chunks) * chunk)
per_frame = st / np.tile(frames, n_streams)[:, None]
tot = per_frame.sum(1)
print(f"# spectrum_stereo_kernel<wide>, {n_streams} streams x {F} frames, chunk {chunk}: s_memtime cycles per frame and wave, by section (median / mean over {len(tot)} waves)")
print("| section | median | mean | share of the iteration |\n|---|---|---|---|")
for i, n in enumerate(names):
    print(f"| {n} | {np.median(per_frame[:, i]):.0f} | {per_frame[:, i].mean():.0f} | {per_frame[:, i].mean() / tot.mean():.2f} |")
print(f"| iteration | {np.median(tot):.0f} | {tot.mean():.0f} | 1.00 |")
print(f"\nper wave (cycles): prologue median {np.median(pro):.0f} mean {pro.mean():.0f} | loop median {np.median(loop):.0f} mean {loop.mean():.0f} | epilogue median {np.median(epi):.0f} mean {epi.mean():.0f}")
print(f"sum over waves of (prologue + loop + epilogue) = {(pro + loop + epi).sum():.4g} wave-cycles; loop share {loop.sum() / (pro + loop + epi).sum():.3f}")
print(f"wave-cycles / (16 waves x 256 CUs) = {(pro + loop + epi).sum() / 4096:.4g} cycles per wave slot (compare: kernel ms x clock)")
r0, r1 = raw[:, 10].astype(np.float64), raw[:, 11].astype(np.float64)
dur = (r1.max() - r0.min()) * 1e-8
clk = (pro + loop + epi) / np.maximum(r1 - r0, 1) * 0.1
print(f"kernel span by the 100-MHz clock: {dur * 1e3:.3f} ms; in-kernel shader clock median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
print(f"wave slots occupied: sum of wave durations / (4096 slots x span) = {(r1 - r0).sum() * 1e-8 / (4096 * dur):.3f}")
# how many waves are resident over time (20 bins)
edges = np.linspace(r0.min(), r1.max(), 21)
res = [((r0 < edges[i + 1]) & (r1 > edges[i])).sum() for i in range(20)]
print("waves touching each twentieth of the span:", res)
# waves of one workgroup = 8 consecutive items (item = stream * chunks_per_stream + chunk): the workgroup's LDS and wave slots are
# released only when its LAST wave exits
nw = (len(r0) // 8) * 8
e0, e1 = r0[:nw].reshape(-1, 8), r1[:nw].reshape(-1, 8)
wg_span = e1.max(1) - e0.min(1)
idle = (wg_span[:, None] - (e1 - e0)).sum() / (8 * wg_span.sum())
print(f"inside a workgroup: waves idle (entered late or exited early) {idle:.3f} of the workgroup's wave-slot time; exit spread median {np.median(e1.max(1) - e1.min(1)) * 10:.0f} ns of {np.median(wg_span) * 10:.0f} ns")
print(f"workgroup spans x 8 waves / (4096 slots x span) = {8 * wg_span.sum() * 1e-8 / (4096 * dur):.3f}")
