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
st = out[:, ::chunk, :12].copy().view(np.uint64).reshape(n_streams * chunks, 6).astype(np.float64)
frames = np.minimum(chunk, F - np.arange(chunks) * chunk)
per_frame = st / np.tile(frames, n_streams)[:, None]
tot = per_frame.sum(1)
print(f"# spectrum_stereo_kernel<wide>, {n_streams} streams x {F} frames, chunk {chunk}: s_memtime cycles per frame and wave, by section (median / mean over {len(tot)} waves)")
print("| section | median | mean | share of the iteration |\n|---|---|---|---|")
for i, n in enumerate(names):
    print(f"| {n} | {np.median(per_frame[:, i]):.0f} | {per_frame[:, i].mean():.0f} | {per_frame[:, i].mean() / tot.mean():.2f} |")
print(f"| iteration | {np.median(tot):.0f} | {tot.mean():.0f} | 1.00 |")
