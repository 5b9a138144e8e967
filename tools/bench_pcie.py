#!/usr/bin/env python3
"""Rate of the C5 graph when the boundary hands over HOST buffers (what the per-frame adapter does): H2D of the inputs, the
graph, D2H of the pitch and spectrum outputs.  Never the headline `value` (bench.py times HBM-resident data); bench.py
reports it as `pcie_inclusive`.

Two pipelines over page-locked staging (nae_malloc_host), both in chunks of `chunk` streams:
* mode "lanes" (rounds 2-4): three contexts — three streams — taking turns, each running upload, graph and both downloads of its
  chunk in order; the copies of different lanes overlap each other;
* mode "queues" (round 5): ONE upload context, ONE download context and the compute lanes between them, tied together by events
  (nae_event_record / nae_ctx_wait_event): every copy direction is a FIFO with one copy in flight at a time — a lone pinned copy
  runs at 57 GB/s where two or three concurrent downloads of different streams shared 44 — while the kernels of chunk i run
  beside the download of chunk i-1 and the upload of chunk i+1.
Per sample-frame 8 B go up and 24.03 B come down; the downlink is the bound: 63 GB/s (PCIe Gen5 x16) / 24.03 B = 2.6e9
sample-frames/s at best.

    python tools/bench_pcie.py [--streams 512] [--chunk 64]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


class Lane:
    """one context with its device buffers and pinned staging for `chunk` streams"""

    def __init__(self, nae, device, chunk, S, p, h_b, zero_copy=False):
        self.nae, self.chunk, self.S, self.zero_copy = nae, chunk, S, zero_copy
        self.ctx = ctx = nae.Context(device)
        self.pl = pl = ctx.stretch_plan(1.0, p, S)
        self.F = F = ctx.spectrum_frames(pl.out_len)
        self.h_in = ctx.pinned(chunk * S * 2)
        self.h_pitch = ctx.pinned(chunk * pl.out_len * 2)
        self.h_spec = ctx.pinned(chunk * F * 2 * 513)
        self.d_a, self.d_b = ctx.empty(chunk * S * 2), ctx.array(h_b)
        self.d_mix, self.d_pitch, self.d_spec = ctx.empty(chunk * S * 2), ctx.empty(chunk * pl.out_len * 2), ctx.empty(chunk * F * 2 * 513)
        g = self.g = nae.Graph4()
        g.in_a = nae.Sig.interleaved(self.d_a.ptr, S, 2)
        g.in_b = nae.Sig.interleaved(self.d_b.ptr, S, 2, shared=True)
        g.vol_a = g.vol_b = 0.5
        g.mix_out = nae.Sig.planar(self.d_mix.ptr, S, 2)
        g.rate, g.pitch = 1.0, p
        g.pitch_out = nae.Sig.interleaved(self.d_pitch.ptr, pl.out_len, 2)
        # zero_copy: the spectrum node writes straight into the page-locked host buffer (hipHostMalloc memory is device-visible),
        # so its 16 B per sample-frame cross PCIe while the kernel runs and only the pitch output needs a copy
        g.spec_out, g.spec_stream_stride = (self.h_spec.ctypes.data if zero_copy else self.d_spec.ptr), F * 2 * 513
        g.S, g.n_streams = S, chunk

    def enqueue(self):
        """upload, graph, download — all asynchronous on this lane's stream"""
        c = self.ctx
        c._ck(c.lib.nae_memcpy_h2d(c.h, self.d_a.ptr, self.h_in.ctypes.data, self.h_in.nbytes))
        c.graph4(self.g)
        c._ck(c.lib.nae_memcpy_d2h(c.h, self.h_pitch.ctypes.data, self.d_pitch.ptr, self.h_pitch.nbytes))
        if not self.zero_copy:
            c._ck(c.lib.nae_memcpy_d2h(c.h, self.h_spec.ctypes.data, self.d_spec.ptr, self.h_spec.nbytes))

    def close(self):
        for a in (self.h_in, self.h_pitch, self.h_spec):
            self.ctx.free_pinned(a)
        self.ctx.close()


def run_queues(nae, device, L, n_chunks):
    """the "queues" pipeline over the lanes' buffers: one upload stream, one download stream, events between them and the lanes"""
    lanes = len(L)
    up, down = nae.Context(device), nae.Context(device)
    ev_up = [up.event() for _ in L]                          # the lane's input has arrived
    ev_graph = [ln.ctx.event() for ln in L]                  # the lane's graph has run (its input buffer is free, its outputs are ready)
    ev_down = [down.event() for _ in L]                      # the lane's outputs have left the device buffers (and reached the host)

    def wait_host(ctx, ev):
        while not ctx.query(ev):
            time.sleep(0.0002)

    def chunk(i):
        k = i % lanes
        ln = L[k]
        if i >= lanes:
            wait_host(down, ev_down[k])                      # the caller has its results of chunk i - lanes; staging and device buffers are free
            up.wait_event(ev_graph[k])                       # (implied by the line above; kept for the device-side order)
        up._ck(up.lib.nae_memcpy_h2d(up.h, ln.d_a.ptr, ln.h_in.ctypes.data, ln.h_in.nbytes))
        up.record(ev_up[k])
        ln.ctx.wait_event(ev_up[k])
        ln.ctx.graph4(ln.g)
        ln.ctx.record(ev_graph[k])
        down.wait_event(ev_graph[k])
        down._ck(down.lib.nae_memcpy_d2h(down.h, ln.h_pitch.ctypes.data, ln.d_pitch.ptr, ln.h_pitch.nbytes))
        if not ln.zero_copy:
            down._ck(down.lib.nae_memcpy_d2h(down.h, ln.h_spec.ctypes.data, ln.d_spec.ptr, ln.h_spec.nbytes))
        down.record(ev_down[k])

    for i in range(lanes):                                   # untimed: first launches, first touches
        chunk(i)
    down.sync()
    t0 = time.perf_counter()
    for i in range(lanes, lanes + n_chunks):
        chunk(i)
    down.sync()
    dt = time.perf_counter() - t0
    # events go before the contexts they were created from (include/nae_gpu.h)
    for ctx_of, evs in ((up, ev_up), (down, ev_down)):
        for e in evs:
            ctx_of.destroy_event(e)
    for ln, e in zip(L, ev_graph):
        ln.ctx.destroy_event(e)
    up.close(); down.close()
    return dt


def measure(nae, device=0, streams=512, chunk=64, S=480000, semitones=3.0, lanes=3, zero_copy=False, mode="queues"):
    p = 2.0 ** (semitones / 12.0)
    h_b = np.random.default_rng(2).uniform(-1, 1, S * 2).astype(np.float32)
    L = [Lane(nae, device, chunk, S, p, h_b, zero_copy) for _ in range(lanes)]
    src = np.random.default_rng(1).uniform(-1, 1, chunk * S * 2).astype(np.float32)
    for ln in L:
        ln.h_in[:] = src                                   # (a real caller fills the staging buffer while the lane is busy)
    n_chunks = max(lanes, streams // chunk)
    if mode == "queues":
        dt = run_queues(nae, device, L, n_chunks)
    else:
        for ln in L:
            ln.enqueue()
        for ln in L:
            ln.ctx.sync()
        t0 = time.perf_counter()
        for i in range(n_chunks):
            ln = L[i % lanes]
            ln.ctx.sync()                                       # its previous chunk has left the staging buffers
            ln.enqueue()
        for ln in L:
            ln.ctx.sync()
        dt = time.perf_counter() - t0
    # serial reference on one lane: the three phases one after the other
    ln = L[0]
    c = ln.ctx
    t = [time.perf_counter()]
    c._ck(c.lib.nae_memcpy_h2d(c.h, ln.d_a.ptr, ln.h_in.ctypes.data, ln.h_in.nbytes)); c.sync(); t.append(time.perf_counter())
    c.graph4(ln.g); c.sync(); t.append(time.perf_counter())
    c._ck(c.lib.nae_memcpy_d2h(c.h, ln.h_pitch.ctypes.data, ln.d_pitch.ptr, ln.h_pitch.nbytes))
    c._ck(c.lib.nae_memcpy_d2h(c.h, ln.h_spec.ctypes.data, ln.d_spec.ptr, ln.h_spec.nbytes)); c.sync(); t.append(time.perf_counter())
    up, down = ln.h_in.nbytes, ln.h_pitch.nbytes + ln.h_spec.nbytes
    out = {"value": n_chunks * chunk * S / dt, "unit": "sample-frames/s", "streams": n_chunks * chunk, "chunk_streams": chunk, "lanes": lanes, "pipeline": mode,
           "staging": "page-locked (hipHostMalloc)", "spectrum_zero_copy": zero_copy, "seconds": round(dt, 4),
           "bytes_per_sample_frame": {"up": up / (chunk * S), "down": down / (chunk * S)},
           "down_GBps_in_pipeline": n_chunks * down / dt / 1e9, "up_GBps_in_pipeline": n_chunks * up / dt / 1e9,
           "one_chunk_serial": {"h2d_ms": (t[1] - t[0]) * 1e3, "graph_ms": (t[2] - t[1]) * 1e3, "d2h_ms": (t[3] - t[2]) * 1e3,
                                "h2d_GBps": up / (t[1] - t[0]) / 1e9, "d2h_GBps": down / (t[3] - t[2]) / 1e9,
                                "sample_frames_per_s": chunk * S / (t[3] - t[0])},
           "bound": "downlink: 24.03 B per sample-frame over PCIe Gen5 x16 (63 GB/s spec) = 2.6e9 sample-frames/s"}
    for ln in L:
        ln.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--zero-copy", action="store_true", help="spectrum output written by the kernel straight into pinned host memory")
    ap.add_argument("--mode", default="queues", choices=["queues", "lanes"])
    a = ap.parse_args()
    import naeload
    print(json.dumps(measure(naeload.load(), streams=a.streams, chunk=a.chunk, lanes=a.lanes, zero_copy=a.zero_copy, mode=a.mode)))
