#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

One "step" = one pass of the 4-node graph  input -> mix(2) -> pitch(+3 semitones) -> FFT spectrum  over every
stream of the job (BASELINE.json configs[4] / SURVEY.md §8d C5: 1024 independent 48 kHz stereo f32 streams of 10 s,
the second mix input shared by all streams).  The job is FIXED at `--total-streams` (default 1024) and sharded over
the GPUs — strong scaling, rank r owns streams [r*T/N, (r+1)*T/N) — with no data-path collective; the shared source
buffer is broadcast once over RCCL at setup.  (`--streams N` instead gives every rank N streams: weak scaling.)
Inputs are resident in HBM before the timed region.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher around it: starts the line above as a CHILD process (before this process has
                                          # touched HIP), relays rank 0's JSON line, exits non-zero with the child's stderr tail on failure

Prints ONE JSON line on rank 0 (see README / DESIGN.md §4 for the fields).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
SEMITONES = 3.0
BINS = 513
# vector issue: cycles per wave64 f32 instruction per SIMD (add/sub/mul/fma mix), measured by WALL CLOCK and by SQ counters
# (tools/ubench/valu_wallclock.hip -> profiles/r05_valu_wallclock.md): 4.6 for a lone wave, 2.15-2.2 for two or more waves per SIMD
# (= the chip guide's 2 cycles per wave64 instruction on a SIMD-32; v_fma_f32 130 TFLOP/s at the 2.13 GHz held under that load).
# Rounds 2-4 used an s_memtime table (1.26 at 8 waves, 1.94 at 4) that implied 195 TFLOP/s: superseded.
def valu_cycles_per_instr(waves_per_simd):
    return 4.6 if waves_per_simd <= 1 else 2.15


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--total-streams", type=int, default=1024, help="streams of the whole job, sharded over the GPUs (C5: 1024; strong scaling)")
    ap.add_argument("--streams", type=int, default=None, help="weak scaling instead: this many streams on EVERY GPU")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="also report ms per step over a back-to-back run of at least this long (0: skip)")
    ap.add_argument("--settle-seconds", type=float, default=0.5, help="untimed back-to-back steps in front of the warm-up until the clock has settled (0: none)")
    ap.add_argument("--seconds", type=float, default=10.0, help="length of every stream at 48 kHz (C5: 10 s)")
    ap.add_argument("--cpu-streams", type=int, default=128, help="streams the CPU-oracle baseline is timed on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--semitones", type=float, default=SEMITONES, help="pitch node setting (headline: +3)")
    ap.add_argument("--rate", type=float, default=1.0, help="SoundTouch setRate of the pitch node (headline: 1)")
    ap.add_argument("--no-alt", action="store_true", help="skip the side measurement of the SoundTouch-shaped pitch node")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffers-on-both-sides measurement (pcie_inclusive)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the plugin-boundary measurement (host_path)")
    ap.add_argument("--dry-run", action="store_true", help="rehearsal of the N > 1 plumbing on CPU (gloo): launcher, rendezvous, sharding, the broadcast, "
                                                           "the rank reports and the one JSON line — no GPU work, `value` null (tests/test_shard_gloo.py)")
    return ap.parse_args(argv)


def usable_cores():
    """CPUs this process may really use: its affinity mask, cut by a cgroup CPU quota if one is set (a quota leaves the mask at the host's count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return cores, {"host_cpus": os.cpu_count(), "affinity": n, "cgroup_cpu_quota": quota}


def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE in the environment): start the documented
    torch.distributed.run line as a CHILD process — this process has imported neither torch nor the HIP library, so nothing that has
    initialised a GPU forks or execs — relay rank 0's single JSON line, and on failure exit non-zero with the tail of the child's stderr
    (torch.distributed.run names the failing rank there).  One rank per GPU, as /root/reference/src/infra/runner.cpp:142-154 runs every
    branch by itself."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's only working mode on this pool's hosts
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        sys.stderr.write(f"bench.py --gpus {a.gpus}: the {a.gpus}-rank child job failed (exit code {r.returncode}, {len(lines)} result line(s)); "
                         "the tail of its stderr:\n" + r.stderr[-4000:] + "\n")
        sys.exit(r.returncode if r.returncode else 1)
    if os.environ.get("NAE_BENCH_VERBOSE"):
        sys.stderr.write(r.stderr[-4000:])
    print(lines[0])
    sys.stdout.flush()


def dry_run_rank(a, rank, world):
    """One rank of the CPU rehearsal (gloo): the same sharding, broadcast, rank reports and maximum over ranks as the GPU path, with a
    small host buffer in place of the shared source and made-up step times (10 ms x (rank + 1)); rank 0 prints the JSON line."""
    import torch
    import naeload
    naeload.load()
    from nodey_audio_editor_amd import shard
    if world != a.gpus:
        raise SystemExit(f"bench.py: launched as {world} rank(s) for --gpus {a.gpus}")
    dist = shard.init("gloo", rank, world)
    first, last = shard.strong_range(rank, world, a.total_streams) if a.streams is None else shard.stream_range(rank, a.streams)
    t_b = torch.full((4096,), float(rank == 0), dtype=torch.float32)
    bms, bbytes = shard.timed_broadcast_shared(dist, t_b, 0)
    assert float(t_b[0]) == 1.0
    ranks = shard.gather_rank_reports(dist, {"rank": rank, "first_stream": first, "last_stream": last, "streams": last - first,
                                             "ms_per_step": 10.0 * (rank + 1), "clock_GHz": 0.0, "broadcast_ms": bms, "broadcast_bytes": bbytes})
    elapsed = shard.max_over_ranks(dist, 0.010 * (rank + 1) * a.steps)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "stereo f32 sample-frames/s through the 4-node graph input->mix(2)->pitch->FFT-spectrum @48 kHz",
                          "value": None, "dry_run": True, "unit": "sample-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": elapsed / a.steps * 1e3, "scaling": "weak" if a.streams is not None else "strong",
                          "ranks": {"ranks_seen": ranks["ranks_seen"], "backend": "gloo (dry run)", "launched_world_size": world,
                                    "per_rank": ranks["per_rank"], "streams_covered": sum(d["streams"] for d in ranks["per_rank"])}}))
        sys.stdout.flush()
    dist.destroy_process_group()


def host_path(seconds=30.0):
    """The plugin boundary as the editor sees it (include/infra/processor.hpp:108-113): 1152-sample frames through
    audio_volume_adjust -> audio_amix(2) -> pitch_modifier in the C++ host mirror's fiber runner (one thread, every hop a host frame),
    one branch and 16 independent branches.  Runs tests/host/selftest (test code: it checks branch 0 against the oracle and times the
    oracle on the same frames).  Reported beside the headline, never part of `value`."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nodey-audio-editor_amd", "host"), "-j4"], capture_output=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host")], capture_output=True)
    if not os.path.exists(exe):
        return {"error": "tests/host/selftest is not built"}
    try:
        r = subprocess.run([exe, "bench", str(seconds)], capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "tests/host/selftest bench timed out"}
    runs = [json.loads(line[len("HOST_PATH "):]) for line in r.stdout.splitlines() if line.startswith("HOST_PATH ")]
    if r.returncode != 0 or len(runs) != 2:
        return {"error": f"selftest bench rc {r.returncode}", "tail": r.stdout[-400:]}
    return {"what": "C++ host mirror (nodey-audio-editor_amd/host): fiber runner on one thread, per-node contexts, every node batches the frames "
                    "already waiting behind one wait of its own stream; frames/s = 1152-sample source frames",
            "one_branch": runs[0], "sixteen_branches": runs[1]}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_run:
        return dry_run_rank(a, rank, world)
    dist = None
    torch = None
    # stdout carries exactly ONE line, the JSON result: until that line is printed, file descriptor 1 points at stderr
    # (RCCL writes a version banner to stdout when its first communicator is created)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import naeload
    if world > 1 or a.gpus > 1 or os.environ.get("NAE_FORCE_DIST"):
        # torch first: its bundled libamdhip64.so.7 is then the one HIP runtime of the process (same soname as
        # /opt/rocm's, so libnae_gpu.so binds to it and device pointers are interchangeable)
        import torch
        if world != a.gpus:
            raise SystemExit(f"bench.py: launched as {world} rank(s) for --gpus {a.gpus}")
        n_dev = torch.cuda.device_count()                    # (counting does not initialise a device)
        if local_rank >= n_dev:
            raise SystemExit(f"bench.py --gpus {a.gpus}: rank {rank} wants GPU {local_rank}, this box shows {n_dev} GPU(s)")
        torch.cuda.set_device(local_rank)
        nae = naeload.load()
        from nodey_audio_editor_amd import shard
        dist = shard.init("nccl", rank, world)               # "nccl" is RCCL on ROCm

    import numpy as np
    nae = naeload.load()
    from nodey_audio_editor_amd import shard
    ctx = nae.Context(local_rank)
    S = int(round(a.seconds * 48000))
    weak = a.streams is not None
    if weak:
        first_stream, last_stream = shard.stream_range(rank, a.streams)
    else:
        first_stream, last_stream = shard.strong_range(rank, world, a.total_streams)
    n_streams = last_stream - first_stream              # streams resident on THIS GPU
    if n_streams < 1:
        raise SystemExit(f"rank {rank}: no stream to process ({a.total_streams} streams over {world} GPUs)")
    pitch = 2.0 ** (a.semitones / 12.0)
    pl = ctx.stretch_plan(a.rate, pitch, S)
    F = ctx.spectrum_frames(pl.out_len)

    # ---- buffers (all HBM-resident before timing)
    d_a = ctx.empty(n_streams * S * 2)
    d_mix = ctx.empty(n_streams * S * 2)
    d_pitch = ctx.empty(n_streams * pl.out_len * 2)
    d_spec = ctx.empty(n_streams * F * 2 * BINS)
    ctx.fill_uniform(d_a.ptr, S * 2, S * 2, n_streams, first_stream, 0)
    if dist is not None:
        # shared second mix input: generated on rank 0, broadcast over RCCL/xGMI (the only collective)
        t_b = torch.empty(S * 2, dtype=torch.float32, device=f"cuda:{local_rank}")
        if rank == 0:
            ctx.fill_uniform(t_b.data_ptr(), S * 2, 0, 1, 0, 1)
            ctx.sync()
        shard.broadcast_shared(dist, torch.zeros(1, dtype=torch.float32, device=f"cuda:{local_rank}"), 0)   # untimed: creates the RCCL communicator
        bcast_ms, bcast_bytes = shard.timed_broadcast_shared(dist, t_b, 0, sync=torch.cuda.synchronize)
        b_ptr = t_b.data_ptr()
    else:
        d_b = ctx.empty(S * 2)
        ctx.fill_uniform(d_b.ptr, S * 2, 0, 1, 0, 1)
        b_ptr = d_b.ptr
    ctx.sync()

    g = nae.Graph4()
    g.in_a = nae.Sig.interleaved(d_a.ptr, S, 2)
    g.in_b = nae.Sig.interleaved(b_ptr, S, 2, shared=True)
    g.vol_a = g.vol_b = 0.5
    g.mix_out = nae.Sig.planar(d_mix.ptr, S, 2)
    g.rate, g.pitch = a.rate, pitch
    g.pitch_out = nae.Sig.interleaved(d_pitch.ptr, pl.out_len, 2)
    g.spec_out, g.spec_stream_stride = d_spec.ptr, F * 2 * BINS
    g.S, g.n_streams = S, n_streams

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    # untimed: the chip raises / lowers its clock over the first tenths of a second of a load (a small batch timed cold reads
    # 15-25 % slow: profiles/r03_strong_proxy.md); every rank runs the same wall time here, then the W warm-up steps
    if a.settle_seconds > 0:
        ts = time.perf_counter()
        while time.perf_counter() - ts < a.settle_seconds:
            for _ in range(4):
                ctx.graph4(g)
            ctx.sync()
    for _ in range(a.warmup):
        ctx.graph4(g)
    barrier()
    if not a.no_kernel_timing:
        ctx.prof_reset()
        ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.graph4(g)
    ctx.sync()
    if dist is not None:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    clock_ghz = ctx.clock_ghz()                            # probe kernel directly behind the timed steps
    kernels = {}
    if not a.no_kernel_timing:
        ctx.prof_enable(False)
        kernels = ctx.prof_report()
    sustained = None
    if a.sustain_seconds > 0:
        # the same step back to back for at least --sustain-seconds (clock and power settled), no per-kernel events
        n_sus, ts0 = 0, time.perf_counter()
        while True:
            for _ in range(20):
                ctx.graph4(g)
            ctx.sync()
            n_sus += 20
            if time.perf_counter() - ts0 >= a.sustain_seconds:
                break
        ts1 = time.perf_counter()
        sustained = {"seconds": round(ts1 - ts0, 2), "steps": n_sus, "ms_per_step": (ts1 - ts0) / n_sus * 1e3, "clock_GHz_after": round(ctx.clock_ghz(), 3)}
    ranks = None
    if dist is not None:
        # every rank's own figures, gathered before the maximum replaces them: the N > 1 line explains itself
        ranks = shard.gather_rank_reports(dist, {"rank": rank, "first_stream": first_stream, "last_stream": last_stream, "streams": n_streams,
                                                 "ms_per_step": elapsed / a.steps * 1e3, "clock_GHz": clock_ghz, "broadcast_ms": bcast_ms,
                                                 "broadcast_bytes": bcast_bytes}, device=f"cuda:{local_rank}")
        elapsed = shard.max_over_ranks(dist, elapsed, device=f"cuda:{local_rank}")
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    job_streams = world * n_streams if weak else a.total_streams
    value = shard.job_throughput(world, n_streams, S, a.steps, elapsed) if weak else shard.job_throughput_total(job_streams, S, a.steps, elapsed)

    # ---- roofline of the dominant kernel: algorithmic (compulsory) bytes of that launch / its mean duration
    sf = n_streams * S                                   # sample-frames one launch covers on this GPU
    mid_ratio = pl.mid_len / S
    alg_bytes = {                                        # per launch; DESIGN.md §4 derives each figure
        "amix_i2p_kernel": 24.0 * sf,                    # 2 x 8 B in + 8 B out            (mix n=2 = 24 B)
        "mix_resample_tile_kernel": 24.0 * sf + 8.0 * sf * mid_ratio,   # the mix node's 24 B + the transposed signal once
        "pv_phase_kernel": 8.0 * sf,                     # re-read of the input (pass 1); output negligible
        "pv_scan_kernel": 0.0,
        "pv_pipe_kernel": 8.0 * sf + 8.0 * sf * mid_ratio,    # input once + stretched signal once
        "pv_flow_kernel": 8.0 * sf + 8.0 * sf * mid_ratio,    # (the same node on at most one workgroup per CU: a rank's share of a multi-GPU job)
        "resample_kernel": 8.0 * sf * mid_ratio + 8.0 * sf,   # stretched signal once + output once
        "resample_tile_kernel": 8.0 * sf * mid_ratio + 8.0 * sf,
        "spectrum_stereo_kernel": 8.0 * sf + 2 * BINS * 4.0 * n_streams * F,   # 24.03 B per sample-frame
        "spectrum_kernel": 8.0 * sf + 2 * BINS * 4.0 * n_streams * F,          # (generic layouts)
    }
    roofline = None
    kern_report = {}
    for name, (ms, cnt) in kernels.items():
        avg = ms / max(cnt, 1)
        gbs = alg_bytes.get(name, 0.0) / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        kern_report[name] = {"avg_ms": round(avg, 4), "launches": int(cnt), "alg_GBps": round(gbs, 1),
                             "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    # HBM traffic and dynamic instruction counts of each kernel come from rocprofv3 --pmc passes of THIS build over the
    # same workload (tools/pmc_sq.sh, tools/summarize_prof.py), committed as profiles/traffic.json: counters cannot be
    # read from inside the run.  Both are scaled by the sample-frames of the launch.
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tdata = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k][0])
        avg_s = kernels[dom][0] / max(kernels[dom][1], 1) * 1e-3
        ach = alg_bytes.get(dom, 0.0) / avg_s / 1e9
        td = tdata.get(dom, {})
        if dom == "pv_flow_kernel" and not td and "pv_pipe_kernel" in tdata:
            # a rank's share of a multi-GPU job runs the one-barrier build of the same node: the same instructions, LDS traffic and bytes per frame
            # (profiles/r05_flow.md), one workgroup per CU = 4 waves per SIMD; the counters were collected on the full batch's kernel
            td = dict(tdata["pv_pipe_kernel"], waves_per_simd=4)
        scale = sf / td["sample_frames"] if td.get("sample_frames") else None
        traffic = td["hbm_bytes_per_launch"] * scale if scale and "hbm_bytes_per_launch" in td else None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": tdata.get("_source") if traffic is not None else None,
                    "avg_launch_ms": round(avg_s * 1e3, 4), "alg_bytes_per_launch": alg_bytes.get(dom, 0.0),
                    "note": "K7 is bound by the vector ALU (two 512-point FFTs, atan2, sin/cos per bin: busy `valu.frac` of the kernel at 2.15 cycles "
                            "per wave64 instruction), with the LDS array busy beside it, not by HBM (valu block; DESIGN.md §4 and profiles/r05_valu_wallclock.md)"}
        if scale and "valu_instr_per_launch" in td:
            # vector ALU: dynamic wave-instructions of the launch (SQ_INSTS_VALU) x 2.15 cycles per wave-instruction per SIMD (wall-clock
            # measured, profiles/r05_valu_wallclock.md) against the kernel's cycles at the clock measured in this run.  1024 SIMDs = 256 CUs x 4.
            # The LDS array's cycles (SQ_LDS_IDX_ACTIVE per CU) run BESIDE the vector ALU: the two overlap partially, they do not add.
            instr = td["valu_instr_per_launch"] * scale
            k_cyc = avg_s * clock_ghz * 1e9
            ach_cpi = k_cyc * 1024 / instr
            peak_cpi = valu_cycles_per_instr(td.get("waves_per_simd", 6))
            l_cyc = td["lds_idx_active_per_cu"] * scale if td.get("lds_idx_active_per_cu") else None
            roofline["valu"] = {"wave_instr_per_launch": instr, "clock_GHz": round(clock_ghz, 3), "waves_per_simd": td.get("waves_per_simd", 6),
                                "achieved": round(ach_cpi, 3), "peak": peak_cpi, "unit": "cycles per wave-instruction per SIMD (lower is better)",
                                "frac": round(peak_cpi / ach_cpi, 3),
                                "peak_source": "profiles/r05_valu_wallclock.md (hipEvent wall clock + SQ_INSTS_VALU / GRBM_GUI_ACTIVE)",
                                "note": "every instruction priced at the f32 add / mul / fma rate; by class (profiles/r05_valu_ops.md: compares, selects, conversions, "
                                        "integer multiplies hold the arithmetic path ~1.6 cycles and their own port 4.1; transcendentals 8.1) this kernel reads the same",
                                "valu_busy_cycles_per_simd": instr / 1024.0 * peak_cpi, "kernel_cycles": k_cyc,
                                "lds": {"idx_active_cycles_per_cu": l_cyc, "frac_of_kernel": round(l_cyc / k_cyc, 3) if l_cyc else None,
                                        "note": "runs beside the vector ALU (valu + lds > kernel cycles: partial overlap, not a sum)"},
                                "floor_ms_at_this_clock": round(instr / 1024.0 * peak_cpi / (clock_ghz * 1e9) * 1e3, 3)}
    chain_gbs = 64.03 * (n_streams * S * a.steps / elapsed) / 1e9   # 24 + 16 + 24.03 B per sample-frame, rank 0's GPU
    # the two memory-side kernels against what a PLAIN streaming kernel with their read : write mix reaches on this chip
    # (tools/ubench/rw_mix.hip -> profiles/r04_rw_mix.md, 4096 workgroups; nt stores in brackets): real HBM bytes (PMC passes) / duration
    PLAIN_GBPS = {"spectrum_stereo_kernel": ("1 read : 2 writes", 4670.0, 5080.0), "mix_resample_tile_kernel": ("2 reads : 3 writes", 4700.0, 4800.0)}
    memory_side = {}
    for name, (mix, plain, plain_nt) in PLAIN_GBPS.items():
        td = tdata.get(name, {})
        if name in kernels and td.get("hbm_bytes_per_launch") and td.get("sample_frames"):
            avg_s = kernels[name][0] / max(kernels[name][1], 1) * 1e-3
            gbs = td["hbm_bytes_per_launch"] * (sf / td["sample_frames"]) / avg_s / 1e9
            memory_side[name] = {"hbm_GBps": round(gbs, 1), "read_write_mix": mix, "plain_streaming_kernel_GBps": plain,
                                 "plain_streaming_kernel_nt_GBps": plain_nt, "frac_of_plain": round(gbs / plain, 3)}

    out = {
        "metric": "stereo f32 sample-frames/s through the 4-node graph input->mix(2)->pitch->FFT-spectrum @48 kHz",
        "value": value, "unit": "sample-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "settle_seconds": a.settle_seconds,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C5 (BASELINE.json configs[4]): independent 48 kHz stereo f32 streams through "
                               f"input->mix(2)->pitch({a.semitones:+g} semitones, rate {a.rate:g}, phase vocoder N=1024 hop=256)"
                               "->spectrum(N=1024 hop=256)",
                   "job_streams": job_streams, "streams_on_rank0": n_streams, "seconds_per_stream": a.seconds, "sample_frames_per_stream": S,
                   "shared_second_input": True, "parallelism": f"streams sharded over {world} GPU(s), no data-path collective",
                   "device": ctx.name()},
        "roofline": roofline, "clock_GHz": round(clock_ghz, 3), "sustained": sustained,
        "chain": {"alg_bytes_per_sample_frame": 64.03, "alg_GBps_per_gpu": round(chain_gbs, 1),
                  "frac_hbm_peak": round(chain_gbs / HBM_PEAK_GBS, 4)},
        "kernels": kern_report,
        "memory_side": memory_side,
    }
    if ranks is not None:
        # (src/infra/runner.cpp:35-50, 142-154: one product per link, every branch runs by itself — the ranks share only the broadcast)
        out["ranks"] = {"ranks_seen": ranks["ranks_seen"], "backend": "nccl (RCCL)", "launched_world_size": world,
                        "collective": "one broadcast of the shared second mix input from rank 0 at setup; the timed region has none",
                        "per_rank": ranks["per_rank"],
                        "slowest_rank": max(ranks["per_rank"], key=lambda d: d["ms_per_step"])["rank"],
                        "streams_covered": sum(d["streams"] for d in ranks["per_rank"])}

    # ---- the same pitch node with the SoundTouch-shaped WSOLA chain (K7 option A) instead of the phase vocoder:
    # reported beside the headline, never part of `value`
    if not a.no_alt and world == 1:
        wpl = ctx.wsola_plan(48000, 2, a.rate, pitch, S)
        wdst = nae.Sig.interleaved(d_pitch.ptr, wpl.out_len, 2)
        assert wpl.out_len <= pl.out_len
        ctx.wsola_block(48000, a.rate, pitch, g.mix_out, S, 2, n_streams, wdst)     # plan, table, workspaces
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        tw0 = time.perf_counter()
        for _ in range(2):
            ctx.wsola_block(48000, a.rate, pitch, g.mix_out, S, 2, n_streams, wdst)
        ctx.sync()
        w_ms = (time.perf_counter() - tw0) / 2 * 1e3
        ctx.prof_enable(False)
        w_clock = ctx.clock_ghz()                            # probe kernel directly behind the WSOLA launches (not the vocoder step's clock)
        wk = {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.prof_report().items()}
        other = sum(v["avg_ms"] for k, v in kern_report.items() if not k.startswith(("pv_", "resample")))
        out["pitch_node_soundtouch_algorithm"] = {
            "what": "input->mix(2)->pitch->spectrum with the pitch node running the WSOLA + anti-alias FIR + cubic "
                    "transposer chain restated from SoundTouch 2.3.2 (nae_wsola_block_f32) instead of the phase vocoder "
                    "(st_aa_kernel at rate > 1 is the instantiation with the cubic stage fused behind the filter)",
            "pitch_node_ms": round(w_ms, 3), "kernels_avg_ms": wk, "sequences_per_stream": int(wpl.n_seq),
            "graph_sample_frames_per_s": n_streams * S / ((w_ms + other) * 1e-3)}
        # what binds the two kernels: vector issue at the measured interval + LDS-array cycles against the kernel's cycles, from the
        # rocprofv3 --pmc passes of this build with the WSOLA leg in the run (ALT=1 tools/pmc_sq.sh -> profiles/r05_wsola_sq.md)
        wpath = os.path.join(ROOT, "profiles", "r05_wsola_traffic.json")
        wdata = json.load(open(wpath)) if os.path.exists(wpath) else {}
        valu = {}
        for kname, avg_ms in wk.items():
            td = wdata.get(kname, {})
            if not td.get("valu_instr_per_launch") or not td.get("sample_frames"):
                continue
            scale = sf / td["sample_frames"]
            instr = td["valu_instr_per_launch"] * scale
            cyc = avg_ms * 1e-3 * w_clock * 1e9
            peak_cpi = valu_cycles_per_instr(td.get("waves_per_simd", 4))
            v_cyc, l_cyc = instr / 1024.0 * peak_cpi, td.get("lds_idx_active_per_cu", 0.0) * scale
            valu[kname] = {"wave_instr_per_launch": instr, "waves_per_simd": td.get("waves_per_simd", 4),
                           "achieved": round(cyc * 1024 / instr, 3), "peak": peak_cpi, "unit": "cycles per wave-instruction per SIMD (lower is better)",
                           "frac": round(peak_cpi * instr / 1024 / cyc, 3),
                           "valu_busy_cycles_per_simd": v_cyc, "lds_cycles_per_cu": l_cyc, "kernel_cycles": cyc,
                           "floor_ms_at_this_clock": round(v_cyc / (w_clock * 1e9) * 1e3, 3)}
        if valu:
            out["pitch_node_soundtouch_algorithm"]["clock_GHz"] = round(w_clock, 3)
            out["pitch_node_soundtouch_algorithm"]["valu"] = valu
            out["pitch_node_soundtouch_algorithm"]["valu_source"] = wdata.get("_source")
        ctx.graph4(g)                                        # restore the vocoder result for the parity check below
        ctx.sync()

    # ---- the same graph with HOST buffers on both sides of the boundary (pinned staging, upload / compute / download pipelined): what an
    # editor process behind the plugin API would see.  Reported beside the headline, never part of `value`.
    if not a.no_pcie and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_pcie
        # (round 5: one upload queue, one download queue and four compute lanes tied by events; 32-stream chunks; 1024 streams so that
        # the figure is the sustained one — rounds 2-4's three independent lanes hold 1.8e9 over a run this long)
        out["pcie_inclusive"] = bench_pcie.measure(nae, device=local_rank, streams=1024, chunk=32, lanes=4, S=S, semitones=a.semitones, mode="queues")

    if not a.no_host_path and world == 1:
        ctx.sync()
        out["host_path"] = host_path()

    # ---- CPU baseline (reported, not the target): the oracle's restatement of the same graph, 1 thread
    if not a.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc                                          # oracle: allowed here only as the timed checker
        k = max(1, min(a.cpu_streams, n_streams))
        b = orc.fill_uniform(S * 2, orc.stream_seed(0, 1))
        ins = [orc.fill_uniform(S * 2, orc.stream_seed(s, 0)) for s in range(k)]
        tc0 = time.perf_counter()
        refs = []
        for s in range(k):
            x = ins[s]
            L, R = orc.amix([x[0::2], b[0::2]], [x[1::2], b[1::2]], [0.5, 0.5])
            p_out = orc.stretch(orc.interleave([L, R]), 2, a.rate, pitch)
            sp = orc.spectrum(p_out, 2)
            if s == 0:
                refs = [p_out, sp]
        tc1 = time.perf_counter()
        out["cpu_baseline"] = {"value": k * S / (tc1 - tc0), "unit": "sample-frames/s", "cores": 1, "kind": "port",
                               "sample": f"{k} of the {n_streams} streams ({a.seconds:g} s each), same 4-node graph, "
                                         f"oracle C restatement (-O3, no fast-math), 1 thread as the reference's runner "
                                         f"is single-threaded (src/infra/runner.cpp:65-69); host has {os.cpu_count()} cpus",
                               "seconds": round(tc1 - tc0, 2)}
        # the same restatement on the box's CPU share, one stream per worker thread (a fair upper bound
        # for a CPU implementation; the reference itself runs every node on ONE thread).  ctypes drops the GIL in C.
        from concurrent.futures import ThreadPoolExecutor
        workers, cpu_info = usable_cores()                 # every core this process may use; `cores` says how many

        def one(x):
            L, R = orc.amix([x[0::2], b[0::2]], [x[1::2], b[1::2]], [0.5, 0.5])
            orc.spectrum(orc.stretch(orc.interleave([L, R]), 2, a.rate, pitch), 2)

        reps = max(1, (4 * workers + k - 1) // k)           # at least four stream-runs per thread
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, (ins * reps)[:workers]))     # untimed: the first concurrent pass runs serially (arena set-up)
            tm0 = time.perf_counter()
            list(ex.map(one, ins * reps))
            tm1 = time.perf_counter()
        out["cpu_baseline"]["all_workers"] = {"value": k * reps * S / (tm1 - tm0), "unit": "sample-frames/s", "cores": workers,
                                              "sample": f"{k * reps} stream-runs over {workers} threads, one stream per thread at a time", "seconds": round(tm1 - tm0, 2), **cpu_info}
        # parity of this very run: stream 0 of the GPU result against the oracle
        gp = np.empty(pl.out_len * 2, np.float32)
        gs = np.empty(F * 2 * BINS, np.float32)
        ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, gp.ctypes.data, d_pitch.ptr, gp.nbytes))
        ctx._ck(ctx.lib.nae_memcpy_d2h(ctx.h, gs.ctypes.data, d_spec.ptr, gs.nbytes))
        ctx.sync()
        rr = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)) / np.sqrt(np.mean(y.astype(np.float64) ** 2)))
        out["rms_err_vs_oracle"] = {"pitch_out": rr(gp, refs[0]), "spectrum_out": rr(gs, refs[1].reshape(-1)), "tolerance": 1e-4}
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out))
    sys.stdout.flush()
    os.dup2(2, 1)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
