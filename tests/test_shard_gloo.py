"""CPU, world_size 2 and 4, gloo: the N>1 path of bench.py — strong-scaling stream sharding (a fixed job cut into contiguous
slices, BASELINE.json configs[4]), the one-shot broadcast of the shared source buffer, and the max-over-ranks timing —
exercised without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import naeload
    import orc
    nae = naeload.load()
    from nodey_audio_editor_amd import shard
    dist = shard.init("gloo", rank, world)
    first, last = shard.strong_range(rank, world, total)
    # every rank generates ITS streams; rank 0 also generates the shared second input and broadcasts it
    mine = [orc.fill_uniform(n, shard.stream_seed(s, 0)) for s in range(first, last)]
    shared = torch.zeros(n, dtype=torch.float32)
    if rank == 0:
        shared = torch.from_numpy(orc.fill_uniform(n, shard.stream_seed(0, 1)))
    shard.broadcast_shared(dist, shared, 0)
    bms, bbytes = shard.timed_broadcast_shared(dist, shared, 0)          # the timed form bench.py uses (a second, idempotent broadcast)
    # the per-rank report of bench.py's N > 1 line, gathered the way bench.py gathers it
    ranks = shard.gather_rank_reports(dist, {"rank": rank, "first_stream": first, "last_stream": last, "streams": last - first,
                                             "ms_per_step": 10.0 * (rank + 1), "clock_GHz": 2.0 + 0.1 * rank, "broadcast_ms": bms,
                                             "broadcast_bytes": bbytes})
    elapsed = shard.max_over_ranks(dist, 0.010 * (rank + 1))
    dist.barrier()
    q.put((rank, first, last, [float(m[0]) for m in mine], shared.numpy().copy(), elapsed, ranks))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 5), (4, 10)])       # 5 streams over 2 ranks: slices of 2 and 3; 10 over 4: 2, 3, 2, 3
def test_rank_sharding_and_broadcast(world, total):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    n = 4096
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_shared = orc.fill_uniform(n, (0x9E3779B97F4A7C15 * 1 + 1) & 0xFFFFFFFFFFFFFFFF)
    import naeload
    naeload.load()
    from nodey_audio_editor_amd import shard as _shard
    shard_fields = _shard.RANK_REPORT_FIELDS
    covered = []
    for rank, first, last, firsts, shared, elapsed, ranks in res:
        # every rank holds the same report: both ranks seen, their slices, their own step times and clocks, the broadcast's size
        assert ranks["ranks_seen"] == world and [d["rank"] for d in ranks["per_rank"]] == list(range(world))
        cuts = [(r * total // world, (r + 1) * total // world) for r in range(world)]
        assert [(d["first_stream"], d["last_stream"], d["streams"]) for d in ranks["per_rank"]] == [(a, b, b - a) for a, b in cuts]
        assert [d["ms_per_step"] for d in ranks["per_rank"]] == [10.0 * (r + 1) for r in range(world)]
        assert [round(d["clock_GHz"], 6) for d in ranks["per_rank"]] == [round(2.0 + 0.1 * r, 6) for r in range(world)]
        assert all(d["broadcast_bytes"] == 4 * n and d["broadcast_ms"] >= 0.0 for d in ranks["per_rank"])
        assert set(ranks["per_rank"][0]) == set(shard_fields)
        assert (first, last) == (rank * total // world, (rank + 1) * total // world)
        covered += list(range(first, last))
        assert np.array_equal(shared, ref_shared)               # broadcast delivered rank 0's buffer
        assert abs(elapsed - 0.010 * world) < 1e-9              # max over ranks
        for i, s in enumerate(range(first, last)):
            assert firsts[i] == float(orc.fill_uniform(1, (0x9E3779B97F4A7C15 * (1 + s)) & 0xFFFFFFFFFFFFFFFF)[0])
    assert covered == list(range(total))                        # disjoint and complete


def test_job_throughput_is_whole_job(nae):
    from nodey_audio_editor_amd import shard
    assert shard.job_throughput(8, 1024, 480000, 5, 0.25) == 8 * 1024 * 480000 * 5 / 0.25
    assert shard.stream_seed(0, 1) == (0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF
    assert shard.stream_range(3, 128) == (384, 512)
    # strong scaling: 1024 streams over 8 ranks = 128 each; uneven jobs differ by at most one stream and cover the job
    assert [shard.strong_range(r, 8, 1024) for r in (0, 7)] == [(0, 128), (896, 1024)]
    for total, world in ((1024, 8), (1000, 8), (5, 2), (3, 4)):
        cuts = [shard.strong_range(r, world, total) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    assert shard.job_throughput_total(1024, 480000, 5, 0.25) == 1024 * 480000 * 5 / 0.25


def test_bench_self_launch_dry_run_world_2():
    """`python bench.py --gpus 2` with no launcher around it: the parent (which has touched neither torch nor HIP) starts
    torch.distributed.run as a child, two gloo ranks rendezvous on 127.0.0.1, shard 5 streams 2 + 3, broadcast, gather their reports, and
    the parent relays exactly ONE JSON line carrying the `ranks` block."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--total-streams", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["scaling"] == "strong"
    rk = d["ranks"]
    assert rk["ranks_seen"] == 2 and rk["launched_world_size"] == 2 and rk["streams_covered"] == 5
    assert [(p["rank"], p["first_stream"], p["last_stream"]) for p in rk["per_rank"]] == [(0, 0, 2), (1, 2, 5)]
    assert d["ms_per_step"] == pytest.approx(20.0)            # the slowest rank's (made-up) 10 ms x (rank + 1)


def test_bench_self_launch_reports_a_failing_child_plainly():
    """a rank that cannot run (here: --gpus 2 on a box without GPUs, or a world / --gpus mismatch) ends the parent non-zero with that rank's
    own message in the relayed stderr tail, and with no result line on stdout"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "launched as 1 rank(s) for --gpus 2" in r.stderr and r.stdout.strip() == ""


def test_bench_counts_the_cores_it_may_really_use(tmp_path, monkeypatch):
    """`cpu_baseline.all_workers` runs on every core the process may use: the affinity mask, cut by a cgroup CPU quota — on the
    GPU boxes of this pool the mask shows the host's 256 CPUs while the quota is 16"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cores, info = bench.usable_cores()
    assert 1 <= cores <= info["affinity"] <= (os.cpu_count() or 1)
    if info["cgroup_cpu_quota"] is not None:
        assert cores == max(1, min(info["affinity"], int(info["cgroup_cpu_quota"] + 0.5)))
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("200000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    monkeypatch.setattr("builtins.open", fake_open)
    assert bench.usable_cores()[0] == 2
