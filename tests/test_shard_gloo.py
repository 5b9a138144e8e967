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
