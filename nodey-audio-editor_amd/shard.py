"""Multi-GPU sharding of independent streams.  Two job shapes:

* strong scaling (BASELINE.json configs[4], the default of bench.py): the job is `total` streams, rank r owns the
  contiguous slice [r*total//world, (r+1)*total//world) — sizes differ by at most one stream;
* weak scaling (`bench.py --streams N`): every rank owns N streams, stream s of the job lives on rank s // N.

Either way there is no data-path collective: streams share nothing (/root/reference/src/infra/runner.cpp:35-50, one
product per link).  The only exchange is the one-shot broadcast of the shared source buffer from rank 0 at setup
(RCCL over xGMI on GPUs, gloo in the CPU tests).  torch.distributed is plumbing only; nothing here computes audio."""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def stream_range(rank: int, per_rank: int) -> Tuple[int, int]:
    """global stream ids [first, last) owned by `rank`"""
    return rank * per_rank, (rank + 1) * per_rank


def strong_range(rank: int, world: int, total: int) -> Tuple[int, int]:
    """global stream ids [first, last) of `rank` when a job of `total` streams is cut into `world` contiguous slices"""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request rank={rank} world={world} total={total}")
    return rank * total // world, (rank + 1) * total // world


def stream_seed(stream: int, input_index: int = 0) -> int:
    """seed = 0x9E3779B97F4A7C15 * (1 + s) + k  (mod 2^64)"""
    return (0x9E3779B97F4A7C15 * (1 + stream) + input_index) & 0xFFFFFFFFFFFFFFFF


def init(backend: str, rank: int, world: int):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def broadcast_shared(dist, tensor, src: int = 0):
    """the single collective of the path: every rank leaves with rank `src`'s copy of the shared source buffer"""
    dist.broadcast(tensor, src=src)
    return tensor


def max_over_ranks(dist, seconds: float, device=None) -> float:
    """elapsed time of the job = slowest rank (bench.py contract)"""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(world: int, per_rank_streams: int, frames_per_stream: int, steps: int, elapsed_s: float) -> float:
    """weak mode, whole-job sample-frames/s: all ranks' units over the slowest rank's time"""
    return world * per_rank_streams * frames_per_stream * steps / elapsed_s


def job_throughput_total(total_streams: int, frames_per_stream: int, steps: int, elapsed_s: float) -> float:
    """strong mode: the job's `total_streams` (however they were cut) over the slowest rank's time"""
    return total_streams * frames_per_stream * steps / elapsed_s


RANK_REPORT_FIELDS = ("rank", "first_stream", "last_stream", "streams", "ms_per_step", "clock_GHz", "broadcast_ms", "broadcast_bytes")


def timed_broadcast_shared(dist, tensor, src: int = 0, sync=None):
    """broadcast_shared with this rank's wall time around it (ms) and the bytes moved; `sync` (e.g. torch.cuda.synchronize) is
    called before both clock readings so that the time covers the transfer, not its enqueueing"""
    import time
    if sync is not None:
        sync()
    t0 = time.perf_counter()
    dist.broadcast(tensor, src=src)
    if sync is not None:
        sync()
    return (time.perf_counter() - t0) * 1e3, tensor.numel() * tensor.element_size()


def gather_rank_reports(dist, report: dict, device=None) -> dict:
    """What every rank of the job saw, gathered on every rank (one all_gather of eight doubles per rank): the N > 1 bench line
    carries it so that the line says by itself how many ranks the collective library connected (`ranks_seen` = the process
    group's world size AFTER the broadcast went through it), which streams each rank owned, and each rank's own step time and
    clock.  `report` holds RANK_REPORT_FIELDS; the result is {"ranks_seen": n, "per_rank": [dict per rank, in rank order]}."""
    import torch
    mine = torch.tensor([float(report[k]) for k in RANK_REPORT_FIELDS], dtype=torch.float64, device=device)
    world = dist.get_world_size()
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    per_rank = []
    for row in rows:
        vals = row.cpu().tolist()
        d = dict(zip(RANK_REPORT_FIELDS, vals))
        for k in ("rank", "first_stream", "last_stream", "streams", "broadcast_bytes"):
            d[k] = int(d[k])
        per_rank.append(d)
    per_rank.sort(key=lambda d: d["rank"])
    return {"ranks_seen": world, "per_rank": per_rank}
