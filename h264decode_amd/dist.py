"""Multi-GPU helpers: stream -> rank partitioning and the closing statistics all-reduce.

Streams are independent, so the decode path has NO collective: every rank decodes
its own shard on its own GPU.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node,
"gloo" in CPU tests) is used only for the barrier around the timed region and for one small
all-reduce of per-rank statistics."""
from typing import Dict, List, Sequence


def shard_streams(n_streams: int, world_size: int, rank: int) -> List[int]:
    """Static round-robin `s mod world_size == rank` (equal-cost streams)."""
    return [s for s in range(n_streams) if s % world_size == rank]


def shard_streams_lpt(costs: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Longest-processing-time-first greedy on per-stream cost (e.g. total slice bytes: entropy
    decoding time is proportional to bits).  Deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world_size
    owner = {}
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return sorted(i for i, r in owner.items() if r == rank)


def allreduce_stats(stats: Dict[str, float], device=None) -> Dict[str, float]:
    """Sum `frames`, `pixels`, `bytes_in`, `ranks` (1 per rank: how many ranks the collective saw); max `seconds`; xor-fold
    `checksum` across ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(stats)
    keys_sum = [k for k in ("frames", "pixels", "bytes_in", "ranks") if k in stats]
    t = torch.tensor([float(stats[k]) for k in keys_sum], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = dict(zip(keys_sum, t.tolist()))
    if "seconds" in stats:
        m = torch.tensor([float(stats["seconds"])], dtype=torch.float64, device=device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        out["seconds"] = float(m.item())
    if "checksum" in stats:
        world = dist.get_world_size()
        c = torch.zeros(world, dtype=torch.int64, device=device)
        c[dist.get_rank()] = int(stats["checksum"]) & 0x7FFFFFFFFFFFFFFF
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        x = 0
        for v in c.tolist():
            x ^= int(v)
        out["checksum"] = x
    return out
