"""N > 1 path on CPU: world_size-2 gloo rendezvous, stream sharding, statistics all-reduce.
The decode itself needs a GPU; here each rank runs the ORACLE on its shard (test infrastructure) so
that the sharding + reduction logic is exercised end to end."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from h264decode_amd.dist import allreduce_stats, shard_streams, shard_streams_lpt


def test_shard_streams_partition():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += shard_streams(256, world, r)
        assert sorted(seen) == list(range(256))
        assert all(len(shard_streams(256, world, r)) == 256 // world for r in range(world))
    assert shard_streams(0, 2, 0) == []
    assert shard_streams(3, 8, 5) == []          # ragged: more ranks than streams


def test_lpt_balances():
    costs = [100, 1, 1, 1, 50, 49, 2, 3]
    shards = [shard_streams_lpt(costs, 2, r) for r in range(2)]
    assert sorted(shards[0] + shards[1]) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    import streamgen
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_streams = 5  # ragged on purpose
    mine = shard_streams(n_streams, world, rank)
    frames = pixels = nbytes = 0
    csum = 0
    for s in mine:
        stream, rec, _ = streamgen.encode(width=64, height=48, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=1000 + s)
        out, info = oracle.decode(stream, crop=True)
        assert np.array_equal(out, rec)
        frames += info.n_frames
        pixels += info.n_frames * info.width * info.height
        nbytes += len(stream)
        csum ^= int.from_bytes(hashlib.md5(out.tobytes()).digest()[:7], "big")
    dist.barrier()
    tot = allreduce_stats(dict(frames=frames, pixels=pixels, bytes_in=nbytes, seconds=0.1 * (rank + 1), checksum=csum))
    if rank == 0:
        q.put(tot)
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tot = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tot["frames"] == 15 and tot["pixels"] == 15 * 64 * 48 and abs(tot["seconds"] - 0.2) < 1e-9
    # the xor-folded checksum must equal the single-process value
    import oracle
    import streamgen
    want = 0
    for s in range(5):
        stream, _, _ = streamgen.encode(width=64, height=48, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=1000 + s)
        out, _ = oracle.decode(stream, crop=True)
        want ^= int.from_bytes(hashlib.md5(out.tobytes()).digest()[:7], "big")
    assert tot["checksum"] == want


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses) must start two ranks as a child
    process, relay rank 0's single JSON line and exit code, and the closing all-reduce must have seen both ranks.
    --dry-run: rendezvous + all-reduce only, so this runs without a GPU (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["H264MI_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["slowest_rank_seconds"] == 1.0
    # a launcher whose world size contradicts --gpus is an error, not a silent one-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 4" in (bad.stderr + bad.stdout)


@pytest.mark.gpu
def test_gpu_bench_two_ranks_rehearsal():
    """bench.py's multi-rank path (barriers, max-over-ranks time, summed frames, rank-0 JSON) with two ranks sharing the one GPU
    of the test box over gloo; on the 8-GPU node the same code runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, H264MI_BENCH_DEVICE="0", H264MI_BENCH_BACKEND="gloo")
    # no launcher: bench.py --gpus 2 starts its two ranks itself
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "8", "--frames", "4", "--width", "320", "--height", "240"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_step_per_gpu"] == 32 and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 32 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.02  # all ranks' frames over the slowest rank's time
