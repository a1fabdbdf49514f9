"""The multi-GPU arrangement (independent streams, one process per GPU,
max-over-ranks timing) exercised with two gloo processes on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visual_odometry_amd import shard


def test_streams_round_robin():
    assert shard.streams_of_rank(8, 1, 0) == list(range(8))
    assert shard.streams_of_rank(8, 4, 1) == [1, 5]
    assert shard.streams_of_rank(3, 8, 5) == []
    all_ = sorted(s for r in range(8) for s in shard.streams_of_rank(20, 8, r))
    assert all_ == list(range(20))
    with pytest.raises(ValueError):
        shard.streams_of_rank(4, 2, 2)


def test_single_process_identity():
    assert shard.max_over_ranks(1.25) == 1.25
    assert shard.aggregate_fps(100, 2.0) == 50.0


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    frames = 100 * (rank + 1)
    fps = shard.aggregate_fps(frames, elapsed, dist, "cpu")
    mx = shard.max_over_ranks(elapsed, dist, "cpu")
    dist.barrier()
    q.put((rank, fps, mx, shard.streams_of_rank(5, world, rank)))
    dist.destroy_process_group()


def test_two_rank_aggregation_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, fps, mx, streams in out:
        assert mx == 2.0 and fps == pytest.approx(300 / 2.0)
    assert out[0][3] == [0, 2, 4] and out[1][3] == [1, 3]


def _placement_worker(rank, world, port, q):
    """The kitti8 placement of bench.py (eight streams dealt round-robin) with gloo ranks, no GPU calls."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.stream_seeds(8, world, rank, base_seed=4000)
    owned = shard.gather_stream_ids([s for s, _ in mine], dist)
    frames_per_stream, steps = 16, 3
    frames = len(mine) * frames_per_stream * steps
    elapsed = 0.5 + 0.1 * rank             # the last rank is the slowest
    fps = shard.aggregate_fps(frames, elapsed, dist, "cpu")
    dist.barrier()
    q.put((rank, mine, owned, fps))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_kitti8_placement_gloo(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_placement_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = []
    for rank, mine, owned, fps in out:
        assert len(mine) == 8 // world                              # 8 / N streams per rank
        assert all(seed == 4000 + sid for sid, seed in mine)        # a stream's frames do not depend on N
        assert sorted(owned) == list(range(8))                      # every rank sees: each stream owned exactly once
        seen += [sid for sid, _ in mine]
        # whole-job rate: all frames of all ranks over the slowest rank's time
        assert fps == pytest.approx(8 * 16 * 3 / (0.5 + 0.1 * (world - 1)))
    assert sorted(seen) == list(range(8))
