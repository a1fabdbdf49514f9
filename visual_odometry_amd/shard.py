"""Multi-GPU arrangement of the front end: independent camera streams, one
process per GPU, no collective on the data path.

A single stream is sequential in time (frame t is matched against frame t-1,
reference src/frame_processor.cpp:36-38, include/visual_odometry/frame_processor.hpp:55-58)
and far too small to split, so streams -- not frames -- are what gets sharded:
stream s runs on rank s mod world.  The only cross-rank traffic is the
benchmark's barrier and the max-reduction of the elapsed time.
"""
from __future__ import annotations


def streams_of_rank(n_streams: int, world: int, rank: int) -> list[int]:
    """Global stream ids owned by `rank` (round-robin)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_streams, world))


def stream_seeds(n_streams: int, world: int, rank: int, base_seed: int) -> list[tuple[int, int]]:
    """(global stream id, scene seed) of every stream `rank` owns.  A stream keeps its seed -- hence its frames --
    whatever the number of ranks, so runs on 1, 2, 4 and 8 GPUs process the same job (strong scaling)."""
    return [(s, base_seed + s) for s in streams_of_rank(n_streams, world, rank)]


def gather_stream_ids(ids: list[int], dist=None) -> list[int]:
    """All ranks' stream ids, concatenated in rank order (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(ids)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(ids))
    return [s for part in out for s in part]


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of a scalar (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_fps(frames_this_rank: int, elapsed_this_rank: float, dist=None, device=None) -> float:
    """Whole-job frames/s: all ranks' frames over the slowest rank's time."""
    frames = sum_over_ranks(float(frames_this_rank), dist, device)
    elapsed = max_over_ranks(elapsed_this_rank, dist, device)
    return frames / elapsed


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    """Environment of rank `rank` of a one-node job: what torch.distributed.run would export."""
    import os
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return env


def launch_ranks(world: int, child_argv: list[str], out=None, poll_s: float = 0.2) -> int:
    """Start `world` fresh child processes of `child_argv`, one per GPU (one sequential caller per stream and
    process, reference src/vo_node.cpp:940-962), with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, and wait for
    them.  Rank 0's standard output is relayed to `out` (default: this process' stdout) once all ranks have ended
    well; the other ranks' standard output goes to stderr, so a job prints exactly what its rank 0 prints.  When a
    rank ends with a non-zero status the remaining ranks are terminated (by their pid) and that status is returned.
    The caller must not have initialised the GPU: the children are new processes, nothing is re-executed."""
    import subprocess
    import sys
    import time
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(child_argv, env=rank_env(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    # rank 0's lines are read on a thread so that a long line never blocks the child on a full pipe
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    status = 0
    live = set(range(world))
    while live and status == 0:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                status = rc if rc > 0 else 128 - rc
                break
        if live and status == 0:
            time.sleep(poll_s)
    if status != 0:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + 10.0
        for r in live:
            try:
                procs[r].wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    reader.join(10.0)
    text = (captured[0] if captured else b"").decode(errors="replace")
    if status == 0:
        (out or sys.stdout).write(text)
        (out or sys.stdout).flush()
    else:
        sys.stderr.write(text)
    return status
