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
