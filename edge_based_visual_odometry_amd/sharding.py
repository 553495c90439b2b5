"""Multi-GPU sharding of the hot path: independent sequences, one per rank, no data-path collective.

Every stereo pair's TOED + stereo matching is independent of every other pair, and temporal matching
couples a frame only to keyframe 0 of its own sequence (src/Pipeline.cpp:133-138), so a node runs one
process per GPU, each owning a whole sequence.  torch.distributed (RCCL on GPUs, gloo in the CPU tests)
is used only to line the ranks up and to take the MAX of the timed region.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(frozen=True)
class RankInfo:
    rank: int
    local_rank: int
    world: int


def rank_info() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                    int(os.environ.get("WORLD_SIZE", "1")))


def rank_workload(rank: int) -> dict:
    """Sequence owned by `rank`: its own scene and noise seeds."""
    return dict(scene=7 + rank, noise_base=100 * rank, disparity=12)


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """MAX over ranks of the timed region (the job is as slow as its slowest sequence)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(world: int, steps_per_rank: int, seconds: float) -> float:
    """Whole-job pairs/s: every rank processed `steps_per_rank` pairs (weak scaling)."""
    return world * steps_per_rank / seconds
