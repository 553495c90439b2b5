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


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(world: int, argv: list, env: dict | None = None, timeout: float | None = None) -> int:
    """Starts `world` fresh processes `argv`, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in
    their environment, the rendezvous on 127.0.0.1), waits for all of them and returns 0 only if every rank returned 0;
    the first failure ends the others.  Must be called BEFORE the calling process has touched the GPU: the ranks are
    children (fork + exec of a process that holds no HIP state), the caller is never replaced.  The ranks inherit stdout /
    stderr, so rank 0's one JSON line is the job's."""
    import subprocess
    import time
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world),
                LOCAL_WORLD_SIZE=str(world))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e))
    t_end = None if timeout is None else time.monotonic() + timeout
    rc = 0
    alive = set(range(world))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        if alive and (rc != 0 or (t_end is not None and time.monotonic() > t_end)):
            for r in alive:                      # exactly the processes started above, by pid
                procs[r].terminate()
            for r in alive:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            rc = rc or 124
            break
        if alive:
            time.sleep(0.02)
    return rc


def gather_over_ranks(value: float, dist=None, device=None) -> list:
    """Every rank's `value`, in rank order (rank 0 reports the per-rank throughputs)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
