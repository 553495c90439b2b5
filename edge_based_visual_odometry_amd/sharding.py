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


# ---- host side of a rank: which cores it runs on ------------------------------------------------------------------------
# One sequence per GPU means one host thread per GPU that must never wait for a core (BASELINE configs[4]): every rank is
# pinned to its own cores -- those of its GPU's NUMA node when sysfs tells which node that is -- disjoint from every other
# rank's, and its thread pools (OpenMP / torch) are capped to that set.

def _parse_cpulist(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_node(device: int, sysfs: str = "/sys"):
    """NUMA node of HIP device `device` (render node 128 + device), or None when sysfs does not say (a container, one node)."""
    for path in (f"{sysfs}/class/drm/renderD{128 + device}/device/numa_node", f"{sysfs}/class/drm/card{device}/device/numa_node"):
        try:
            n = int(open(path).read().strip())
            if n >= 0:
                return n
        except (OSError, ValueError):
            continue
    return None


def numa_cpus(node: int, sysfs: str = "/sys"):
    try:
        return _parse_cpulist(open(f"{sysfs}/devices/system/node/node{node}/cpulist").read())
    except (OSError, ValueError):
        return None


def rank_cpu_sets(local_world: int, n_devices: int, available=None, sysfs: str = "/sys") -> list:
    """The core set of every local rank, computed identically by every rank (no communication): rank r drives device
    r % n_devices; ranks whose GPUs sit on the same NUMA node share that node's available cores in contiguous, disjoint
    chunks; without NUMA information all ranks share all available cores the same way.  An entry is None when there are
    fewer cores than ranks in a pool (nothing sensible to pin to)."""
    avail = sorted(os.sched_getaffinity(0) if available is None else available)
    pools = {}
    for r in range(local_world):
        node = gpu_numa_node(r % max(1, n_devices), sysfs)
        cpus = None if node is None else numa_cpus(node, sysfs)
        key = node if cpus and set(cpus) & set(avail) else None
        pools.setdefault(key, []).append(r)
    if None in pools and len(pools) > 1:             # mixed information: treat every rank alike
        pools = {None: list(range(local_world))}
    sets = [None] * local_world
    for key, ranks in pools.items():
        pool = avail if key is None else [c for c in numa_cpus(key, sysfs) if c in set(avail)]
        per = len(pool) // len(ranks)
        if per < 1:
            continue
        for k, r in enumerate(ranks):
            sets[r] = pool[k * per:(k + 1) * per]
    return sets


def pin_rank(local_rank: int, local_world: int, n_devices: int, max_threads: int = 16):
    """Pins the calling process to its share of the cores and caps its thread pools; returns the core list (or None when
    not pinned).  Call before the first thread pool is created (before importing torch is best; after works for affinity)."""
    cpus = rank_cpu_sets(local_world, n_devices)[local_rank]
    if not cpus:
        return None
    os.sched_setaffinity(0, cpus)
    n = str(max(1, min(len(cpus), max_threads)))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = n
    os.environ.setdefault("EBVO_CPU_THREADS", n)
    return cpus


def gather_int_lists(values: list, width: int, dist=None, device=None) -> list:
    """Every rank's list of ints (padded with -1 to `width`), in rank order."""
    vals = (list(values) + [-1] * width)[:width]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [[v for v in vals if v >= 0]]
    import torch
    t = torch.tensor(vals, dtype=torch.int64, device=device or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(v) for v in x.tolist() if v >= 0] for x in out]


def rank_balance(per_rank: list, world: int, value: float) -> dict:
    """min / max of the per-rank rates and the job's efficiency against N times its best rank."""
    best, worst = max(per_rank), min(per_rank)
    return {"per_rank_min": worst, "per_rank_max": best, "per_rank_min_over_max": worst / best if best else None,
            "efficiency_vs_best_rank": value / (world * best) if best else None}
