"""Multi-GPU driver pieces: the per-volume fit is embarrassingly parallel (superresDWI.py:29 patient
loop, superresHybrid.py:79 TE loop, master.py:64-95 seed/case/direction loops), so fits are
partitioned over one-process-per-GPU ranks with NO data-path collective.  The only exchange is the
final gather of fixed-size metric records (RCCL ``all_gather`` over xGMI on GPUs; ``gloo`` in the CPU
tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def partition_fits(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of fit jobs to ranks.

    ``costs[i]`` is the work of job i (coordinate count x steps).  Returns ``world_size`` lists of job
    indices; deterministic (ties broken by job index, then by rank), so every rank computes the same
    schedule without communicating.
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    plan: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        plan[r].append(i)
        loads[r] += float(costs[i])
    return plan


def _split_ranks(world_size: int, weights: Sequence[float]) -> List[List[int]]:
    """Partition ranks 0..world_size-1 into len(weights) contiguous groups sized in proportion to the weights
    (largest-remainder rounding, every group at least one rank)."""
    n = len(weights)
    total = float(sum(weights))
    sizes = [1] * n
    spare = world_size - n
    shares = [(float(w) / total * world_size - 1.0) if total > 0 else 0.0 for w in weights]
    base = [max(0, int(x)) for x in shares]
    while sum(base) > spare:                      # rounding may overshoot when many groups sit at the minimum
        base[max(range(n), key=lambda i: (base[i], -i))] -= 1
    for i in range(n):
        sizes[i] += base[i]
    left = world_size - sum(sizes)
    for i in sorted(range(n), key=lambda i: (-(shares[i] - base[i]), i))[:left]:
        sizes[i] += 1
    groups, r = [], 0
    for k in sizes:
        groups.append(list(range(r, r + k)))
        r += k
    return groups


def plan_fits(costs: Sequence[float], world_size: int, shard_overhead: float = 0.03) -> Dict[str, object]:
    """Schedule with row-sharded fits (SURVEY.md 8 e): whole-volume packing alone tops out at 6.04x on the reference's
    11 patients and 8 GPUs, because three ranks get two volumes.  Here the jobs that do not fill a whole round are run
    FIRST as a gang phase -- every such job split over its own group of ranks (``ShardedSirenFitter``: one gradient
    all-reduce per step inside the group), all groups starting together at t = 0, so no rank ever waits for a
    partner -- and the remaining jobs are packed whole, LPT, on top of the group finish times.

    Candidates: plain LPT; a gang phase with the r largest jobs, or with the r smallest, for every r up to min(n, world).
    The cheapest by the cost model wins (sharded time = cost / k * (1 + shard_overhead)).  Deterministic.
    Returns {"gangs": [(job, [ranks])...], "whole": [[job...] per rank], "makespan": modelled time}."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    n = len(costs)
    c = [float(x) for x in costs]

    def finish(gang_jobs):
        gangs, start = [], [0.0] * world_size
        if gang_jobs:
            groups = _split_ranks(world_size, [c[j] for j in gang_jobs])
            for j, ranks in zip(gang_jobs, groups):
                t = c[j] / len(ranks) * (1.0 + (shard_overhead if len(ranks) > 1 else 0.0))
                for r in ranks:
                    start[r] = t
                gangs.append((j, ranks))
        rest = sorted((i for i in range(n) if i not in set(gang_jobs)), key=lambda i: (-c[i], i))
        loads, whole = list(start), [[] for _ in range(world_size)]
        for i in rest:
            r = min(range(world_size), key=lambda k: (loads[k], k))
            whole[r].append(i)
            loads[r] += c[i]
        return {"gangs": gangs, "whole": whole, "makespan": max(loads, default=0.0)}

    by_cost = sorted(range(n), key=lambda i: (-c[i], i))
    candidates = [finish([])]
    if world_size > 1:
        for r in range(1, min(n, world_size) + 1):          # gang phase with the r largest / the r smallest jobs
            candidates.append(finish(by_cost[:r]))
            if r < n:
                candidates.append(finish(sorted(by_cost[-r:], key=lambda i: (-c[i], i))))
    return min(candidates, key=lambda p: p["makespan"])     # ties: the earlier (simpler) candidate


def makespan(costs: Sequence[float], plan: Sequence[Sequence[int]]) -> float:
    return max((sum(float(costs[i]) for i in jobs) for jobs in plan), default=0.0)


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_records(record: Dict[str, float]) -> List[Dict[str, float]]:
    """All ranks contribute one record of scalars with identical keys; every rank gets the list ordered by
    rank.  Implemented as ONE fixed-size tensor ``all_gather`` (float64 x len(record)): on the ``nccl``
    backend this is the RCCL collective, on ``gloo`` a CPU tensor is used."""
    keys = sorted(record)
    if _world() == 1:
        return [dict(record)]
    backend = dist.get_backend()
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(record[k]) for k in keys], dtype=torch.float64, device=device)
    everyone = torch.empty(_world() * len(keys), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(everyone, mine)
    rows = everyone.cpu().view(_world(), len(keys)).tolist()
    return [dict(zip(keys, row)) for row in rows]


def gather_job_records(local: List[Dict[str, float]], keys: Sequence[str], max_jobs_per_rank: int) -> List[Dict[str, float]]:
    """Gather a variable number (<= ``max_jobs_per_rank``) of per-fit records from every rank with one
    fixed-size ``all_gather`` (unused slots are NaN-padded and dropped)."""
    keys = list(keys)
    if len(local) > max_jobs_per_rank:
        raise ValueError("more local records than max_jobs_per_rank")
    width = len(keys)
    buf = [float("nan")] * (max_jobs_per_rank * width)
    for j, rec in enumerate(local):
        for c, k in enumerate(keys):
            buf[j * width + c] = float(rec[k])
    if _world() == 1:
        rows = [buf]
    else:
        backend = dist.get_backend()
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        mine = torch.tensor(buf, dtype=torch.float64, device=device)
        everyone = torch.empty(_world() * len(buf), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(everyone, mine)
        rows = everyone.cpu().view(_world(), len(buf)).tolist()
    out = []
    for row in rows:
        for j in range(max_jobs_per_rank):
            vals = row[j * width:(j + 1) * width]
            if vals[0] == vals[0]:  # not NaN
                out.append(dict(zip(keys, vals)))
    return out
