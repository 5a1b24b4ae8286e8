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
