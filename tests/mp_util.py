"""Multi-process test harness: spawn `world` ranks, collect one result per rank, never hang.

Every worker always reports -- its result or its traceback -- and the parent polls with a deadline, joins with a
timeout, checks exit codes and terminates survivors, so a rank that dies before its first collective becomes a test
failure with a traceback instead of a partner stuck in the collective and a parent stuck in ``get()``."""
import datetime
import os
import queue
import socket
import time
import traceback

import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(rank, world, port, q, fn, args, backend):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        kw = {}
        if backend == "nccl":               # RCCL: one device per rank, bound before the first collective
            import torch
            torch.cuda.set_device(rank)
            kw["device_id"] = torch.device("cuda", rank)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120), **kw)
        try:
            q.put(("ok", rank, fn(rank, world, *args)))
        finally:
            dist.destroy_process_group()
    except BaseException:  # noqa: BLE001 -- the parent must hear about every failure
        q.put(("err", rank, traceback.format_exc()))


def run_ranks(fn, world, args=(), timeout=300.0, backend="gloo"):
    """Runs ``fn(rank, world, *args)`` on `world` spawned processes (module-level fn); returns results ordered by rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, q, fn, args, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got, deadline = {}, time.time() + timeout
    try:
        while len(got) < world and time.time() < deadline:
            try:
                status, rank, payload = q.get(timeout=1.0)
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
                continue
            got[rank] = (status, payload)
            if status == "err":
                break
    finally:
        for p in procs:
            p.join(10 if len(got) == world else 1)
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(5)
    errors = [f"rank {r}:\n{pl}" for r, (st, pl) in sorted(got.items()) if st == "err"]
    assert not errors, "\n".join(errors)
    assert len(got) == world, f"only ranks {sorted(got)} of {world} reported (exit codes {[p.exitcode for p in procs]})"
    return [got[r][1] for r in range(world)]
