"""CPU tests of the drivers' failure handling (the reference's only recovery logic is the re-creation of a
collapsed network, INR_ERD.py:211-217): the health rule, the re-seed loop of ``fit_volume`` and -- world_size 2 on gloo -- the
marking / re-queueing of ``run_volumes`` when one fit goes NaN and one rank raises.  The fit itself is replaced by a stand-in
(``drivers._fit_volume_once`` needs the GPU); the -m gpu twin runs the real one (tests/test_gpu_drivers.py)."""
import numpy as np
import pytest
import torch

from mri_super_resolution_amd import drivers
from tests.mp_util import run_ranks


def test_fit_health_rule():
    assert drivers.fit_health(1e-4, torch.rand(4, 4)) == "ok"
    assert drivers.fit_health(float("nan"), torch.rand(4, 4)) == "nan"
    assert drivers.fit_health(float("inf"), None) == "nan"
    assert drivers.fit_health(1e-4, torch.zeros(4, 4)) == "collapsed"            # `if not model_output...max()`
    assert drivers.fit_health(1e-4, torch.tensor([0.1, float("nan")])) == "nan"
    assert drivers.fit_health(None, None) == "ok"


def _fake_once(volume, steps=2500, seed=0, group=None, _poison=False, **kw):
    vol = np.asarray(volume)
    bad = _poison or (vol.flat[1] == -1.0)                       # flat[1] == -1: this volume never fits
    recon = torch.zeros(2, 2) if vol.flat[1] == -2.0 else torch.ones(2, 2)       # flat[1] == -2: collapses, whatever the seed
    return {"n_coords": float(vol.size), "steps": steps, "t_fit": 0.01 * vol.size, "t_recon": 0.0,
            "final_loss": float("nan") if bad else 1e-5, "psnr_db": 30.0 + (seed or 0) * 1e-4, "ssim_mean": 0.9,
            "_recon_probe": recon, "seed_used": seed}


def test_fit_volume_reseeds_a_diverged_fit(monkeypatch):
    monkeypatch.setattr(drivers, "_fit_volume_once", _fake_once)
    vol = np.ones((4, 4), np.float32)
    ok = drivers.fit_volume(vol, steps=3, seed=5)
    assert (ok["status"], ok["reseeds"], ok["health"], ok["seed_used"]) == (drivers.FIT_OK, 0, "ok", 5)
    re = drivers.fit_volume(vol, steps=3, seed=5, _fault=lambda attempt: "nan" if attempt == 0 else None)
    assert (re["status"], re["reseeds"], re["health"], re["seed_used"]) == (drivers.FIT_RESEEDED, 1, "ok", 5 + 7919)
    never = vol.copy()
    never.flat[1] = -1.0
    bad = drivers.fit_volume(never, steps=3, seed=5, max_reseeds=2)
    assert (bad["status"], bad["reseeds"], bad["health"]) == (drivers.FIT_FAILED, 2, "nan")
    dead = vol.copy()
    dead.flat[1] = -2.0
    assert drivers.fit_volume(dead, steps=3, seed=0)["health"] == "collapsed"
    assert "_recon_probe" not in ok


def _worker(rank, world):
    from mri_super_resolution_amd import drivers as D
    D._fit_volume_once = _fake_once
    calls = []

    def fit(volume, steps, return_recon=False, **kw):
        job = int(np.asarray(volume).flat[0])
        calls.append(job)
        if rank == 1 and job in (0, 3):                        # "rank 1 is broken": whatever it is given from some point on raises
            raise RuntimeError(f"device lost while fitting volume {job}")
        fault = (lambda attempt: "nan" if attempt == 0 else None) if job == 2 else None
        return D.fit_volume(volume, steps=steps, _fault=fault, **kw)

    # five volumes; sizes make the LPT plan [[4, 1, 2? ...]] deterministic: job id in flat[0]
    vols = []
    for j, side in enumerate((6, 3, 4, 5, 7)):
        v = np.ones((side, side), np.float32)
        v.flat[0] = j
        vols.append(v)
    vols[1].flat[1] = -1.0                                     # volume 1 never fits (NaN after every re-seed)
    stats = {}
    recs = D.run_volumes(vols, steps=4, allow_sharding=False, stats=stats, fit_fn=fit, hidden_features=64)
    return recs, calls, stats["plan"]["whole"], stats.get("requeued")


def test_run_volumes_marks_failures_and_requeues_a_failed_ranks_fits_world_2_gloo():
    (recs0, calls0, plan0, rq0), (recs1, calls1, plan1, rq1) = run_ranks(_worker, 2, timeout=180)
    assert plan0 == plan1 and sorted(j for p in plan0 for j in p) == [0, 1, 2, 3, 4]
    assert [r["job"] for r in recs0] == [0.0, 1.0, 2.0, 3.0, 4.0]
    for a, b in zip(recs0, recs1):                             # the same records on every rank (NaN-tolerant comparison)
        assert a.keys() == b.keys() and all((a[k] == b[k]) or (a[k] != a[k] and b[k] != b[k]) for k in a)
    by = {int(r["job"]): r for r in recs0}
    lost = [j for j in plan0[1] if j in (0, 3)]                 # what rank 1 was given and raised on
    assert lost, "the plan must give rank 1 at least one of the jobs it fails on"
    assert rq0 == rq1 == sorted(lost)
    for j in lost:                                              # re-run on the survivor (rank 0), marked
        assert by[j]["status"] == drivers.FIT_OK and by[j]["requeued"] == 1.0 and by[j]["rank"] == 0.0 and j in calls0
    assert by[2]["status"] == drivers.FIT_RESEEDED and by[2]["reseeds"] == 1.0 and by[2]["final_loss"] == pytest.approx(1e-5)
    assert by[1]["status"] == drivers.FIT_FAILED and by[1]["final_loss"] != by[1]["final_loss"]      # kept and marked, not re-queued
    assert by[4]["status"] == drivers.FIT_OK and by[4]["requeued"] == 0.0


def test_concurrent_is_sequential_without_a_gpu():
    """`run_volumes(concurrent=k)` needs streams: on a host without a GPU the fits run one after the other, same records."""
    from mri_super_resolution_amd import drivers
    vols = [np.full((4, 4), float(k + 1), np.float32) for k in range(3)]

    def fake(volume, steps, return_recon=False, **kw):
        return {"n_coords": float(volume.size), "t_fit": 0.0, "t_recon": 0.0, "final_loss": float(volume[0, 0]), "status": drivers.FIT_OK}

    a = drivers.run_volumes(vols, steps=1, fit_fn=fake)
    b = drivers.run_volumes(vols, steps=1, fit_fn=fake, concurrent=3)
    assert [r["final_loss"] for r in a] == [r["final_loss"] for r in b] == [1.0, 2.0, 3.0]


def test_run_volumes_single_process_errors_are_not_swallowed():
    """with one process nobody can take a failed fit over -- the runtime error reaches the caller (after the other
    fits ran), a programming error propagates at once, and `errors="record"` is the explicit way to get NaN records instead."""
    vols = [np.full((4, 4), float(k + 1), np.float32) for k in range(3)]
    ran = []

    def fit(volume, steps, return_recon=False, **kw):
        ran.append(float(volume[0, 0]))
        if volume[0, 0] == 2.0:
            raise RuntimeError("device lost")
        return {"n_coords": float(volume.size), "t_fit": 0.0, "t_recon": 0.0, "final_loss": 1e-5, "status": drivers.FIT_OK}

    with pytest.raises(RuntimeError, match="device lost"):
        drivers.run_volumes(vols, steps=1, fit_fn=fit)
    assert sorted(ran) == [1.0, 2.0, 3.0]                      # the healthy fits were not abandoned
    recs = drivers.run_volumes(vols, steps=1, fit_fn=fit, errors="record")
    assert [r["status"] for r in recs] == [drivers.FIT_OK, drivers.FIT_ERROR, drivers.FIT_OK] and recs[1]["final_loss"] != recs[1]["final_loss"]

    def typo(volume, steps, return_recon=False, **kw):
        raise TypeError("fit_volume() got an unexpected keyword argument 'hiden_features'")

    with pytest.raises(TypeError):
        drivers.run_volumes(vols, steps=1, fit_fn=typo, errors="record")
    with pytest.raises(ValueError):
        drivers.run_volumes(vols, steps=1, fit_fn=fit, errors="ignore")


def _all_fail_worker(rank, world):
    from mri_super_resolution_amd import drivers as D
    vols = [np.full((4, 4), float(k + 1), np.float32) for k in range(4)]

    def fit(volume, steps, return_recon=False, **kw):
        raise RuntimeError("library missing on every rank")

    try:
        D.run_volumes(vols, steps=1, allow_sharding=False, fit_fn=fit, hidden_features=64)
    except D.FitError as e:
        return sorted(int(r["job"]) for r in e.records if r["status"] == D.FIT_ERROR)
    return None


def test_run_volumes_raises_on_every_rank_when_no_rank_survives_world_2_gloo():
    a, b = run_ranks(_all_fail_worker, 2, timeout=120)
    assert a == b == [0, 1, 2, 3]
