"""AutoERD acceptance weights (master.py:77-93).  CPU: the NumPy oracle against the partitions scikit-learn itself produced
(tests/golden/erd.npz, oracle/gen_golden_erd.py).  -m gpu: the device kernel against those fixtures and against the oracle's two
rejection rules on a slice-shaped stack, and the master entry script with --erd."""
import json
import os

import numpy as np
import pytest

from oracle import erd_oracle as E


def _samples(golden):
    g = golden("erd.npz")
    for v, n, l in zip(g["values"], g["lengths"], g["labels"]):
        yield v[:n], l[:n]


def test_oracle_partitions_equal_sklearn_fixture(golden):
    count = 0
    for x, lab in _samples(golden):
        m = E.complete_linkage_two_clusters(x)
        assert np.array_equal(m, lab == lab[0]) or np.array_equal(~m, lab == lab[0]), (x, lab, m)
        count += 1
    assert count == 2460


def test_oracle_rules():
    x = np.array([10.0, 11.0, 10.5, 50.0, 10.2, 10.9])          # one bright outlier among six
    assert list(E.accept_mask(x, 1, 6)) == [1, 1, 1, 0, 1, 1]    # majority (5 >= 4) rejects the outlier
    assert list(E.accept_mask(x, 2, 6)) == [0, 0, 0, 1, 0, 0]    # intensity-cognisant keeps the brighter cluster
    assert list(E.accept_mask(x, 2, 6, erd_positive=False)) == [1] * 6
    y = np.array([1.0, 2.0, 9.0, 10.0])                          # 2 + 2: no cluster reaches 2/3 of 4
    assert list(E.accept_mask(y, 1, 4)) == [1, 1, 1, 1]
    z = np.array([1.0, 1.0, 1.0])                                # (2/3) * 3 == 2.0 in floating point: two equal points suffice
    assert int(E.accept_mask(z, 1, 3).sum()) == 2
    with pytest.raises(ValueError):
        E.accept_mask(x, 3, 6)


@pytest.mark.gpu
def test_device_partitions_equal_sklearn_fixture(golden):
    """Every fixture sample as one 'pixel' (grouped by length): under rule 2 with distinct cluster means the device's accept
    mask IS one of sklearn's two clusters; under rule 1 it follows from the cluster sizes."""
    from mri_super_resolution_amd import erd
    by_n = {}
    for x, lab in _samples(golden):
        by_n.setdefault(x.size, []).append((x, lab))
    checked = 0
    for n, items in by_n.items():
        img = np.stack([x for x, _ in items]).reshape(1, len(items), n)
        for rule in (1, 2):
            got = erd.auto_erd(img, rule)[0]
            for (x, lab), keep in zip(items, got):
                groups = (lab == 0, lab == 1)
                want = np.ones(n, np.int64)
                if rule == 1:
                    for k in range(2):
                        if groups[k].sum() >= (2 / 3) * n:
                            want[groups[1 - k]] = 0
                else:
                    means = [x[g].mean() for g in groups]
                    for k in range(2):
                        if means[k] > means[1 - k]:
                            want[groups[1 - k]] = 0
                assert np.array_equal(keep, want), (rule, x, lab, keep)
                checked += 1
    assert checked == 2 * 2460


@pytest.mark.gpu
def test_device_rules_on_a_slice_vs_oracle():
    from mri_super_resolution_amd import erd
    rng = np.random.default_rng(8)
    img = np.round(rng.normal(300.0, 25.0, (20, 17, 12)))          # integer-valued, 12 acquisitions (4 + 4 + 4)
    img[rng.random((20, 17, 12)) < 0.08] += 150.0                  # motion-corrupted acquisitions
    emap = rng.random((20, 17)) - 0.3
    for rule, em in ((1, None), (2, emap), (2, None)):
        assert np.array_equal(erd.auto_erd(img, rule, em), E.auto_erd(img, rule, em)), rule
    assert np.array_equal(erd.auto_erd(img.astype(np.float32), 1), E.auto_erd(img, 1))
    with pytest.raises(Exception):
        erd.auto_erd(np.zeros((2, 2, 1)), 1)                       # a single acquisition cannot be clustered
    with pytest.raises(Exception):
        erd.auto_erd(img, 3)


@pytest.mark.gpu
def test_master_script_with_auto_erd(tmp_path, golden):
    """`master.py --erd 1` (experiments/sr1_exp_2.txt: weight = True): the acceptance weights come from the device kernel,
    enter the weighted loss and the 'ERD' image; one corrupted acquisition is rejected where it was planted."""
    from mri_super_resolution_amd import contrast, matio, reports
    from mri_super_resolution_amd.scripts import master as master_script
    vol = golden("pat07_volume.npz")["vol"].astype(np.float64)
    rng = np.random.default_rng(6)
    dwi = np.stack([0.4 * vol * (1 + 0.01 * rng.standard_normal(vol.shape)) for _ in range(6)], axis=-1)
    dwi[50:70, 50:70, 11, 2] *= 0.2                                 # signal dropout in acquisition 2 (a motion artefact)
    data_dir = tmp_path / "anon_data"
    data_dir.mkdir()
    matio.savemat(str(data_dir / "pat07_alldata.mat"), {"data": dwi.astype(np.float32)})
    matio.savemat(str(data_dir / "pat07_mean_b0.mat"), {"data_mean_b0": vol.astype(np.float32)})
    spec = [{"pt_id": "18-1681-07", "b": 900, "cancer_loc": [60, 70], "contralateral_loc": [60, 55], "noise": [45, 45],
             "cancer_slice": 11, "acquisitions": [2, 2, 2]}]
    with open(str(tmp_path / "cases.json"), "w") as fh:
        json.dump(spec, fh)
    cases = master_script.load_cases(master_script.build_parser().parse_args(["--data_dir", str(data_dir), "--cases",
                                                                               str(tmp_path / "cases.json")]))
    args = master_script.build_parser().parse_args(["--out_folder", str(tmp_path / "exp"), "--out_img_folder", str(tmp_path / "img"),
                                                    "--total_steps", "30", "--seg", "10", "--hidden_layers", "2",
                                                    "--hidden_features", "32", "--scale", "2", "--exp_name", "e1", "--erd", "1"])
    out = master_script.run(args, cases)
    acc = cases[0].accept
    want = E.auto_erd(cases[0].dwi[40:100, 40:100, 11, :], 1)
    assert np.array_equal(acc[40:100, 40:100, 11, :], want)
    assert acc[55:65, 55:65, 11, 2].sum() == 0 and acc[55:65, 55:65, 11, [0, 1, 3, 4, 5]].all()      # the dropout is rejected
    assert acc[:40].all() and acc[..., 10, :].all()                                                    # nothing outside the ROI / slice
    rows = reports.read_csv(out["csv"])
    erd_rows = [float(r["performance"]) for r in rows if r["image"] == "ERD" and r["direction"] == "y"]
    mean_rows = [float(r["performance"]) for r in rows if r["image"] == "mean" and r["direction"] == "y"]
    assert len(erd_rows) == 3 and erd_rows != mean_rows and all(np.isfinite(erd_rows))
