"""CPU: the MATLAB 7.3 (HDF5) reader behind ``matio.loadmat`` -- the reference's ``mat73.loadmat`` fallback (superresDWI.py:40-43).
Files come from tests/hdf5_fixture.py, a byte-level writer of the layout ``save -v7.3`` produces (no HDF5 library here: reader and
writer are two restatements of the HDF5 File Format Specification, unpinned against MATLAB itself)."""
import numpy as np
import pytest

from mri_super_resolution_amd import matio
from mri_super_resolution_amd.mat73io import loadmat73
from tests.hdf5_fixture import Writer, write_mat73


def test_numeric_arrays_in_every_layout(tmp_path):
    rng = np.random.default_rng(0)
    A = rng.random((3, 4))
    V = rng.random((13, 9, 7)).astype(np.float32)
    I = rng.integers(-300, 300, (5, 2)).astype(np.int16)
    w = Writer()
    vars_ = {"A": w.dataset(A, "double", with_max=True),
             "small": w.dataset(np.arange(6, dtype="<i4").reshape(2, 3), "int32", layout="compact"),
             "V": w.dataset(V, "single", layout="chunked", chunk=(3, 4, 5)),                  # ragged edge chunks in every axis
             "Vs": w.dataset(V, "single", layout="chunked", chunk=(7, 9, 4), shuffle=True, split=True),
             "V2": w.dataset(V, "single", layout="chunked", chunk=(2, 3, 5), two_level=True),  # a chunk B-tree with an inner node
             "I": w.dataset(I, "int16"),
             "u8": w.dataset(np.arange(40, dtype="<u1").reshape(8, 5), "uint8"),
             "scalar": 3.5, "empty": np.zeros((0, 3)), "flag": np.array([[True, False, True]])}
    path = str(tmp_path / "v73.mat")
    write_mat73(path, vars_, w)
    got = matio.loadmat(path)                                   # dispatches on the HDF5 signature behind the 512-byte user block
    assert set(got) == set(vars_)
    assert got["A"].dtype == np.float64 and np.array_equal(got["A"], A)
    assert np.array_equal(got["small"], np.arange(6).reshape(2, 3)) and got["small"].dtype == np.int32
    for k in ("V", "Vs", "V2"):
        assert got[k].dtype == np.float32 and got[k].shape == (13, 9, 7) and np.array_equal(got[k], V), k
    assert np.array_equal(got["I"], I) and np.array_equal(got["u8"], np.arange(40).reshape(8, 5))
    assert got["scalar"].shape == (1, 1) and got["scalar"][0, 0] == 3.5
    assert got["empty"].shape == (0, 3) and got["flag"].dtype == bool and got["flag"].tolist() == [[True, False, True]]


def test_cells_structs_strings_and_the_hybrid_raw_shape(tmp_path):
    """``hybrid_raw`` of the reference's master.mat files: a 4 x 4 cell (b-value x echo time) of (X, Y, Z[, n]) volumes, read as
    ``data['hybrid_raw'][b][te]`` (superresDWI.py:44-55)."""
    rng = np.random.default_rng(1)
    cell = np.empty((4, 4), dtype=object)
    for b in range(4):
        for te in range(4):
            cell[b, te] = rng.random((6, 5, 3) if b == 0 else (6, 5, 3, b + 1)).astype(np.float32)
    vars_ = {"hybrid_raw": cell, "name": "pat099", "info": {"TE": np.array([[57.0, 70.0, 150.0, 200.0]]), "site": "x",
                                                              "nested": {"k": np.int16(4)}},
             "row": [np.arange(3.0), "txt", np.zeros((2, 2))]}
    path = str(tmp_path / "master.mat")
    write_mat73(path, vars_)
    got = loadmat73(path)
    assert set(got) == {"hybrid_raw", "name", "info", "row"}    # '#refs#' is the container of the cell elements, not a variable
    hr = got["hybrid_raw"]
    assert hr.shape == (4, 4) and hr.dtype == object
    for b in range(4):
        for te in range(4):
            assert np.array_equal(hr[b][te], cell[b, te]) and hr[b][te].dtype == np.float32
    assert got["name"] == "pat099" and got["info"]["site"] == "x" and np.array_equal(got["info"]["TE"], [[57.0, 70.0, 150.0, 200.0]])
    assert got["info"]["nested"]["k"].item() == 4
    assert got["row"].shape == (1, 3) and np.array_equal(got["row"][0, 0], [[0.0, 1.0, 2.0]]) and got["row"][0, 1] == "txt"


def test_load_mat_volume_reads_a_v73_file_and_rejections(tmp_path):
    from mri_super_resolution_amd import drivers
    vol = np.random.default_rng(2).random((8, 8, 4)).astype(np.float32)
    w = Writer()
    path = str(tmp_path / "pat_mean_b0.mat")
    write_mat73(path, {"data_mean_b0": w.dataset(vol, "single", layout="chunked", chunk=(2, 4, 4))}, w)
    assert np.array_equal(drivers.load_mat_volume(path), vol)
    bad = tmp_path / "bad.mat"
    bad.write_bytes(b"MATLAB 7.3 MAT-file".ljust(512) + b"not an hdf5 file at all".ljust(256))
    with pytest.raises(matio.MatFormatError, match="superblock"):
        matio.loadmat(str(bad))
    cut = open(path, "rb").read()[:900]                        # truncated behind the superblock: a format error, not an IndexError
    (tmp_path / "cut.mat").write_bytes(cut)
    with pytest.raises(matio.MatFormatError, match="damaged or truncated|superblock|signature"):
        matio.loadmat(str(tmp_path / "cut.mat"))
    # a version-2 object header (newer libver, not MATLAB): named, not mis-read
    good = bytearray(open(path, "rb").read())
    from mri_super_resolution_amd.mat73io import _File
    f = _File(bytes(good), path)
    hdr = f.abs(f.group_members(f.root["btree"], f.root["heap"])["data_mean_b0"])
    good[hdr:hdr + 4] = b"OHDR"
    (tmp_path / "v2.mat").write_bytes(bytes(good))
    with pytest.raises(matio.MatFormatError, match="version-2 object header"):
        matio.loadmat(str(tmp_path / "v2.mat"))


def test_the_hybrid_entry_script_reads_v5_and_v73_master_files_alike(tmp_path):
    """``scripts/superresHybrid.py`` (superresHybrid.py:34-54): the same ``hybrid_raw`` cell saved as MAT-5 and as MATLAB 7.3 loads
    to the same [X, Y, Z, b, TE] array -- the reference reads the first with scipy and the second with mat73."""
    from mri_super_resolution_amd.scripts import superresHybrid as S
    rng = np.random.default_rng(3)
    cell = np.empty((4, 4), dtype=object)
    for b in range(4):
        for te in range(4):
            cell[b, te] = rng.random((6, 6, 3) if b == 0 else (6, 6, 3, 2)).astype(np.float32)
    extra = {"b": np.array([[0.0, 150.0, 1000.0, 1500.0]]), "TE": np.array([[57.0, 70.0, 150.0, 200.0]])}
    matio.savemat(str(tmp_path / "v5.mat"), {"hybrid_raw": cell, **extra})
    write_mat73(str(tmp_path / "v73.mat"), {"hybrid_raw": cell, **extra})
    raw5, b5, te5 = S.load_hybrid(str(tmp_path / "v5.mat"))
    raw7, b7, te7 = S.load_hybrid(str(tmp_path / "v73.mat"))
    assert raw5.shape == (6, 6, 3, 4, 4) and np.array_equal(raw5, raw7) and np.array_equal(b5, b7) and np.array_equal(te5, te7)


def test_stored_base_address_zero_behind_the_user_block(tmp_path):
    """libhdf5 resolves every address against the place where it FOUND the superblock (the 512-byte user block) and
    overrides the superblock's stored base-address field when the two differ -- a file whose stored field is 0 reads the same."""
    rng = np.random.default_rng(4)
    V = rng.random((6, 5, 4)).astype(np.float32)
    got = {}
    for stored in (512, 0):
        w = Writer()
        w.stored_base = stored
        vars_ = {"V": w.dataset(V, "single", layout="chunked", chunk=(4, 3, 3)), "c": [np.arange(3.0), "txt"]}
        path = str(tmp_path / f"b{stored}.mat")
        write_mat73(path, vars_, w)
        got[stored] = matio.loadmat(path)
    for g in got.values():
        assert np.array_equal(g["V"], V) and np.array_equal(np.asarray(g["c"].reshape(-1)[0]).reshape(-1), np.arange(3.0)) \
            and g["c"].reshape(-1)[1] == "txt"
