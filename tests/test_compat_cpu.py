"""CPU: the drop-in module surface resolves exactly as the reference's scripts import it (no GPU needed to import / construct)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "mri-super-resolution_amd", "compat")


def _run(code):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp",
                         env=dict(os.environ, PYTHONPATH=COMPAT))
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_multi_image_master_import_lines_resolve():
    """multi-image-super-resolution/master.py:1-3 verbatim, then the constructor call of :29-30 and the surface of :50."""
    out = _run("from nn_mri import cases, save_dicom\n"
               "from utils.network import RAMS\n"
               "from utils.prediction import predict_tensor\n"
               "from utils.loss import l1_loss, psnr\n"
               "rams_network = RAMS(scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12)\n"
               "import inspect\n"
               "print(len(rams_network.specs), callable(rams_network), list(inspect.signature(predict_tensor).parameters), cases)\n")
    assert out.strip() == "71 True ['model', 'x'] []"


def test_inr_import_lines_resolve():
    """superresDWI.py:13, superresHybrid.py:14, inrDWI.py:9 and master.py:1 (INR side)."""
    out = _run("from SRDWI import calculate_combinations, ImageFitting_set, Siren, PN, get_mgrid, input_mapping, calculate_ADC, resize_array\n"
               "from PIA import hybrid_fit\n"
               "import INRmodel\n"
               "from nn_mri import cases, calculate_contrast, save_dicom, Siren as Siren2d\n"
               "m = Siren(256, 512, 3, 1)\n"
               "print(sorted(m.state_dict())[:2], sum(p.numel() for p in m.parameters()))\n")
    assert "920065" in out


def test_rams_weights_round_trip(tmp_path):
    import numpy as np
    import pytest

    from mri_super_resolution_amd import rams
    a = rams.RAMS(3, 32, 3, 9, 8, 12, seed=1)
    path = a.save_weights(str(tmp_path / "w.npz"))
    b = rams.RAMS(3, 32, 3, 9, 8, 12, seed=2).load_weights(path)
    assert a.params.keys() == b.params.keys() and all(np.array_equal(a.params[k], b.params[k]) for k in a.params)
    with pytest.raises(ValueError):
        rams.RAMS(3, 16, 3, 9, 8, 12).load_weights(path)              # another width: shapes are checked
    bad = dict(a.params)
    bad.pop("stem/g")
    np.savez(str(tmp_path / "bad.npz"), **bad)
    with pytest.raises(KeyError):
        rams.RAMS(3, 32, 3, 9, 8, 12).load_weights(str(tmp_path / "bad.npz"))
