"""Drop-in for the INR-side ``nn_mri`` module (master.py:1, INR_ERD.py:1, automate_INR.py:9): 2-D
``get_mgrid(sidelen, dim)``, PIL-style ``ImageFitting_set``, ``Siren``/``SineLayer``/``PN``/
``input_mapping``.  ``Siren`` returns ``(output, coords)`` as master.py:142 expects when constructed
through this module.  master.py:1's whole import line resolves: ``cases`` (the patient list the reference never
published: empty until a driver fills it), ``case``, ``calculate_contrast`` (host arithmetic as in the
reference) and ``save_dicom`` (DICOM export is out of scope: the name raises a clear error when called)."""
import _bootstrap  # noqa: F401
from mri_super_resolution_amd import inr as _inr  # noqa: E402
from mri_super_resolution_amd.contrast import calculate_contrast, case, cases, save_dicom  # noqa: F401,E402
from mri_super_resolution_amd.inr import ImageFitting_set, SineLayer, get_mgrid, input_mapping  # noqa: F401,E402


class Siren(_inr.Siren):
    def __init__(self, in_features, hidden_features, hidden_layers, out_features, first_omega_0=30.,
                 hidden_omega_0=30., return_coords=True):
        super().__init__(in_features, hidden_features, hidden_layers, out_features, first_omega_0,
                         hidden_omega_0, flavor="SRDWI", return_coords=return_coords)


class PN(_inr.PN):
    def __init__(self, in_features, hidden_features):
        super().__init__(in_features, hidden_features, 2)
