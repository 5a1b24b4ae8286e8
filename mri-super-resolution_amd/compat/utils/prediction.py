"""``utils/prediction.py`` of the reference: ``predict_tensor(model, x)`` (prediction.py:76-83: fp32 cast -> model -> clip to
[0, 2**16] -> round half to even) and the RAMS+ test-time ensembling helpers (prediction.py:10-74, 86-97)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: F401,E402
from mri_super_resolution_amd.rams import (ensemble, flip, geometric_ensemble, predict_tensor,  # noqa: F401,E402
                                           predict_tensor_permute, random_ensemble, rotate, shuffle_last_axis, unensemble)
