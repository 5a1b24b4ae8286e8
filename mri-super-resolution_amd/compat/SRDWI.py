"""Drop-in for the reference's ``SRDWI`` module: put this directory first on ``sys.path`` and
``from SRDWI import calculate_combinations, ImageFitting_set, Siren, PN, get_mgrid, input_mapping,
calculate_ADC, resize_array`` (superresDWI.py:13, superresHybrid.py:13, forbagci.py:10) resolves to
the MI355X implementation."""
import _bootstrap  # noqa: F401
from mri_super_resolution_amd.inr import (ImageFitting_set, PN, SineLayer, Siren, calculate_ADC,  # noqa: F401,E402
                                          calculate_combinations, get_mgrid, input_mapping, resize_array)
