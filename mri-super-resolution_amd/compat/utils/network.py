"""``utils/network.py`` of the reference: ``RAMS(scale, filters, kernel_size, channels, r, N)`` (network.py:91) builds the network
and returns a callable model; here the object whose ``__call__`` is ``inr_rams_forward``.  ``MEAN`` / ``STD`` are network.py:18-19."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: F401,E402
from mri_super_resolution_amd.rams import MEAN, STD, RAMS  # noqa: F401,E402
