"""``utils/loss.py`` of the reference: ``l1_loss`` (cL1 over the 7 x 7 shifts, loss.py:26-75) and ``psnr`` (cPSNR, loss.py:77-127)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: F401,E402
from mri_super_resolution_amd.rams import l1_loss, psnr  # noqa: F401,E402
