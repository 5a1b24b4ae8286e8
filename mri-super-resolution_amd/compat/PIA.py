"""Drop-in for the fit half of the reference's ``PIA`` module: ``from PIA import hybrid_fit``
(superresHybrid.py:14) resolves to the MI355X kernel.  The `PIA` autoencoder class itself is outside SURVEY.md 8."""
import _bootstrap  # noqa: F401
from mri_super_resolution_amd.pia import hybrid_fit, three_compartment_fit  # noqa: F401,E402
