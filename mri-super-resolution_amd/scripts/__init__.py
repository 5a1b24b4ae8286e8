"""Entry points with the reference drivers' call surface (``superresDWI.py``, ``master.py``): ``.mat`` in -> up-scaled volume
(``.mat`` / ``.npy``) + the reference's CSV files out.  ``python -m mri_super_resolution_amd.scripts.superresDWI --help``."""
