"""Drop-in for the reference's ``multi-image-super-resolution/utils`` package: with ``compat/`` first on ``sys.path`` the import
lines of ``multi-image-super-resolution/master.py:1-3`` -- ``from nn_mri import cases, save_dicom``, ``from utils.network import
RAMS``, ``from utils.prediction import predict_tensor`` -- resolve to the MI355X implementation (``mri_super_resolution_amd.rams``).
RAMS parity is UNPINNED (TensorFlow absent, shipped checkpoints incomplete: SURVEY.md 8c)."""
