"""Lesion-contrast bookkeeping of the 2-D slice driver: ``case`` / ``cases`` / ``calculate_contrast`` (nn_mri.py:28-85,
imported by master.py:1), ``minmax_normalize`` and ``calc_adc`` (master.py:45-51).  Host-side, a few 2 x 2 regions per
image -- exactly where the reference computes them."""
from __future__ import annotations

import os

import numpy as np

from . import matio

eps = 1e-7      # master.py:42
mag = 1000      # master.py:43


class case:
    """nn_mri.py:28-56: one patient of the 2-D study.  ``data_dir`` (default ``../anon_data`` as in the reference) holds
    ``pat<NN>_alldata.mat`` (``data`` [X, Y, Z, acquisitions]), ``pat<NN>_mean_b0.mat`` and ``pat<NN>_ERD.mat``."""

    def __init__(self, pt_id, b, cancer_loc, contralateral_loc, noise, cancer_slice, acquisitions, data_dir="../anon_data"):
        self.pt_id = pt_id
        self.cancer_loc = cancer_loc
        self.contralateral_loc = contralateral_loc
        self.noise = noise
        self.cancer_slice = cancer_slice
        self.acquisitions = acquisitions
        self.b = b
        pt_no = self.pt_id.split('-')[-1]
        self.dwi = matio.loadmat(os.path.join(data_dir, 'pat' + pt_no + '_alldata.mat'))['data']
        self.b0 = matio.loadmat(os.path.join(data_dir, 'pat' + pt_no + '_mean_b0.mat'))['data_mean_b0']
        self.accept = np.ones(self.dwi.shape, dtype=int)
        erd_path = os.path.join(data_dir, 'pat' + pt_no + '_ERD.mat')
        self.erd = matio.loadmat(erd_path)['ADC_alldata_mm_ERD'] if os.path.exists(erd_path) else None


# master.py:1 imports a module-level list ``cases`` that nn_mri.py never defines (the patient table was not published);
# drivers fill it (``cases.append(case(...))``) or pass their own list to ``scripts.master.run``.
cases = []


def calculate_contrast(case, scale, image, focus):
    """nn_mri.py:59-85: contrast C, CNR and CNR2 between the 2*scale-wide squares around the cancer, contralateral and
    noise locations (given in full-image pixels; ``focus`` = ROI origin).  Same arithmetic, same order."""
    cc_x, cc_y = tuple((i - focus) * scale for i in case.cancer_loc)
    cb_x, cb_y = tuple((i - focus) * scale for i in case.contralateral_loc)
    cn_x, cn_y = tuple((i - focus) * scale for i in case.noise)
    cancer_area = image[cc_x - scale: cc_x + scale, cc_y - scale: cc_y + scale]
    contralateral_area = image[cb_x - scale: cb_x + scale, cb_y - scale: cb_y + scale]
    noise_area = image[cn_x - scale: cn_x + scale, cn_y - scale: cn_y + scale]
    varc = np.std(cancer_area) ** 2
    varb = np.std(contralateral_area) ** 2
    varn = np.std(noise_area)           # (the reference divides by the noise STANDARD DEVIATION and calls it a variance)
    C = cancer_area.mean() / (contralateral_area.mean() + 1e-7)
    CNR = abs(cancer_area.mean() - contralateral_area.mean()) / np.sqrt(varc + varb)
    CNR2 = abs(cancer_area.mean() - contralateral_area.mean()) / varn
    return C, CNR, CNR2


def minmax_normalize(img, ref):
    """master.py:45-47."""
    return ((img - img.min()) / (img.max() - img.min())) * (ref.max() - ref.min()) + ref.min()


def calc_adc(dwi, b0, b):
    """master.py:49-51: ADC in 1e-6 mm^2/s from one DWI and the b = 0 image."""
    adc = -np.log((dwi / (b0 + eps)) + eps) / b
    return adc * mag * mag


def save_dicom(img, filename):
    """nn_mri.py:19-26 writes DICOM through SimpleITK: out of the accelerated path (OUT OF SCOPE).  The name
    exists so that ``from nn_mri import ... save_dicom`` (master.py:1) resolves; calling it says why nothing is written."""
    raise NotImplementedError("save_dicom: DICOM export (SimpleITK) is outside this build's scope; use matio.savemat / numpy "
                              f"to store '{filename}'")
