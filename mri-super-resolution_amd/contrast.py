"""Lesion-contrast bookkeeping of the 2-D slice driver: ``case`` / ``cases`` / ``calculate_contrast`` (nn_mri.py:28-85,
imported by master.py:1), ``minmax_normalize`` and ``calc_adc`` (master.py:45-51).  Host-side, a few 2 x 2 regions per
image -- exactly where the reference computes them."""
from __future__ import annotations

import os

import numpy as np

from . import matio

eps = 1e-7      # master.py:42
mag = 1000      # master.py:43


_PATIENT_FILES = {"dwi": ("_alldata.mat", "data"), "b0": ("_mean_b0.mat", "data_mean_b0"), "erd": ("_ERD.mat", "ADC_alldata_mm_ERD")}


def _patient_array(data_dir, pt_no, kind, required=True):
    suffix, key = _PATIENT_FILES[kind]
    path = os.path.join(data_dir, f"pat{pt_no}{suffix}")
    if not required and not os.path.exists(path):
        return None
    return matio.loadmat(path)[key]


class case:
    """One patient of the 2-D study -- the record ``master.py`` iterates over (constructor arguments and attribute names are
    the reference's, nn_mri.py:28-56, because the driver reads them: ``pt_id, b, cancer_loc, contralateral_loc, noise,
    cancer_slice, acquisitions`` plus the loaded ``dwi`` [X, Y, Z, acquisitions], ``b0``, ``accept`` (all ones) and ``erd``).
    ``data_dir`` defaults to the reference's ``../anon_data``; the ERD file is optional here (it is stripped from the
    published data), the other two are not."""

    def __init__(self, pt_id, b, cancer_loc, contralateral_loc, noise, cancer_slice, acquisitions, data_dir="../anon_data"):
        self.pt_id, self.b = pt_id, b
        self.cancer_loc, self.contralateral_loc, self.noise = cancer_loc, contralateral_loc, noise
        self.cancer_slice, self.acquisitions = cancer_slice, acquisitions
        pt_no = pt_id.split("-")[-1]
        self.dwi = _patient_array(data_dir, pt_no, "dwi")
        self.b0 = _patient_array(data_dir, pt_no, "b0")
        self.erd = _patient_array(data_dir, pt_no, "erd", required=False)
        self.accept = np.ones(self.dwi.shape, dtype=int)


# master.py:1 imports a module-level list ``cases`` that nn_mri.py never defines (the patient table was not published);
# drivers fill it (``cases.append(case(...))``) or pass their own list to ``scripts.master.run``.
cases = []


def _square(image, centre, focus, scale):
    """The 2*scale-wide square around a landmark given in full-image pixels, on an image whose origin is ``focus`` and
    whose pixels are 1/scale of the original."""
    r, c = ((int(v) - focus) * scale for v in centre)
    return np.asarray(image)[r - scale:r + scale, c - scale:c + scale]


def calculate_contrast(case, scale, image, focus):
    """Lesion conspicuity of one image (nn_mri.py:59-85): ratio of the mean signal in the square around the lesion to the one
    around its contralateral mirror point, their difference over the pooled standard deviation of the two squares, and the same
    difference over the standard deviation of a background square (the reference names that last divisor a variance; it is a
    standard deviation, kept).  Returns ``(C, CNR, CNR2)``; pinned by tests/golden/contrast.npz."""
    lesion, mirror, background = (_square(image, p, focus, scale) for p in (case.cancer_loc, case.contralateral_loc, case.noise))
    gap = abs(lesion.mean() - mirror.mean())
    pooled = np.sqrt(np.std(lesion) ** 2 + np.std(mirror) ** 2)
    return lesion.mean() / (mirror.mean() + 1e-7), gap / pooled, gap / np.std(background)


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
