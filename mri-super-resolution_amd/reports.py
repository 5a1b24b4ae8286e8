"""The two CSV schemas the reference's drivers write, byte-compatible headers and row formats.

* ``SsimCsv``     -- superresDWI.py:27,36-37,186-187: ``Pt_id, b-value, slice, SSIM-spline, SSIM-SR`` (note the
                     comma-space separators), one row per (slice, b-value).
* ``ContrastCsv`` -- master.py:59-62,190-195,254-259: ``seed,patient,direction,image,metric,performance``.
"""
from __future__ import annotations

import os

SSIM_HEADER = 'Pt_id, b-value, slice, SSIM-spline, SSIM-SR\n'
CONTRAST_HEADER = 'seed,patient,direction,image,metric,performance\n'
CONTRAST_METRICS = ['C', 'CNR', 'CNR2']


class SsimCsv:
    def __init__(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.path = path
        self._fh = open(path, mode='w')
        self._fh.write(SSIM_HEADER)

    def row(self, pt_id, bvalue, _slice, ssim_spline, ssim_sr):
        self._fh.write(f'{pt_id}, {bvalue}, {_slice}, {ssim_spline}, {ssim_sr}\n')

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ContrastCsv:
    def __init__(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.path = path
        with open(path, 'w') as f:
            f.write(CONTRAST_HEADER)

    def rows(self, seed, pt_no, direction, images, contrast_fn):
        """One row per (image, metric): ``contrast_fn(image) -> (C, CNR, CNR2)``."""
        with open(self.path, 'a') as f:
            for image in images.keys():
                values = contrast_fn(images[image])
                for inx, metric in enumerate(CONTRAST_METRICS):
                    f.write('{},{},{},{},{},{}\n'.format(seed, pt_no, direction, image, metric, values[inx]))


def read_csv(path, sep=','):
    """Small reader for tests and drivers: list of dicts keyed by the (stripped) header names."""
    with open(path) as fh:
        lines = [l.rstrip('\n') for l in fh if l.strip()]
    keys = [k.strip() for k in lines[0].split(sep)]
    return [dict(zip(keys, [v.strip() for v in l.split(sep)])) for l in lines[1:]]
