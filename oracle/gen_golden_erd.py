#!/usr/bin/env python3
"""Fixtures for the AutoERD acceptance weights (master.py:77-93): the two-cluster partitions of
``sklearn.cluster.AgglomerativeClustering(n_clusters=2, metric='euclidean', linkage='complete')`` -- the reference's call
(``affinity=`` in its scikit-learn) -- on 1-D samples of the kinds the driver meets: continuous values, integer-valued
intensities (many equal distances: the partition then hangs on scipy's tie handling), samples with duplicates, two-point
samples, n = 2 .. 16.  Build container only (scikit-learn 1.7.2, scipy 1.15.3); writes tests/golden/erd.npz (numbers only):
samples (NaN-padded to 16), their lengths, and the sklearn labels.
    python oracle/gen_golden_erd.py
"""
import os
import warnings

import numpy as np
from sklearn.cluster import AgglomerativeClustering

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "erd.npz")


def main():
    rng = np.random.default_rng(20261004)
    samples = []
    for n in range(2, 17):
        for _ in range(40):
            samples.append(rng.random(n) * 100.0)                              # continuous
            samples.append(rng.integers(0, 12, n).astype(np.float64))           # small integers: ties and duplicates everywhere
            samples.append(rng.integers(100, 400, n).astype(np.float64))        # integer-valued intensities
            samples.append(np.round(rng.normal(200.0, 30.0, n)))                # the same, bell-shaped
        samples.append(np.arange(n, dtype=np.float64))                          # equally spaced: every adjacent distance ties
        samples.append(np.arange(n, dtype=np.float64)[::-1].copy())
        samples.append(np.full(n, 7.0))                                         # all equal
        x = np.zeros(n)
        x[-1] = 1.0
        samples.append(x)                                                       # one outlier
    vals = np.full((len(samples), 16), np.nan)
    lens = np.zeros(len(samples), np.int64)
    labels = np.full((len(samples), 16), -1, np.int64)
    for k, x in enumerate(samples):
        db = AgglomerativeClustering(n_clusters=2, metric="euclidean", linkage="complete").fit(x.reshape(-1, 1))
        vals[k, :x.size] = x
        lens[k] = x.size
        labels[k, :x.size] = db.labels_
    np.savez_compressed(OUT, values=vals, lengths=lens, labels=labels)
    print(f"{len(samples)} samples -> {OUT}")


if __name__ == "__main__":
    main()
