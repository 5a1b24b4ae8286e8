"""AutoERD acceptance weights -- TEST INFRASTRUCTURE (oracle), never imported by the product package.

Restates ``implicit-neural-representations/master.py:77-93``: for every pixel of the ROI the values of all acquisitions
(a 1-D sample of n <= ~12 numbers) are split into two clusters by
``sklearn.cluster.AgglomerativeClustering(n_clusters=2, affinity='euclidean', linkage='complete')`` and acquisitions are
rejected (``case.accept[...] = 0``) by one of two rules:

* ``--erd 1`` (majority voting, master.py:85-88): if one cluster holds at least 2/3 of the acquisitions, the other is rejected;
* ``--erd 2`` (intensity-cognisant, master.py:89-93): where the patient's ERD map is positive, the cluster with the LOWER mean
  is rejected.

The clustering itself is third-party code that is not in /root/reference: scikit-learn (1.7.2 in this image; the reference's
own pin spells the metric argument ``affinity``) routes ``linkage='complete'`` without a connectivity matrix to
``scipy.cluster.hierarchy.linkage(X, 'complete', 'euclidean')`` (scipy 1.15.3 here), i.e. the nearest-neighbour-chain algorithm
of ``scipy/cluster/_hierarchy.pyx:nn_chain`` followed by a STABLE sort of the merges by distance, and cuts the tree at its last
merge.  With integer-valued intensities equal distances are common, so the partition depends on how ties fall: the chain walk
and the stable sort are restated step by step below, and pinned against sklearn itself on random, integer-valued (tie-rich) and
duplicate-rich samples (tests/golden/erd.npz, written by oracle/gen_golden_erd.py).
"""
from __future__ import annotations

import numpy as np


def complete_linkage_two_clusters(values) -> np.ndarray:
    """Boolean mask (len n): membership of one of the TWO clusters complete-linkage agglomeration leaves of the 1-D sample
    `values` (which of the two is True is arbitrary: both rejection rules are symmetric in the labels).  n >= 2."""
    x = np.asarray(values, np.float64).reshape(-1)
    n = x.size
    if n < 2:
        raise ValueError("need at least two acquisitions")
    D = np.abs(x[:, None] - x[None, :])            # pairwise Euclidean distances of 1-D points (float64, as scipy computes them)
    size = np.ones(n, np.int64)
    chain = np.zeros(n, np.int64)
    chain_len = 0
    merges = []                                    # (x, y, dist) in the order nn_chain finds them
    for _ in range(n - 1):
        if chain_len == 0:
            chain_len = 1
            chain[0] = int(np.flatnonzero(size > 0)[0])
        while True:
            a = chain[chain_len - 1]
            if chain_len > 1:                      # prefer the previous element of the chain on ties (no cycles)
                b = chain[chain_len - 2]
                cur = D[a, b]
            else:
                b = -1
                cur = np.inf
            for i in range(n):
                if size[i] == 0 or i == a:
                    continue
                if D[a, i] < cur:
                    cur = D[a, i]
                    b = i
            if chain_len > 1 and b == chain[chain_len - 2]:
                break
            chain[chain_len] = b
            chain_len += 1
        chain_len -= 2
        lo, hi = (a, b) if a < b else (b, a)
        merges.append((lo, hi, cur))
        nx, ny = size[lo], size[hi]
        size[lo] = 0                               # cluster lo is dropped, hi becomes the union
        size[hi] = nx + ny
        for i in range(n):
            if size[i] == 0 or i == hi:
                continue
            D[i, hi] = D[hi, i] = max(D[i, lo], D[i, hi])      # complete linkage
    # scipy then sorts the merges by distance (STABLE) and relabels them with a union-find over the POINTS lo / hi (slot i always
    # contains point i); sklearn cuts the tree at the last merge of that list: unite all but the last, two components remain
    order = np.argsort(np.asarray([m[2] for m in merges]), kind="mergesort")
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for k in order[:-1]:
        lo, hi, _ = merges[k]
        parent[find(lo)] = find(hi)
    root = find(0)
    return np.asarray([find(i) == root for i in range(n)])


def accept_mask(values, rule: int, n_total: int, erd_positive: bool = True) -> np.ndarray:
    """master.py:81-93 for one pixel: 1 = keep, 0 = reject, per acquisition."""
    x = np.asarray(values, np.float64).reshape(-1)
    keep = np.ones(x.size, np.int64)
    in0 = complete_linkage_two_clusters(x)
    groups = (in0, ~in0)
    if rule == 1:
        for k in range(2):
            if groups[k].sum() >= (2 / 3) * n_total:
                keep[groups[1 - k]] = 0
    elif rule == 2:
        if erd_positive:
            means = [x[g].mean() for g in groups]
            for k in range(2):
                if means[k] > means[1 - k]:
                    keep[groups[1 - k]] = 0
    else:
        raise ValueError("rule must be 1 (majority voting) or 2 (intensity-cognisant)")
    return keep


def auto_erd(img, rule: int, erd_map=None) -> np.ndarray:
    """`img` [H, W, n] -> accept [H, W, n] (int64 0/1); `erd_map` [H, W] for rule 2 (master.py:89: ``case.erd[...] > 0``)."""
    img = np.asarray(img)
    H, W, n = img.shape
    out = np.ones((H, W, n), np.int64)
    for i in range(H):
        for j in range(W):
            pos = True if erd_map is None else bool(erd_map[i, j] > 0)
            out[i, j] = accept_mask(img[i, j], rule, n, pos)
    return out
