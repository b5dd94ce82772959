"""CPU restatement of the training-target baking.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pinned by tests/golden/bake.npz (G9), produced by the
reference's ``bake_skeleton`` CPU path and ``average_baked_skeletons``.

Tie rule: the reference takes ``torch.cdist(...).argmin(dim=0)`` (lib/skeleton.py:423-428); cdist forms distances as
sqrt(|a|^2 + |b|^2 - 2ab) for larger inputs, so WHICH of several equidistant skeleton points wins depends on float
rounding there.  The restatement uses exact squared distances (integer-valued coordinates times the anisotropy, in
float64) and the first minimal index; tests compare with the fixture exactly where the minimum is unique and by
distance where it is tied.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np


def bake_skeleton(masks: np.ndarray, skeletons: Dict[int, np.ndarray], anisotropy: Sequence[float]) -> np.ndarray:
    """lib/skeleton.py:370-445: baked (3, X, Y, Z) float32; for a voxel of instance k the coordinates of the nearest
    point of skeleton k under the anisotropic Euclidean distance; background 0."""
    X, Y, Z = masks.shape
    baked = np.zeros((3, X, Y, Z), dtype=np.float32)
    an = np.asarray(anisotropy, dtype=np.float64).reshape(1, 3)
    for k in np.unique(masks):
        if k == 0:
            continue
        vox = np.argwhere(masks == k).astype(np.float64)               # (N, 3)
        sk = np.asarray(skeletons[int(k)], dtype=np.float64)           # (M, 3)
        d2 = (((vox[None] - sk[:, None]) * an[None]) ** 2).sum(-1)      # (M, N)
        ind = d2.argmin(axis=0)                                        # first minimal index
        v = vox.astype(np.int64)
        baked[:, v[:, 0], v[:, 1], v[:, 2]] = sk[ind].T.astype(np.float32)
    return baked


def min_distance2(masks: np.ndarray, skeletons: Dict[int, np.ndarray], anisotropy: Sequence[float]) -> np.ndarray:
    """Squared anisotropic distance of every foreground voxel to its instance's nearest skeleton point."""
    out = np.zeros(masks.shape, dtype=np.float64)
    an = np.asarray(anisotropy, dtype=np.float64).reshape(1, 3)
    for k in np.unique(masks):
        if k == 0:
            continue
        vox = np.argwhere(masks == k).astype(np.float64)
        sk = np.asarray(skeletons[int(k)], dtype=np.float64)
        d2 = (((vox[None] - sk[:, None]) * an[None]) ** 2).sum(-1)
        v = vox.astype(np.int64)
        out[v[:, 0], v[:, 1], v[:, 2]] = d2.min(axis=0)
    return out


def average_baked_skeletons(baked: np.ndarray) -> np.ndarray:
    """lib/skeleton.py:18-48: per channel, sum of the zero-padded 3x3x3 neighbourhood divided by the number of its
    entries > 0 (a coordinate equal to 0 counts as empty -- the reference's quirk), count 0 -> 1."""
    c, X, Y, Z = baked.shape
    pad = np.zeros((c, X + 2, Y + 2, Z + 2), dtype=np.float64)
    pad[:, 1:-1, 1:-1, 1:-1] = baked
    s = np.zeros((c, X, Y, Z), dtype=np.float64)
    n = np.zeros((c, X, Y, Z), dtype=np.float64)
    for dx in range(3):
        for dy in range(3):
            for dz in range(3):
                w = pad[:, dx:dx + X, dy:dy + Y, dz:dz + Z]
                s += w
                n += w > 0
    n[n == 0] = 1
    return (s / n).astype(np.float32)
