"""Seeded synthetic network-output fields for the post-model stages:
non-overlapping ellipsoids, prob ~ 1 inside, vectors pointing at the blob centre
(clipped to [-1, 1] after division by the vector scaling), a small ball of skeleton
probability at the centre.  Analytic answer: every blob voxel receives its blob's
label, so the number of instances equals the number of blobs placed."""
from __future__ import annotations

import numpy as np
import torch


def blob_field(shape, seed=0, n_blobs=40, scale=(60, 60, 12), rmax=(18, 18, 5), margin=(52, 52, 6),
               dtype=torch.float16, noise=0.01):
    """Returns (out (5, X, Y, Z) tensor, number of blobs placed)."""
    X, Y, Z = shape
    rng = np.random.default_rng(seed)
    out = np.zeros((5, X, Y, Z), dtype=np.float32)
    taken = np.zeros((X, Y, Z), dtype=bool)
    placed = 0
    for _ in range(n_blobs * 20):
        if placed >= n_blobs:
            break
        r = np.array([rng.integers(4, rmax[0] + 1), rng.integers(4, rmax[1] + 1), rng.integers(2, rmax[2] + 1)])
        lo = np.minimum(np.array(margin) + r, np.array(shape) // 2)
        hi = np.maximum(np.array(shape) - lo, lo + 1)
        c = np.array([rng.integers(lo[k], hi[k]) for k in range(3)])
        a = np.maximum(c - r - 4, 0)
        b = np.minimum(c + r + 5, shape)
        sub = tuple(slice(a[k], b[k]) for k in range(3))
        if taken[sub].any():
            continue
        gx, gy, gz = np.meshgrid(*[np.arange(a[k], b[k]) for k in range(3)], indexing="ij")
        inside = ((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0
        taken[sub] = True
        o = out[(slice(None),) + sub]
        o[4][inside] = 0.95
        for k, gg in enumerate((gx, gy, gz)):
            o[k][inside] = np.clip((c[k] - gg) / scale[k], -1, 1)[inside]
        core = (gx - c[0]) ** 2 + (gy - c[1]) ** 2 + (gz - c[2]) ** 2 <= 2
        o[3][core & inside] = 0.92
        placed += 1
    if noise:
        out += (rng.random(out.shape, dtype=np.float32) - 0.5) * noise
    return torch.from_numpy(out).to(dtype), placed
