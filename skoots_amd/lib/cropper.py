"""Tile / crop grid of the reference's sliding window.

Mirrors ``skoots/lib/cropper.py`` (``get_total_num_crops`` :8-55, ``crops`` :58-144):
loop positions advance by ``crop - 2*overlap`` while ``< dim`` and the emitted origin
is clamped to ``dim - crop``; x outer, y, z inner.  Pure host logic (integers).

On top of the reference's generator this module derives what the device path needs:
the *distinct* origins (clamped duplicates recompute identical tiles in the
reference) and, per axis, which origin writes each coordinate last (the reference
scatters tile interiors in generator order, last writer wins).
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


def _check(spatial: Sequence[int], crop: Sequence[int], overlap: Sequence[int]) -> None:
    assert len(spatial) == len(crop) == len(overlap) == 3, (
        f"Image Shape must equal the shape of the crop.\n{spatial=}, {crop=}{overlap=}")
    for c, o, name in zip(crop, overlap, "xyz"):
        assert c - 2 * o > 0, (
            f"Overlap in {name} dimension cannot be equal to or larger than half the crop size... "
            f"{o * 2=} >= {c}")


def clamp_crop_(crop_size: List[int], spatial: Sequence[int]) -> List[int]:
    """cropper.py:13-16 / 81-84: the caller's list is clamped IN PLACE (eval() relies on it)."""
    for i in range(3):
        if not crop_size[i] < spatial[i]:
            crop_size[i] = int(spatial[i])
    return crop_size


def axis_positions(dim: int, crop: int, overlap: int) -> List[int]:
    """Clamped origins along one axis, one entry per loop position (duplicates kept)."""
    out, v = [], 0
    while v < dim:
        out.append(v if v + crop <= dim else dim - crop)
        v += crop - 2 * overlap
    return out


def crop_origins(spatial: Sequence[int], crop_size: List[int], overlap: Sequence[int]
                 ) -> List[Tuple[int, int, int]]:
    """Every origin the reference generator emits, in its order (crop_size clamped in place)."""
    clamp_crop_(crop_size, spatial)
    _check(spatial, crop_size, overlap)
    ax = [axis_positions(d, c, o) for d, c, o in zip(spatial, crop_size, overlap)]
    return [(x, y, z) for x in ax[0] for y in ax[1] for z in ax[2]]


def get_total_num_crops(image_shape, crop_size: List[int], overlap) -> int:
    """Number of crops ``crops`` will yield for an image of shape (C, X, Y, Z)."""
    overlap = (0, 0, 0) if overlap is None else overlap
    return len(crop_origins(list(image_shape)[1:], crop_size, overlap))


def crops(image: Tensor, crop_size: List[int], overlap: Optional[Tuple[int, int, int]] = (0, 0, 0),
          device="cpu") -> Iterator[Tuple[Tensor, List[int]]]:
    """Generator of (1, C, w, h, d) crops and their [x, y, z] origins (reference order).

    Kept for API compatibility; the device pipeline never materialises crops -- it
    reads tiles in place from the HBM-resident volume.
    """
    for (x, y, z) in crop_origins(list(image.shape)[1:], crop_size, overlap):
        c = image[:, x:x + crop_size[0], y:y + crop_size[1], z:z + crop_size[2]]
        c = torch.from_numpy(c) if isinstance(c, np.ndarray) else c
        yield c.unsqueeze(0).to(device, non_blocking=True), [x, y, z]


def distinct_origins(spatial: Sequence[int], crop_size: List[int], overlap: Sequence[int]
                     ) -> List[Tuple[int, int, int]]:
    """Origins with clamped duplicates removed, keeping the reference's relative order.

    A duplicate origin is the same input window -> the same output; dropping all but
    the last occurrence leaves every voxel's last writer unchanged.
    """
    clamp_crop_(crop_size, spatial)
    _check(spatial, crop_size, overlap)
    ax = []
    for d, c, o in zip(spatial, crop_size, overlap):
        pos = axis_positions(d, c, o)
        ax.append([p for i, p in enumerate(pos) if p not in pos[i + 1:]])
    return [(x, y, z) for x in ax[0] for y in ax[1] for z in ax[2]]


def owner_table(dim: int, crop: int, overlap: int) -> np.ndarray:
    """owner[v] = origin of the crop whose interior [o+ov, o+crop-ov) writes coordinate v
    last along this axis, or -1 where no interior reaches (the never-written frame,
    SURVEY.md section 0.3)."""
    own = np.full(dim, -1, dtype=np.int32)
    for o in axis_positions(dim, crop, overlap):
        own[o + overlap:o + crop - overlap] = o
    return own
