"""Minimal zarr v2 directory store (uncompressed), enough for the two intermediates ``eval()`` leaves next to the
image -- ``<base>_skoots_skeleton.zarr`` and ``<base>_skoots_vectors.zarr`` (skoots/lib/eval.py:101-111,160-176).

The ``zarr`` package is not in this image.  What is written here follows the published v2 layout (``.zarray`` JSON +
one raw C-order file per chunk, edge chunks padded to the full chunk shape, ``compressor: null``), so the real
package opens it; ``load`` reads back stores of that kind only (it refuses compressed ones)."""
from __future__ import annotations

import itertools
import json
import os
import shutil
from typing import Optional, Sequence

import numpy as np


def _dtype_str(dt: np.dtype) -> str:
    dt = np.dtype(dt)
    return dt.str if dt.itemsize > 1 else "|" + dt.str[1:]


def save(path: str, arr: np.ndarray, chunks: Optional[Sequence[int]] = None) -> None:
    arr = np.ascontiguousarray(arr)
    if chunks is None:
        chunks = [min(s, c) for s, c in zip(arr.shape, (1, 256, 256, 64)[-arr.ndim:])]
    chunks = [max(1, int(c)) for c in chunks]
    if os.path.isdir(path):
        shutil.rmtree(path)
    os.makedirs(path)
    meta = {"zarr_format": 2, "shape": list(arr.shape), "chunks": chunks, "dtype": _dtype_str(arr.dtype),
            "compressor": None, "fill_value": 0, "order": "C", "filters": None}
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f, indent=2)
    grid = [range((s + c - 1) // c) for s, c in zip(arr.shape, chunks)]
    for idx in itertools.product(*grid):
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block = np.zeros(chunks, dtype=arr.dtype)
        part = arr[sl]
        block[tuple(slice(0, n) for n in part.shape)] = part
        block.tofile(os.path.join(path, ".".join(str(i) for i in idx)))


def load(path: str) -> np.ndarray:
    with open(os.path.join(path, ".zarray")) as f:
        meta = json.load(f)
    if meta.get("zarr_format") != 2 or meta.get("compressor") is not None or meta.get("filters"):
        raise RuntimeError(f"{path}: only uncompressed zarr v2 stores written by skoots_amd can be read here")
    if meta.get("order", "C") != "C":
        raise RuntimeError(f"{path}: only C-order stores are supported")
    shape, chunks, dt = meta["shape"], meta["chunks"], np.dtype(meta["dtype"])
    out = np.full(shape, meta.get("fill_value") or 0, dtype=dt)
    grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)]
    for idx in itertools.product(*grid):
        fn = os.path.join(path, ".".join(str(i) for i in idx))
        if not os.path.exists(fn):
            continue  # missing chunk = fill value
        block = np.fromfile(fn, dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out
