"""Minimal zarr v2 directory store, enough for the two intermediates ``eval()`` leaves next to the image --
``<base>_skoots_skeleton.zarr`` and ``<base>_skoots_vectors.zarr`` (skoots/lib/eval.py:101-111,160-176).

The ``zarr`` / ``numcodecs`` packages are not in this image.  What is written here follows the published v2 layout
(``.zarray`` JSON + one C-order file per chunk, edge chunks padded to the full chunk shape), so the real package opens
it.  Chunks are compressed with the numcodecs ``zlib`` codec (``{"id": "zlib", "level": 1}`` -- the stdlib's zlib; the
reference's stores use zarr's default Blosc codec, which needs the absent ``numcodecs``) or stored raw
(``compressor=None``); chunks that hold only the fill value are not written, as zarr does.  ``load`` reads raw, zlib
and gzip stores and refuses every other codec by name."""
from __future__ import annotations

import gzip
import itertools
import json
import os
import shutil
import zlib
from typing import Optional, Sequence

import numpy as np


def _dtype_str(dt: np.dtype) -> str:
    dt = np.dtype(dt)
    return dt.str if dt.itemsize > 1 else "|" + dt.str[1:]


def save(path: str, arr: np.ndarray, chunks: Optional[Sequence[int]] = None, compressor: Optional[str] = "zlib",
         level: int = 1) -> None:
    arr = np.ascontiguousarray(arr)
    if chunks is None:
        chunks = [min(s, c) for s, c in zip(arr.shape, (1, 256, 256, 64)[-arr.ndim:])]
    chunks = [max(1, int(c)) for c in chunks]
    if os.path.isdir(path):
        shutil.rmtree(path)
    os.makedirs(path)
    if compressor not in (None, "zlib"):
        raise ValueError("compressor must be None or 'zlib'")
    meta = {"zarr_format": 2, "shape": list(arr.shape), "chunks": chunks, "dtype": _dtype_str(arr.dtype),
            "compressor": {"id": "zlib", "level": int(level)} if compressor else None, "fill_value": 0, "order": "C",
            "filters": None}
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f, indent=2)
    grid = [range((s + c - 1) // c) for s, c in zip(arr.shape, chunks)]
    for idx in itertools.product(*grid):
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block = np.zeros(chunks, dtype=arr.dtype)
        part = arr[sl]
        if not part.any():
            continue  # only the fill value: zarr leaves such chunks out (write_empty_chunks=False)
        block[tuple(slice(0, n) for n in part.shape)] = part
        raw = block.tobytes()
        with open(os.path.join(path, ".".join(str(i) for i in idx)), "wb") as f:
            f.write(zlib.compress(raw, int(level)) if compressor else raw)


def load(path: str) -> np.ndarray:
    with open(os.path.join(path, ".zarray")) as f:
        meta = json.load(f)
    comp = meta.get("compressor")
    codec = comp.get("id") if comp else None
    if meta.get("zarr_format") != 2 or meta.get("filters") or codec not in (None, "zlib", "gzip"):
        raise RuntimeError(f"{path}: zarr v2 stores with compressor {codec!r} / filters {meta.get('filters')!r} cannot be "
                           "read here (raw, zlib and gzip only: numcodecs / Blosc are not in this image)")
    if meta.get("order", "C") != "C":
        raise RuntimeError(f"{path}: only C-order stores are supported")
    shape, chunks, dt = meta["shape"], meta["chunks"], np.dtype(meta["dtype"])
    out = np.full(shape, meta.get("fill_value") or 0, dtype=dt)
    grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)]
    for idx in itertools.product(*grid):
        fn = os.path.join(path, ".".join(str(i) for i in idx))
        if not os.path.exists(fn):
            continue  # missing chunk = fill value
        with open(fn, "rb") as f:
            raw = f.read()
        if codec == "zlib":
            raw = zlib.decompress(raw)
        elif codec == "gzip":
            raw = gzip.decompress(raw)
        block = np.frombuffer(raw, dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out
