"""``skoots.lib.vector_to_embedding`` on the MI355X (reference:
skoots/lib/vector_to_embedding.py:135-174 ``vector_to_embedding``, :79-132 ``_vec2embed3D``)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import Tensor

from .. import _ffi


def step_scales(scale, n: int, decay: float) -> List[float]:
    """Per-iteration fp32 scale rows the kernels consume.

    Row 0 is ``scale.float()``; row i is ``float32(decay**i) * float32(scale)`` -- the
    reference multiplies a python double ``scale *= decay`` into the fp32 tensor
    (vector_to_embedding.py:113-115), which torch evaluates as an fp32 product.
    """
    num = np.asarray([float(s) for s in scale], dtype=np.float32)
    assert num.shape == (3,), "scale must have three entries"
    rows, strength = [num.copy()], 1.0
    for _ in range(n - 1):
        strength *= decay
        rows.append(np.float32(strength) * num)
    return [float(v) for r in rows for v in r]


def vector_to_embedding(scale: Tensor, vector: Tensor, N: int = 1, decay: float = 1.0) -> Tensor:
    """Spatial embedding ``phi = v*s + index`` followed by N-1 follow steps.

    Shapes: scale (3), vector (1, 3, X, Y, Z) fp16/fp32 on the GPU -> (1, 3, X, Y, Z) fp32.
    Only the 3-D path is on the eval hot path; 2-D inputs are rejected.
    """
    if vector.ndim != 5:
        raise ValueError("skoots_amd.vector_to_embedding implements the 3-D (5-D tensor) path only")
    if vector.shape[0] != 1 or vector.shape[1] != 3:
        raise ValueError(f"vector must be (1, 3, X, Y, Z), got {tuple(vector.shape)}")
    _ffi.require_gpu(vector, "vector")
    if vector.dtype not in (torch.float16, torch.float32):
        vector = vector.float()
    _, _, w, h, d = vector.shape
    sc = step_scales(scale.tolist() if isinstance(scale, Tensor) else scale, N, decay)
    out = torch.empty((1, 3, w, h, d), dtype=torch.float32, device=vector.device)
    _ffi.check(_ffi.lib.sk_vector_to_embedding(
        _ffi.ptr(vector), _ffi.dtype_code(vector), _ffi.ptr(out), w, h, d,
        _ffi.float_array(sc), N, _ffi.stream_ptr(vector.device)))
    return out
