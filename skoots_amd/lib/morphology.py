"""``skoots.lib.morphology`` dilations on the MI355X
(reference: skoots/lib/morphology.py:155-175 ``binary_dilation``, :178-199 ``binary_dilation_2d``)."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _ffi


def _max_filter(image: Tensor, radius) -> Tensor:
    if image.ndim != 5:
        raise ValueError("image must be (B, C, X, Y, Z)")
    _ffi.require_gpu(image, "image")
    x = image.float() if image.dtype != torch.float32 else image
    out = torch.empty_like(x)
    b, c, w, h, d = x.shape
    for i in range(b * c):
        src = x.view(b * c, w, h, d)[i]
        dst = out.view(b * c, w, h, d)[i]
        _ffi.check(_ffi.lib.sk_max_filter3d(_ffi.ptr(src), _ffi.ptr(dst), w, h, d, *radius,
                                            _ffi.stream_ptr(x.device)))
    return out


def binary_dilation(image: Tensor) -> Tensor:
    """3x3x3 max filter with zero padding on a (B, C, X, Y, Z) tensor."""
    return _max_filter(image, (1, 1, 1))


def binary_dilation_2d(image: Tensor) -> Tensor:
    """3x3x1 max filter with zero padding on a (B, C, X, Y, Z) tensor."""
    return _max_filter(image, (1, 1, 0))
