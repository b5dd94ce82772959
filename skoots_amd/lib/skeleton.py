"""``skoots.lib.skeleton.index_skeleton_by_embed`` on the MI355X
(reference: skoots/lib/skeleton.py:656-695)."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _ffi


def index_skeleton_by_embed(skeleton: Tensor, embed: Tensor) -> Tensor:
    """Instance mask by indexing the labelled skeleton with an embedding.

    Shapes: skeleton (1, 1, X, Y, Z) int16/int32, embed (1, 3, x, y, z) fp32 -> (1, 1, x, y, z) int32.
    """
    assert embed.device == skeleton.device, "embed and skeleton must be on same device"
    assert embed.ndim == 5 and skeleton.ndim == 5, "Embed and skeleton must be a 5D tensor"
    _ffi.require_gpu(embed, "embed")
    _ffi.require_gpu(skeleton, "skeleton")
    if embed.dtype != torch.float32:
        embed = embed.float()
    if skeleton.dtype not in (torch.int16, torch.int32):
        skeleton = skeleton.to(torch.int32)
    _, c, x, y, z = embed.shape
    assert c == 3 and embed.shape[0] == 1
    out = torch.empty((1, 1, x, y, z), dtype=torch.int32, device=embed.device)
    _ffi.check(_ffi.lib.sk_index_skeleton_by_embed(
        _ffi.ptr(skeleton), _ffi.dtype_code(skeleton), skeleton.shape[2], skeleton.shape[3],
        skeleton.shape[4], _ffi.ptr(embed), x * y * z, _ffi.ptr(out), _ffi.stream_ptr(embed.device)))
    return out
