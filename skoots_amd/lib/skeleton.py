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


def average_baked_skeletons(baked_skeleton: Tensor, kernel_size: int = 3) -> Tensor:
    """Smooth a baked skeleton: per channel the sum of the zero-padded 3x3x3 neighbourhood over the number of its
    entries > 0 (skoots/lib/skeleton.py:18-48).  (B, 3, X, Y, Z) fp32 -> same shape."""
    if kernel_size != 3:
        raise ValueError("only the reference's kernel_size = 3 is built")
    x = baked_skeleton.float().contiguous()
    _ffi.require_gpu(x, "baked_skeleton")
    if x.ndim != 5:
        raise ValueError("baked_skeleton must be (B, C, X, Y, Z)")
    b, c, X, Y, Z = x.shape
    out = torch.empty_like(x)
    _ffi.check(_ffi.lib.sk_average_baked_skeletons(_ffi.ptr(x), _ffi.ptr(out), b * c, X, Y, Z, _ffi.stream_ptr(x.device)))
    return out


def bake_skeleton(masks: Tensor, skeletons, anisotropy=(1.0, 1.0, 1.0), average: bool = True, device=None,
                  return_distance: bool = False):
    """For each voxel of instance k: the coordinates of the closest point of skeleton k (training target of the
    embedding loss).  Reference: skoots/lib/skeleton.py:448-528 (GPU path: Triton kernel :51-251).

    masks (X, Y, Z) or (1, X, Y, Z) integer ids on the GPU; skeletons: dict id -> (N, 3) voxel coordinates;
    returns (3, X, Y, Z) fp32 (+ the (1, X, Y, Z) distance map with ``return_distance``).  A ``-1`` key in
    ``skeletons`` returns zeros, as in the reference (:487-492).  Among equidistant skeleton points the first one
    wins (exact distances); the reference's winner depends on cdist's rounding."""
    import numpy as np
    if masks.ndim == 4 and masks.shape[0] == 1:
        masks = masks[0]
    if masks.ndim != 3:
        raise ValueError(f"masks must have have 3 dimensions, not shape: {tuple(masks.shape)}")
    _ffi.require_gpu(masks.contiguous(), "masks")
    dev = masks.device
    X, Y, Z = masks.shape
    if -1 in skeletons:
        baked = torch.zeros((3, X, Y, Z), dtype=torch.float32, device=dev)
        return (baked, torch.zeros((1, X, Y, Z), dtype=torch.float32, device=dev)) if return_distance else baked
    if len(anisotropy) != 3:
        raise ValueError("anisotropy should have 3 values")
    m = masks.to(torch.int32).contiguous()
    ids = sorted(int(k) for k in skeletons)
    pts = [torch.as_tensor(skeletons[k]).reshape(-1, 3).float().cpu() for k in ids]
    offs = np.concatenate([[0], np.cumsum([p.shape[0] for p in pts])]).astype(np.int32)
    ids_d = torch.tensor(ids, dtype=torch.int32, device=dev)
    offs_d = torch.from_numpy(offs).to(dev)
    pts_d = (torch.cat(pts) if pts else torch.zeros((0, 3))).contiguous().to(dev)
    baked = torch.empty((3, X, Y, Z), dtype=torch.float32, device=dev)
    dist = torch.empty((1, X, Y, Z), dtype=torch.float32, device=dev) if return_distance else None
    _ffi.check(_ffi.lib.sk_bake_skeleton(_ffi.ptr(m), _ffi.ptr(ids_d), _ffi.ptr(offs_d), _ffi.ptr(pts_d), len(ids), X, Y, Z,
                                         _ffi.float_array([float(a) for a in anisotropy]), _ffi.ptr(baked), _ffi.ptr(dist),
                                         _ffi.stream_ptr(dev)))
    if average:
        baked = average_baked_skeletons(baked.unsqueeze(0)).squeeze(0)
    return (baked, dist) if return_distance else baked
