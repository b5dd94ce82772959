"""Embedding -> probability against a baked skeleton (reference: skoots/lib/embedding_to_prob.py)."""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

from .. import _ffi


def baked_embed_to_prob(embedding: Tensor, baked_skeletons: Tensor, sigma: Sequence[float], eps: float = 1e-16) -> Tensor:
    """``exp(sum_k (E_k - S_k)^2 / (-2 (sigma_k + eps)^2))`` (embedding_to_prob.py:5-51).

    embedding, baked_skeletons: (B, 3, X, Y, Z) fp32 on the GPU; sigma: 3 values (host).
    Returns (B, 1, X, Y, Z).  Forward value only -- the training step differentiates the fused
    loss kernel instead (skoots_amd/train/engine.py)."""
    if embedding.ndim != 5 or embedding.shape[1] != 3 or embedding.shape != baked_skeletons.shape:
        raise ValueError("baked_embed_to_prob: embedding and baked_skeletons must both be (B, 3, X, Y, Z)")
    e = embedding.float().contiguous()
    s = baked_skeletons.float().contiguous()
    _ffi.require_gpu(e, "embedding")
    _ffi.require_gpu(s, "baked_skeletons")
    B, _, X, Y, Z = e.shape
    out = torch.empty((B, 1, X, Y, Z), dtype=torch.float32, device=e.device)
    sig = _ffi.float_array([float(v) for v in (sigma.tolist() if isinstance(sigma, Tensor) else sigma)])
    _ffi.check(_ffi.lib.sk_baked_embed_to_prob(_ffi.ptr(e), _ffi.ptr(s), _ffi.ptr(out), B, X * Y * Z, sig, float(eps),
                                               _ffi.stream_ptr(e.device)))
    return out
