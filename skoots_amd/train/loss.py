"""Loss functions of the training step (reference: skoots/train/loss.py).

Only ``tversky`` is built: it is the one the reference's defaults use for all three terms
(skoots/config.py:49-59).  The training step itself uses the fused kernel behind
``engine.TrainStep`` (three terms + their gradient in two passes over the logits); this is
the reference's stand-alone callable with the same constructor and call signature.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _ffi


class tversky:
    """``tversky(alpha, beta, eps)(predicted, ground_truth)`` (train/loss.py:95-212).

    predicted (B, 1, X, Y, Z) probabilities; ground_truth (B, 1, X, Y, Z), 0 = background (the
    reference's callers pass ``masks.gt(0).float()``, engine.py:468).  Per sample
    1 - (TP + eps) / (TP + alpha (FP + 1e-10) + beta FN + eps), then the batch mean.  Value only.
    """

    def __init__(self, alpha: float, beta: float, eps: float):
        self.alpha, self.beta, self.eps = float(alpha), float(beta), float(eps)

    def __call__(self, predicted: Tensor, ground_truth: Tensor) -> Tensor:
        if predicted.ndim != 5 or predicted.shape[1] != 1 or predicted.shape != ground_truth.shape:
            raise ValueError("tversky: predicted and ground_truth must both be (B, 1, X, Y, Z)")
        p = predicted.float().contiguous()
        g = ground_truth.float().contiguous()
        _ffi.require_gpu(p, "predicted")
        _ffi.require_gpu(g, "ground_truth")
        B = p.shape[0]
        n = p[0].numel()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        ws = torch.empty(int(_ffi.lib.sk_train_loss_workspace_floats(B, n)), dtype=torch.float32, device=p.device)
        _ffi.check(_ffi.lib.sk_train_tversky(_ffi.ptr(p), _ffi.ptr(g), B, n, self.alpha, self.beta, self.eps,
                                             _ffi.ptr(loss), _ffi.ptr(ws), _ffi.stream_ptr(p.device)))
        return loss[0]

    def __repr__(self):
        return f"LossFn[name=tversky, alpha={self.alpha}, beta={self.beta}, eps={self.eps}"
