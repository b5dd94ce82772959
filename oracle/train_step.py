"""CPU restatement of the reference's training step (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The loss pieces restate the reference's own functions and are pinned by
tests/golden/loss.npz (G8, generated from the reference's ``tversky`` and
``baked_embed_to_prob``); the network body is ``oracle/unet_spec.py`` (parity
unpinned, see there); gradients come from torch autograd and the update from
``torch.optim.AdamW`` -- the optimizer the reference's defaults select
(skoots/config.py:96-101, skoots/train/engine.py:281-285).
"""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

from .pipeline import vector_to_embedding


def tversky(pred: Tensor, gt: Tensor, alpha: float, beta: float, eps: float) -> Tensor:
    """skoots/train/loss.py:95-212.  pred, gt (B, 1, X, Y, Z); per sample the ground truth is expanded
    into one binary mask per non-zero id (:175-182) and broadcast against pred; batch mean (:155)."""
    out = []
    for b in range(pred.shape[0]):
        p, g = pred[b], gt[b]
        ids = torch.unique(g)
        ids = ids[ids != 0]
        nd = torch.stack([(g == i).float().squeeze(0) for i in ids]) if len(ids) else torch.zeros((0,) + tuple(g.shape[1:]))
        tp = (p * nd).sum()                                   # :189
        fp = (torch.logical_not(nd) * p).sum().add(1e-10).mul(alpha)  # :190-192
        fn = ((1 - p) * nd).sum() * beta                      # :193
        out.append(1 - (tp + eps) / (tp + fp + fn + eps))     # :205-209
    return torch.stack(out).mean()


def baked_embed_to_prob(embedding: Tensor, baked: Tensor, sigma: Tensor, eps: float = 1e-16) -> Tensor:
    """skoots/lib/embedding_to_prob.py:38-49."""
    s = (sigma + eps).pow(2).mul(2).mul(-1).reshape(1, -1, 1, 1, 1)
    return torch.exp(((embedding - baked).pow(2) / s).sum(dim=1, keepdim=True))


def step_loss(out: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor, sigma: Tensor, scale: Tensor,
              loss_embed=(0.25, 0.75, 1e-8), loss_prob=(0.5, 0.5, 1e-8), loss_skele=(0.5, 1.5, 1e-8),
              weights: Sequence[float] = (1.0, 1.0, 1.0)):
    """engine.py:461-493 on the model output ``out`` (B, 5, X, Y, Z).  Returns (embed, prob, skeleton, total)."""
    prob, vec, sk = out[:, [-1]], out[:, 0:3], out[:, [-2]]
    emb = torch.cat([vector_to_embedding(scale, vec[b:b + 1]) for b in range(out.shape[0])])  # N = 1: index + v*scale
    pe = baked_embed_to_prob(emb, baked, sigma)
    fg = masks.gt(0).float()
    le = tversky(pe, fg, *loss_embed)
    lp = tversky(prob, fg, *loss_prob)
    ls = tversky(sk, skele_masks.gt(0).float(), *loss_skele)
    return le, lp, ls, weights[0] * le + weights[1] * lp + weights[2] * ls


def make_optimizer(model: torch.nn.Module, lr: float = 5e-4, weight_decay: float = 1e-6) -> torch.optim.Optimizer:
    return torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)


def train_step(model: torch.nn.Module, optimizer, images: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor,
               sigma: Tensor, scale: Tensor):
    """engine.py:456-499: zero_grad -> forward -> loss -> backward -> step."""
    optimizer.zero_grad(set_to_none=True)
    out = model(images)
    le, lp, ls, loss = step_loss(out, masks, skele_masks, baked, sigma, scale)
    loss.backward()
    optimizer.step()
    return torch.stack([le, lp, ls, loss]).detach()
