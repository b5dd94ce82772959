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


# ------------------------------------------------------------------------------------------------------------------
# The same step with the HIP fast path's 16-bit STORAGE restated in torch (test infrastructure, like everything here):
# separates kernel error from dtype error.  skoots_amd.train's "bf16" / "mixed" step keeps in 16 bits exactly these
# tensors (skoots_amd/train/engine.py): the image operand of the stem, every GroupNorm block's conv weights as the
# matrix cores see them (fp32 masters, re-packed each use; the stem's stay exact), its RAW conv output, its activation,
# the gradient w.r.t. its conv output (the GroupNorm backward writes it 16-bit, power-of-two scaled) and the data
# gradient every conv hands to its producer; accumulation, GroupNorm statistics, loss, heads and AdamW are fp32.
# ------------------------------------------------------------------------------------------------------------------
class _Round(torch.autograd.Function):
    """forward: round to ``dtype`` and back to fp32; backward: straight through."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundGrad(torch.autograd.Function):
    """forward: identity; backward: the gradient is stored in ``dtype`` (a power-of-two scale does not change a bf16
    rounding, and the fp16 step picks its scale so that nothing under- or overflows)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).float(), None


def forward_16bit_storage(model: torch.nn.Module, images: Tensor, dtype=torch.bfloat16) -> Tensor:
    """oracle/unet_spec.UNetSpec's graph with the storage roundings of the HIP training step (autograd-capable)."""
    import torch.nn.functional as F
    from .unet_spec import GN_EPS, GN_GROUPS
    rnd = lambda t: _Round.apply(t, dtype)          # noqa: E731
    rgrad = lambda t: _RoundGrad.apply(t, dtype)    # noqa: E731

    def block(m, srcs, stem=False):
        x = torch.cat([rgrad(s) for s in srcs], dim=1) if not stem else srcs[0]   # a conv's data gradient: 16-bit per source
        w = m.conv.weight if stem else rnd(m.conv.weight)
        y = F.conv3d(x, w, m.conv.bias, stride=m.conv.stride, padding=m.conv.padding)   # fp32 accumulate
        y = rgrad(y)                                   # dL/dy as the GroupNorm backward stores it
        B, C = y.shape[:2]
        g = y.reshape(B, GN_GROUPS, -1)                # statistics of the fp32 accumulators
        mu = g.mean(-1, keepdim=True)
        var = (g * g).mean(-1, keepdim=True) - mu * mu
        rstd = 1.0 / (var.clamp_min(0) + GN_EPS).sqrt()
        y16 = rnd(y).reshape(B, GN_GROUPS, -1)         # the raw output is kept 16-bit: x_hat comes from it
        xh = ((y16 - mu) * rstd).reshape(y.shape)
        z = F.silu(xh * m.norm.weight.reshape(1, C, 1, 1, 1) + m.norm.bias.reshape(1, C, 1, 1, 1))
        return rnd(z)

    t = rnd(images)                                    # the stem's image operand
    for i, m in enumerate(model.enc0):
        t = block(m, [t], stem=(i == 0))
    s0 = t
    t = block(model.down0, [t])
    for m in model.enc1:
        t = block(m, [t])
    s1 = t
    t = block(model.down1, [t])
    for m in model.mid:
        t = block(m, [t])
    t = F.interpolate(block(model.red1, [t]), size=s1.shape[2:], mode="nearest")
    t = block(model.dec1[0], [s1, t])
    for m in model.dec1[1:]:
        t = block(m, [t])
    t = F.interpolate(block(model.red0, [t]), size=s0.shape[2:], mode="nearest")
    t = block(model.dec0[0], [s0, t])
    for m in model.dec0[1:]:
        t = block(m, [t])
    y = model.heads(t)                                 # fp32 weights on the 16-bit activation, fp32 logits and gradient
    return torch.cat([torch.tanh(y[:, 0:3]), torch.sigmoid(y[:, 3:5])], dim=1)


def train_step_16bit_storage(model: torch.nn.Module, optimizer, images: Tensor, masks: Tensor, skele_masks: Tensor,
                             baked: Tensor, sigma: Tensor, scale: Tensor, dtype=torch.bfloat16):
    """:func:`train_step` on :func:`forward_16bit_storage`: what the HIP "bf16" step computes up to summation order."""
    optimizer.zero_grad(set_to_none=True)
    out = forward_16bit_storage(model, images, dtype)
    le, lp, ls, loss = step_loss(out, masks, skele_masks, baked, sigma, scale)
    loss.backward()
    optimizer.step()
    return torch.stack([le, lp, ls, loss]).detach()
