"""Epoch-stepped sigma of the embedding loss (reference: skoots/train/sigma.py:10-60)."""
from __future__ import annotations

from typing import Dict, List, Sequence


class Sigma:
    """``sigma(e)`` = initial sigma times the product of every multiplier whose epoch is < e
    (the reference seeds the list with multiplier 1 at epoch -1, so ``sigma(0)`` is the
    initial value, sigma.py:36-54).  Host floats: the loss kernel takes sigma by value."""

    def __init__(self, adjustments: List[Dict[str, float]], initial_sigma: Sequence[float] = (0.1, 0.1, 0.8)):
        self.adjustments = list(adjustments)
        self.initial_sigma = [float(v) for v in initial_sigma]

    def __call__(self, e: int) -> List[float]:
        m = 1.0
        for d in self.adjustments:
            if d["epoch"] < e:
                m *= float(d["multiplier"])
        return [s * m for s in self.initial_sigma]


def init_sigma(cfg, device=None) -> Sigma:
    """``cfg.TRAIN.INITIAL_SIGMA`` / ``cfg.TRAIN.SIGMA_DECAY`` ([multiplier, epoch] pairs), sigma.py:57-60.
    ``device`` is accepted for signature compatibility with the reference (``init_sigma(cfg, device)``) and not
    needed: the values travel to the kernels as host floats."""
    tr = cfg["TRAIN"] if isinstance(cfg, dict) else cfg.TRAIN
    get = (lambda k: tr[k]) if isinstance(tr, dict) else (lambda k: getattr(tr, k))
    return Sigma([{"multiplier": a, "epoch": b} for a, b in get("SIGMA_DECAY")], get("INITIAL_SIGMA"))
