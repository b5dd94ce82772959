"""Validation metrics (reference: skoots/validate/lib.py:170-232,358-438).

``mask_iou`` is the heavy part -- the reference loops over every ground-truth instance in Python and forms
full-volume ``logical_and`` / ``logical_or`` per touching pair; here one kernel pass builds the (gt, pred)
contingency table and a second one turns it into the IoU matrix.  The bookkeeping on the small matrix
(``accuracies_from_iou``, ``f1_score``, ``get_segmentation_errors``) is host logic on its values.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from .. import _ffi


def _lut(mask: Tensor) -> Tuple[Tensor, Tensor, int]:
    ids = torch.unique(mask)
    ids = ids[ids > 0]                       # lib.py:201-205: sorted positive ids
    mx = int(ids.max().item()) if ids.numel() else 0
    lut = torch.zeros(mx + 1, dtype=torch.int32, device=mask.device)
    if ids.numel():
        lut[ids.long()] = torch.arange(1, ids.numel() + 1, dtype=torch.int32, device=mask.device)
    return ids, lut, mx


def mask_iou(gt: Tensor, pred: Tensor) -> Tensor:
    """(N, M) fp32 IoU of every ground-truth instance against every predicted one (rows / columns in ascending id
    order, 0 where they do not touch) -- skoots/validate/lib.py:190-229."""
    assert gt.shape == pred.shape, "Input tensors must be the same shape"
    assert gt.device == pred.device, "Input tensors must be on the same device"
    a = gt.to(torch.int32).contiguous()
    b = pred.to(torch.int32).contiguous()
    _ffi.require_gpu(a, "gt")
    ids_a, lut_a, max_a = _lut(a)
    ids_b, lut_b, max_b = _lut(b)
    N, M = int(ids_a.numel()), int(ids_b.numel())
    iou = torch.zeros((N, M), dtype=torch.float32, device=a.device)
    ws_bytes = int(_ffi.lib.sk_mask_iou_workspace_bytes(N, M))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
    _ffi.check(_ffi.lib.sk_mask_iou(_ffi.ptr(a), _ffi.ptr(b), a.numel(), _ffi.ptr(lut_a), max_a, N, _ffi.ptr(lut_b), max_b, M,
                                    _ffi.ptr(iou), _ffi.ptr(ws), ws_bytes, _ffi.stream_ptr(a.device)))
    return iou


def accuracies_from_iou(iou: Tensor, thr: float = 0.1) -> Tuple[float, float, float]:
    """(true positives, false positives, false negatives) at an IoU threshold -- lib.py:170-187."""
    n, m = iou.shape
    gt_miss = torch.logical_not(iou.max(dim=1)[0].gt(thr)) if m > 0 else torch.ones(0)
    pred_miss = torch.logical_not(iou.max(dim=0)[0].gt(thr)) if n > 0 else torch.ones(0)
    tp = torch.sum(torch.logical_not(gt_miss))
    return tp.cpu().item(), torch.sum(pred_miss).cpu().item(), torch.sum(gt_miss).cpu().item()


def f1_score(tp, fp, fn):
    """lib.py:358-361."""
    return 2 * tp / (2 * tp + fp + fn)


def get_segmentation_errors(ground_truth: Tensor, predicted: Tensor) -> Tuple[float, float]:
    """(over-, under-segmentation rate): the share of ground-truth (predicted) instances that more than one
    predicted (ground-truth) instance overlaps with IoU > 0.2 -- lib.py:400-438."""
    iou = mask_iou(ground_truth, predicted)
    n, m = iou.shape
    over = (iou.gt(0.2).sum(dim=1) > 1).sum().item() / n
    under = (iou.gt(0.2).sum(dim=0) > 1).sum().item() / m
    return over, under
