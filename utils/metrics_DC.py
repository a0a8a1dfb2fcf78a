"""Host-side losses and metrics -- same names and semantics as the reference's
``utils/metrics_DC.py`` (north_star keeps the Focal/Dice loss in PyTorch on the host side).

The autograd of :func:`focal_dice_loss` produces dL/dprobs, which is the input of the HIP
backward pass (head backward kernel).  Reference lines: dice_loss :11-17, combined_loss :19-22,
dice_coef :24-29, FocalLoss :31-63, focal_dice_loss :65-73, calculate_metrics :75-85.
calculate_metrics returns the reference's five values (precision, recall, F1, specificity, 2x2 confusion
matrix laid out like sklearn's: [[tn, fp], [fn, tp]]) and is what train_DC_focal.py's final test evaluation prints; the
reference's confusion-matrix PLOT (:87-116) is reporting, out of scope.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _per_map_dice(a, b, smooth):
    inter = (a * b).sum(dim=(2, 3))
    total = a.sum(dim=(2, 3)) + b.sum(dim=(2, 3))
    return (2.0 * inter + smooth) / (total + smooth)


def dice_loss(pred, target, smooth=1e-7):
    """1 - mean over (n, c) of the soft Dice coefficient computed over (H, W)."""
    return 1 - _per_map_dice(pred.contiguous(), target.contiguous(), smooth).mean()


def combined_loss(pred, target):
    """0.5 * BCE + 0.5 * Dice (the plain-UNet criterion of train.py)."""
    return 0.5 * F.binary_cross_entropy(pred, target) + 0.5 * dice_loss(pred, target)


def dice_coef(y_true, y_pred, smooth=1e-7):
    """Hard Dice: predictions are binarised at 0.5 first (a no-op on an already binary mask)."""
    return _per_map_dice(y_true, (y_pred > 0.5).float(), smooth).mean()


class FocalLoss(nn.Module):
    """Binary focal loss on probabilities: alpha * (1 - pt)^gamma * bce with pt = exp(-bce)."""

    def __init__(self, alpha=1.0, gamma=2.0, reduction="mean"):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def forward(self, inputs, targets):
        bce = F.binary_cross_entropy(inputs, targets, reduction="none")
        pt = torch.exp(-bce)
        loss = self.alpha * (1 - pt) ** self.gamma * bce
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


def focal_dice_loss(pred, target, alpha=1.0, gamma=2.0, ratio=0.3):
    """ratio * focal + (1 - ratio) * dice; train_DC_focal.py:222 uses (1.0, 2.0, 0.3).

    fp32 probability maps on a HIP device take the fused kernels of
    ``unet_dc_segmentation_amd/csrc/loss.hip`` (same arithmetic, three launches instead of ~25 ATen
    ones); set UNETDC_FUSED_LOSS=0 to force the PyTorch formulation."""
    if pred.is_cuda and os.environ.get("UNETDC_FUSED_LOSS", "1") != "0":
        from unet_dc_segmentation_amd import loss as fused
        if fused.supported(pred, target):
            return fused.focal_dice_loss(pred, target, alpha, gamma, ratio)
    fl = FocalLoss(alpha=alpha, gamma=gamma, reduction="mean")(pred, target)
    return ratio * fl + (1 - ratio) * dice_loss(pred, target)


def calculate_metrics(y_true, y_pred):
    """precision / recall / F1 / specificity of the 0.3-thresholded prediction (zero_division=1) and the
    2x2 confusion matrix [[tn, fp], [fn, tp]] (reference :75-85 via sklearn; same numbers, no sklearn needed)."""
    yp = (y_pred > 0.3).reshape(-1).cpu().numpy().astype(bool)
    yt = (y_true.reshape(-1).cpu().numpy() > 0.5)
    tp = int(np.sum(yp & yt)); fp = int(np.sum(yp & ~yt))
    fn = int(np.sum(~yp & yt)); tn = int(np.sum(~yp & ~yt))
    return metrics_from_counts(tn, fp, fn, tp)


def metrics_from_counts(tn, fp, fn, tp):
    """calculate_metrics() from the four cells of the confusion matrix (callers that counted on the device, e.g. the final test
    evaluation of train_DC_focal.py, need no per-pixel label arrays on the host)."""
    tn, fp, fn, tp = int(tn), int(fp), int(fn), int(tp)
    precision = tp / (tp + fp) if tp + fp > 0 else 1.0
    recall = tp / (tp + fn) if tp + fn > 0 else 1.0
    # sklearn's f1_score(zero_division=1): 1.0 when there are no positives at all, 0.0 when only p + r == 0
    if tp + fp + fn == 0:
        f1 = 1.0
    else:
        f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0
    specificity = tn / (tn + fp) if tn + fp > 0 else 0
    conf_matrix = np.array([[tn, fp], [fn, tp]], dtype=np.int64)
    return precision, recall, f1, specificity, conf_matrix
