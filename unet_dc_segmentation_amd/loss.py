"""Fused Focal + Dice loss on the HIP path (autograd aware) -- same semantics as the reference's
``utils/metrics_DC.py::focal_dice_loss`` (:65-73).  ``utils/metrics_DC.focal_dice_loss`` dispatches here
for fp32 probability maps on a HIP device; everything else keeps the PyTorch formulation."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _FocalDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, alpha, gamma, ratio, smooth):
        p = pred.contiguous()
        t = target.contiguous().to(torch.float32)
        nimg, hw = p.shape[0] * p.shape[1], p.shape[2] * p.shape[3]
        nbytes = _lib.load().unetdc_focal_dice_loss_workspace(nimg, hw)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=p.device)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        coef = torch.empty(nimg * 2, dtype=torch.float32, device=p.device)
        call("unetdc_focal_dice_loss_fwd", p.data_ptr(), t.data_ptr(), loss.data_ptr(), coef.data_ptr(), ws.data_ptr(),
             nbytes, nimg, hw, float(alpha), float(gamma), float(ratio), float(smooth), _stream())
        ctx.save_for_backward(p, t, coef)
        ctx.cfg = (nimg, hw, float(alpha), float(gamma), float(ratio))
        return loss

    @staticmethod
    def backward(ctx, gout):
        p, t, coef = ctx.saved_tensors
        nimg, hw, alpha, gamma, ratio = ctx.cfg
        g = gout.contiguous().to(torch.float32)
        dp = torch.empty_like(p)
        call("unetdc_focal_dice_loss_bwd", p.data_ptr(), t.data_ptr(), coef.data_ptr(), g.data_ptr(), dp.data_ptr(),
             nimg, hw, alpha, gamma, ratio, _stream())
        return dp, None, None, None, None, None


def supported(pred, target):
    return (pred.is_cuda and target.is_cuda and pred.dtype == torch.float32 and pred.dim() == 4
            and pred.shape == target.shape and target.dtype in (torch.float32, torch.float64, torch.float16,
                                                                torch.bfloat16, torch.uint8, torch.bool))


def focal_dice_loss(pred, target, alpha=1.0, gamma=2.0, ratio=0.3, smooth=1e-7):
    return _FocalDice.apply(pred, target, alpha, gamma, ratio, smooth)
