"""CPU port of the reference path on PyTorch-CPU (ATen/oneDNN) -- TEST INFRASTRUCTURE ONLY.

Same role and same import rules as ``oracle/unetdc_numpy.py`` (only tests, smoke() and the
``cpu_baseline`` leg of bench.py may import this).  All hot-path arithmetic of the reference lives
in PyTorch ATen ("third-party arithmetic"); this module calls the same ATen
ops in the same order through ``torch.nn.functional`` on a plain state-dict, so it is (a) the
fast full-size checker for the HIP path on the GPU box, where /root/reference does not exist,
and (b) the ``"kind": "port"`` CPU baseline timed by bench.py on the host cores.

Pinned against the live reference by tools/make_goldens.py -> tests/golden/*.npz
(tests/test_oracle_golden.py).  Citations are to files under /root/reference.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BLOCKS = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec4", "dec3", "dec2", "dec1")
DILATIONS_DC = {"enc1": 1, "enc2": 2, "enc3": 4, "enc4": 8, "bottleneck": 16,
                "dec4": 1, "dec3": 1, "dec2": 1, "dec1": 1}          # models/model_2.py:10-30
DILATIONS_PLAIN = {b: 1 for b in BLOCKS}                              # models/model.py:25-33


class _StoreBF16(torch.autograd.Function):
    """Round a tensor through bfloat16 storage (straight-through gradient).  Used to emulate the HIP
    path's bf16 configuration -- bf16 weights / conv outputs / activations, fp32 accumulation and
    fp32 BatchNorm arithmetic -- on top of the fp32 ATen ops.  A CPU study shows the
    forward rounding is what separates bf16 from fp32 gradients at random init (cosine ~0.90 in the
    deepest layers); rounding the stored gradients is invisible (cosine 1.0000)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x, emulate_bf16):
    return _StoreBF16.apply(x) if emulate_bf16 else x


def _double_conv(x, sd, name, d, train, momentum=0.1, eps=1e-5, emulate_bf16=False):
    """double_conv (model_2.py:34-54): (Conv3x3 pad=d dil=d -> BN -> ReLU) x 2."""
    for idx in (0, 3):
        w = sd[f"{name}.{idx}.weight"]
        if not (idx == 0 and w.shape[1] < 8):          # the first layer keeps fp32 weights and input
            w = _q(w, emulate_bf16)
        x = _q(F.conv2d(x, w, sd[f"{name}.{idx}.bias"], padding=d, dilation=d), emulate_bf16)
        x = F.batch_norm(x, sd[f"{name}.{idx + 1}.running_mean"], sd[f"{name}.{idx + 1}.running_var"],
                         sd[f"{name}.{idx + 1}.weight"], sd[f"{name}.{idx + 1}.bias"],
                         training=train, momentum=momentum, eps=eps)
        x = _q(F.relu(x), emulate_bf16)
    return x


def unet_forward(x, sd, dilations=None, train=False, return_logits=False, emulate_bf16=False):
    """UNetDC.forward (model_2.py:56-80). ``sd`` maps the 136 reference keys to CPU tensors;
    in train mode the running_mean/var tensors are updated in place like nn.BatchNorm2d does.
    ``emulate_bf16`` rounds weights / conv outputs / activations through bf16 storage (see _StoreBF16)."""
    dil = DILATIONS_DC if dilations is None else dilations
    skips = []
    h = x
    for name in ("enc1", "enc2", "enc3", "enc4"):
        h = _double_conv(h, sd, name, dil[name], train, emulate_bf16=emulate_bf16)
        skips.append(h)
        h = F.max_pool2d(h, 2)
    h = _double_conv(h, sd, "bottleneck", dil["bottleneck"], train, emulate_bf16=emulate_bf16)
    for lvl in (4, 3, 2, 1):
        up = _q(F.conv_transpose2d(h, _q(sd[f"upconv{lvl}.weight"], emulate_bf16), sd[f"upconv{lvl}.bias"], stride=2),
                emulate_bf16)
        h = _double_conv(torch.cat([up, skips[lvl - 1]], dim=1), sd, f"dec{lvl}", dil[f"dec{lvl}"], train,
                         emulate_bf16=emulate_bf16)
    z = F.conv2d(h, sd["out_conv.weight"], sd["out_conv.bias"])
    p = torch.sigmoid(z)
    return (p, z) if return_logits else p


def dice_loss(pred, target, smooth=1e-7):
    """utils/metrics_DC.py:11-17."""
    inter = (pred * target).sum(dim=(2, 3))
    union = pred.sum(dim=(2, 3)) + target.sum(dim=(2, 3))
    return 1 - ((2.0 * inter + smooth) / (union + smooth)).mean()


def focal_dice_loss(pred, target, alpha=1.0, gamma=2.0, ratio=0.3):
    """utils/metrics_DC.py:65-73 with FocalLoss.forward (:43-63), reduction='mean'."""
    bce = F.binary_cross_entropy(pred, target, reduction="none")
    pt = torch.exp(-bce)
    fl = (alpha * (1 - pt) ** gamma * bce).mean()
    return ratio * fl + (1 - ratio) * dice_loss(pred, target)


PARAM_SUFFIXES = ("weight", "bias")


def param_keys(sd):
    """Keys of learnable tensors in reference order (everything except BN buffers)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def train_step_grads(x, target, sd, dilations=None, emulate_bf16=False):
    """fwd (train mode) + focal_dice_loss + backward (train_DC_focal.py:252-254).
    Returns (loss, probs, {key: grad})."""
    keys = param_keys(sd)
    leaf = {k: (sd[k].detach().clone().requires_grad_(True) if k in keys else sd[k]) for k in sd}
    p = unet_forward(x, leaf, dilations, train=True, emulate_bf16=emulate_bf16)
    loss = focal_dice_loss(p, target, 1.0, 2.0, 0.3)     # train_DC_focal.py:222
    gs = torch.autograd.grad(loss, [leaf[k] for k in keys])
    return loss.detach(), p.detach(), dict(zip(keys, gs))
