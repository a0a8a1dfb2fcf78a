"""Seeded golden-fixture recipe shared by tools/make_goldens.py and tests/ -- TEST INFRASTRUCTURE.

A random-init network in eval() mode outputs probabilities in [0.47, 0.48]: every pixel is
above the 0.3 threshold (train_DC_focal.py:259, quantify_droplets_batch.py:56) and "mask
bit-exact" would be vacuous.  ``perturb_bn`` gives BatchNorm non-trivial
running statistics / affine parameters from a seeded generator, and the generator script stores
the calibrated ``out_conv.bias`` that centres the pre-sigmoid map on ln(0.3/0.7), so that about
half of the pixels fall on each side of the threshold.
"""
from __future__ import annotations

import math

import numpy as np
import torch

LOGIT_THRESH = math.log(0.3 / 0.7)      # sigmoid(z) > 0.3  <=>  z > ln(3/7)


def perturb_bn(state_dict, seed):
    """In-place, deterministic BN perturbation (SURVEY.md section 8c 'golden recipe')."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(state_dict):
        v = state_dict[k]
        if k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.05 + 0.5 * torch.rand(v.shape, generator=g))
        elif ".1." in k or ".4." in k:
            if k.endswith("weight"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    return state_dict


def seeded_input(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g)


def seeded_target(seed, shape, frac=0.3):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) < frac).float()


def sd_checksums(state_dict):
    """Per-key (sum, abs-sum) in float64: lets a test prove two state dicts are identical
    without committing 124 MB of weights."""
    keys = sorted(state_dict)
    out = np.zeros((len(keys), 2))
    for i, k in enumerate(keys):
        v = state_dict[k].detach().double()
        out[i] = (float(v.sum()), float(v.abs().sum()))
    return keys, out


def grad_probe(t, n=64):
    """Deterministic strided sample of a tensor (flattened), used to pin gradients."""
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].clone()
