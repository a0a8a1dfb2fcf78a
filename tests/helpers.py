"""Shared test helpers: rebuild golden networks from their seed, tolerances, comparisons."""
import os

import numpy as np
import torch

from oracle import recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

E2E = {  # tag -> (module path, class name)
    "dc_c1": ("models.model_2", "UNetDC"),
    "dc_c3": ("models.model_2", "UNetDC"),
    "plain_c3": ("models.model", "UNet"),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def build_model(tag, stage="init"):
    """Rebuild the network a golden was generated from.  stage: 'init' (seeded default init),
    'eval' (BN perturbed + calibrated out_conv.bias), 'train' (same weights as 'eval')."""
    import importlib
    g = load_golden("e2e_" + tag)
    mod, cls = E2E[tag]
    klass = getattr(importlib.import_module(mod), cls)
    seed, cin = int(g["seed"]), int(g["cin"])
    torch.manual_seed(seed)
    model = klass(in_channels=cin, out_channels=1)
    if stage != "init":
        recipe.perturb_bn(model.state_dict(), seed + 1)
        with torch.no_grad():
            model.out_conv.bias.copy_(torch.from_numpy(g["out_conv_bias"]))
    return model, g


def sd_numpy(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
