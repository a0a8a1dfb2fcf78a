#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the LIVE reference (runs only in the build container).

Imports /root/reference/models/{model_2,model}.py and utils/metrics_DC.py (behind an empty
``seaborn`` stub module -- metrics_DC.py:8 imports it only for a plotting helper)
and records input/expected-output vectors.  Nothing of the reference (source, bytecode, pickles)
is written: fixtures are plain arrays.  The 124 MB state dict is NOT stored; fixtures carry
per-key checksums instead, and tests rebuild the weights from the seed with the drop-in module
(identical constructors in identical order => identical RNG stream).

Usage:  python tools/make_goldens.py   (from the repo root)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import recipe  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_modules():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    m2 = _load("ref_model_2", f"{REF}/models/model_2.py")
    m1 = _load("ref_model", f"{REF}/models/model.py")
    met = _load("ref_metrics_DC", f"{REF}/utils/metrics_DC.py")
    return m2, m1, met


def e2e_fixture(cls, cin, seed, tag, met, hw=32):
    torch.manual_seed(seed)
    model = cls(in_channels=cin, out_channels=1)
    keys0, sums0 = recipe.sd_checksums(model.state_dict())
    out = {"seed": np.int64(seed), "cin": np.int64(cin), "init_checksums": sums0}

    # ---- eval-mode forward with the calibrated-mask recipe ---------------------------------
    recipe.perturb_bn(model.state_dict(), seed + 1)
    model.eval()
    x = recipe.seeded_input(seed + 2, (2, cin, hw, hw))
    zs = []
    hook = model.out_conv.register_forward_hook(lambda m, i, o: zs.append(o.detach()))
    with torch.no_grad():
        model(x)
        shift = recipe.LOGIT_THRESH - float(zs[-1].median())
        model.out_conv.bias += shift
        zs.clear()
        probs = model(x)
    hook.remove()
    z = zs[-1]
    mask = (probs > 0.3)
    print(f"[{tag}] eval: mask ones fraction {mask.float().mean():.4f}, "
          f"min |z-thr| {float((z - recipe.LOGIT_THRESH).abs().min()):.2e}")
    out.update(eval_x=x.numpy(), eval_z=z.numpy(), eval_probs=probs.numpy(),
               eval_mask=mask.numpy().astype(np.uint8),
               out_conv_bias=model.out_conv.bias.detach().numpy().copy())
    _, sums1 = recipe.sd_checksums(model.state_dict())
    out["eval_checksums"] = sums1

    # ---- train-mode forward + focal/dice loss + backward -----------------------------------
    model.train()
    xt = recipe.seeded_input(seed + 3, (2, cin, hw, hw))
    tt = recipe.seeded_target(seed + 4, (2, 1, hw, hw))
    model.zero_grad()
    p = model(xt)
    loss = met.focal_dice_loss(p, tt, alpha=1.0, gamma=2.0, ratio=0.3)
    gp, = torch.autograd.grad(loss, p, retain_graph=True)
    loss.backward()
    out.update(train_x=xt.numpy(), train_t=tt.numpy(), train_probs=p.detach().numpy(),
               train_loss=np.float64(loss.item()), train_dprobs=gp.numpy())
    names = [k for k, _ in model.named_parameters()]
    out["grad_norms"] = np.array([float(v.grad.double().norm()) for _, v in model.named_parameters()])
    out["grad_probes"] = np.stack([np.pad(recipe.grad_probe(v.grad).numpy(), (0, 64 - min(64, v.grad.numel())))
                                   for _, v in model.named_parameters()])
    sd = model.state_dict()
    out["running_after"] = np.concatenate(
        [sd[k].numpy().reshape(-1)[:8] for k in sorted(sd) if k.endswith("running_mean") or k.endswith("running_var")])
    out["param_names"] = np.array(names)
    out["sd_keys"] = np.array(keys0)
    np.savez_compressed(os.path.join(OUT, f"e2e_{tag}.npz"), **out)
    print(f"[{tag}] train: loss {loss.item():.6f}")


def op_fixtures(met):
    torch.manual_seed(20260407)          # ConvTranspose2d / Conv2d below draw their default init from the GLOBAL RNG
    g = torch.Generator().manual_seed(7)
    out = {}
    # dilated 3x3 conv forward/backward at three dilations on a ragged (non-square) map
    for d in (1, 2, 4, 16):
        x = torch.randn(2, 3, 12, 20, generator=g, requires_grad=True)
        w = torch.randn(5, 3, 3, 3, generator=g, requires_grad=True)
        b = torch.randn(5, generator=g, requires_grad=True)
        y = F.conv2d(x, w, b, padding=d, dilation=d)
        gy = torch.randn(y.shape, generator=g)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
        for n, v in (("x", x), ("w", w), ("b", b), ("y", y), ("gy", gy), ("gx", gx), ("gw", gw), ("gb", gb)):
            out[f"conv_d{d}_{n}"] = v.detach().numpy()
    # BatchNorm train fwd/bwd + running-stat update, and eval
    x = torch.randn(3, 4, 6, 10, generator=g, requires_grad=True)
    bn = torch.nn.BatchNorm2d(4)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(4, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(4, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(4, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(4, generator=g) + 0.5)
    out["bn_rm0"], out["bn_rv0"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
    bn.train()
    y = bn(x)
    gy = torch.randn(y.shape, generator=g)
    gx, gg, gb = torch.autograd.grad(y, (x, bn.weight, bn.bias), gy)
    out.update(bn_x=x.detach().numpy(), bn_gamma=bn.weight.detach().numpy(), bn_beta=bn.bias.detach().numpy(),
               bn_y=y.detach().numpy(), bn_gy=gy.numpy(), bn_gx=gx.numpy(), bn_gg=gg.numpy(), bn_gb=gb.numpy(),
               bn_rm1=bn.running_mean.numpy().copy(), bn_rv1=bn.running_var.numpy().copy())
    bn.eval()
    out["bn_y_eval"] = bn(x).detach().numpy()
    # max pool with deliberate ties (ReLU zeros) fwd/bwd
    x = torch.relu(torch.randn(2, 3, 8, 12, generator=g)).requires_grad_(True)
    y = F.max_pool2d(x, 2)
    gy = torch.randn(y.shape, generator=g)
    gx, = torch.autograd.grad(y, x, gy)
    out.update(pool_x=x.detach().numpy(), pool_y=y.detach().numpy(), pool_gy=gy.numpy(), pool_gx=gx.numpy())
    # transposed conv 2x2 stride 2
    x = torch.randn(2, 6, 5, 7, generator=g, requires_grad=True)
    ct = torch.nn.ConvTranspose2d(6, 4, 2, stride=2)
    y = ct(x)
    gy = torch.randn(y.shape, generator=g)
    gx, gw, gb = torch.autograd.grad(y, (x, ct.weight, ct.bias), gy)
    out.update(ct_x=x.detach().numpy(), ct_w=ct.weight.detach().numpy(), ct_b=ct.bias.detach().numpy(),
               ct_y=y.detach().numpy(), ct_gy=gy.numpy(), ct_gx=gx.numpy(), ct_gw=gw.numpy(), ct_gb=gb.numpy())
    # concat order + 1x1 head + sigmoid
    a = torch.randn(1, 2, 4, 4, generator=g)
    b2 = torch.randn(1, 3, 4, 4, generator=g)
    out["cat_a"], out["cat_b"], out["cat_y"] = a.numpy(), b2.numpy(), torch.cat([a, b2], 1).numpy()
    hx = torch.randn(2, 64, 4, 6, generator=g)
    hc = torch.nn.Conv2d(64, 2, 1)
    out.update(head_x=hx.numpy(), head_w=hc.weight.detach().numpy(), head_b=hc.bias.detach().numpy(),
               head_p=torch.sigmoid(hc(hx)).detach().numpy())
    # losses from the reference's own metrics_DC.py (incl. the known answer of SURVEY.md a17)
    p = torch.rand(2, 1, 8, 8, generator=g).clamp(1e-4, 1 - 1e-4).requires_grad_(True)
    t = (torch.rand(2, 1, 8, 8, generator=g) < 0.4).float()
    loss = met.focal_dice_loss(p, t, alpha=1.0, gamma=2.0, ratio=0.3)
    gp, = torch.autograd.grad(loss, p)
    out.update(loss_p=p.detach().numpy(), loss_t=t.numpy(), loss_val=np.float64(loss.item()), loss_gp=gp.numpy(),
               loss_dice=np.float64(met.dice_loss(p, t).item()),
               loss_combined=np.float64(met.combined_loss(p, t).item()),
               loss_dicecoef=np.float64(met.dice_coef(t, p).item()),
               loss_known=np.float64(met.focal_dice_loss(torch.full((1, 1, 4, 4), 0.5), torch.zeros(1, 1, 4, 4)).item()))
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("[ops] known-answer focal_dice_loss(p=0.5,t=0) =", out["loss_known"])


def droplet_table_fixture():
    """The reference's own sample OUTPUT (outputs/all_droplets.csv: 303 droplets of its run on its sample images, pixel size
    3.45 px/um) as a data fixture: area -> equivalent_diameter / area_sqmicron / eq_diam_micron known answers for
    quantify_droplets_batch._droplet_table (tests/test_entry_points_cpu.py).  The inputs of that run (images, checkpoint) are
    not in the reference, so the rows pin the column FORMULAS, nothing else."""
    import shutil
    shutil.copyfile(f"{REF}/outputs/all_droplets.csv", os.path.join(OUT, "ref_all_droplets.csv"))
    print("[droplets] copied the reference's sample output table:", sum(1 for _ in open(os.path.join(OUT, "ref_all_droplets.csv"))) - 1, "rows")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--droplets-only" in sys.argv:
        return droplet_table_fixture()
    torch.set_num_threads(8)
    m2, m1, met = ref_modules()
    op_fixtures(met)
    e2e_fixture(m2.UNetDC, 1, 1234, "dc_c1", met)
    e2e_fixture(m2.UNetDC, 3, 4321, "dc_c3", met)
    e2e_fixture(m1.UNet, 3, 2468, "plain_c3", met)
    droplet_table_fixture()


if __name__ == "__main__":
    main()
