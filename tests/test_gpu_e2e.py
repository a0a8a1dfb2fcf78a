"""GPU end-to-end parity: the drop-in module on the HIP path vs (a) the committed golden vectors
generated from the live reference and (b) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): fp32 path -- pre-sigmoid within 1e-3, thresholded mask equal on every
pixel outside the |z - ln(3/7)| <= 1e-5 guard band (SURVEY.md section 8c protocol; guard-band pixels are
counted and must not differ by more than the measured kernel error).  Gradients are compared against
an fp64 evaluation with a tolerance tied to the fp32 reference's own rounding error, because
BatchNorm over 8 samples at the 2x2 bottleneck of the 32x32 goldens is ill-conditioned (the fp32
reference itself is 4e-3..9e-3 away from fp64 there, see tests/test_oracle_golden.py).
bf16 path -- throughput configuration: probabilities within 3e-2 of the fp32 reference and 2e-2 of the bf16-STORAGE
emulation of the reference (oracle/unetdc_torch_cpu.py, emulate_bf16=True), loss within 2e-2 / 2e-3 relative of the two;
gradients per large tensor: cosine > 0.93 against the emulation and > 0.80 against pure fp32 at 128 x 128 (at full size:
>= 0.94 and no worse than the emulation's own cosine to fp32 minus 0.03) -- rounding the forward storage to bf16 moves the
deep-layer gradients of a randomly initialised network to cosine ~0.90 from fp32 whatever computes them;
that the optimisation nevertheless follows the fp32 trajectory is checked over 30 steps in test_gpu_training.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import unetdc_torch_cpu as otc
from tests.helpers import build_model, rel_l2

pytestmark = pytest.mark.gpu


def logit(p):
    p = p.double().clamp(1e-12, 1 - 1e-12)
    return torch.log(p / (1 - p))


@pytest.mark.parametrize("tag", ["dc_c1", "dc_c3", "plain_c3"])
def test_eval_forward_fp32_matches_golden(tag):
    model, g = build_model(tag, "eval")
    model = model.cuda().eval()
    x = torch.from_numpy(g["eval_x"]).cuda()
    with torch.no_grad():
        p = model(x).cpu()
    z_ref = torch.from_numpy(g["eval_z"]).double()
    z = logit(p)
    err = float((z - z_ref).abs().max())
    assert err < 1e-3, f"pre-sigmoid max error {err}"
    mask, mask_ref = (p > 0.3).numpy(), g["eval_mask"].astype(bool)
    dist = (z_ref - recipe.LOGIT_THRESH).abs().numpy()
    guard = dist > 1e-5
    assert np.array_equal(mask[guard], mask_ref[guard])
    flips = np.logical_and(~guard, mask != mask_ref)
    assert np.all(dist[flips] <= err + 1e-7)           # a guard-band flip must be explained by the kernel error
    print(f"[{tag}] max|dz|={err:.2e} guard-band pixels={int((~guard).sum())} flips there={int(flips.sum())}")


@pytest.mark.parametrize("tag", ["dc_c1", "plain_c3"])
def test_train_step_fp32_matches_golden_and_fp64(tag):
    from utils.metrics_DC import focal_dice_loss
    model, g = build_model(tag, "train")
    dil = dict(model.DILATIONS)
    names = [str(k) for k in g["param_names"]]
    x, t = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_t"])
    # fp64 / fp32 CPU evaluations of the same step
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    _, p64, g64 = otc.train_step_grads(x.double(), t.double(), sd64, dil)
    sd32 = {k: v.clone() for k, v in sd.items()}
    _, p32, g32 = otc.train_step_grads(x, t, sd32, dil)
    # Conditioning: a ReLU / max-pool decision that sits within rounding of a tie flips under ANY fp32 evaluation order
    # and moves every gradient by a finite amount (plain_c3: 1.4e-6 -> 6.9e-5 or 3e-3).  The yardstick for the HIP
    # path is therefore the worst error of the CPU fp32 evaluation over a few 1-ulp perturbations of the input.
    torch.manual_seed(0)
    e_noise = {}
    for _ in range(6):
        xs = x * (1 + (torch.rand_like(x) - 0.5) * 2.4e-7)
        _, _, gn = otc.train_step_grads(xs, t, {k: v.clone() for k, v in sd.items()}, dil)
        for k, v in gn.items():
            e_noise[k] = max(e_noise.get(k, 0.0), float((v.double() - g64[k]).norm()) / max(float(g64[k].norm()), 1e-30))
    # HIP path
    model = model.cuda().train()
    p = model(x.cuda())
    loss = focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    assert abs(loss.item() - float(g["train_loss"])) < 2e-5
    assert float((p.detach().cpu() - torch.from_numpy(g["train_probs"])).abs().max()) < 1e-4
    worst = 0.0
    for k, prm in model.named_parameters():
        ref = g64[k]
        n64 = float(ref.norm())
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            assert float(prm.grad.abs().max()) < 1e-4, k       # exact value is 0; reference holds fp32 noise
            continue
        e_hip = float((prm.grad.cpu().double() - ref).norm()) / n64
        e_ref = max(float((g32[k].double() - ref).norm()) / n64, e_noise.get(k, 0.0))
        worst = max(worst, e_hip / max(e_ref, 1e-7))
        assert e_hip < max(4.0 * e_ref, 2e-5), (k, e_hip, e_ref)
        i = names.index(k)
        assert abs(float(prm.grad.double().norm()) - g["grad_norms"][i]) <= 3e-2 * g["grad_norms"][i] + 1e-7, k
        # element level against numbers the LIVE reference produced: 64 strided samples of this gradient (tools/make_goldens.py)
        probe = recipe.grad_probe(prm.grad.cpu()).double().numpy()
        ref_probe = g["grad_probes"][i][:probe.size].astype(np.float64)
        scale = n64 / np.sqrt(prm.numel())                    # RMS element of the tensor
        tol = max(16.0 * e_ref, 1e-4) * scale * np.sqrt(probe.size) + 1e-9
        assert float(np.linalg.norm(probe - ref_probe)) <= tol, (k, float(np.linalg.norm(probe - ref_probe)), tol)
    # running statistics were updated like nn.BatchNorm2d does
    run = np.concatenate([v.cpu().numpy().reshape(-1)[:8] for k, v in sorted(model.state_dict().items())
                          if k.endswith("running_mean") or k.endswith("running_var")])
    np.testing.assert_allclose(run, g["running_after"], rtol=1e-4, atol=1e-5)
    assert int(model.enc1[1].num_batches_tracked) == 1
    print(f"[{tag}] worst grad error ratio HIP/fp32-reference (both vs fp64) = {worst:.2f}")


@pytest.mark.parametrize("tag", ["dc_c1", "dc_c3"])
def test_input_gradient_fp32_matches_cpu_module(tag):
    """dL/dx of the whole network (the reference module gives it through plain autograd): the HIP path's input gradient vs
    the same module on the CPU (ATen, the reference's arithmetic), fp32, train mode; parameters keep their gradients too."""
    from utils.metrics_DC import focal_dice_loss
    model, g = build_model(tag, "train")
    x, t = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_t"])
    cpu = build_model(tag, "train")[0].train()
    xc = x.clone().requires_grad_(True)
    focal_dice_loss(cpu(xc), t, alpha=1.0, gamma=2.0, ratio=0.3).backward()
    model = model.cuda().train()
    xg = x.cuda().requires_grad_(True)
    focal_dice_loss(model(xg), t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3).backward()
    assert xg.grad is not None and xg.grad.shape == x.shape
    # the 2 x 2 bottleneck of the 32 x 32 goldens is ill-conditioned (see the module docstring): the bar is the parameter
    # gradients' own bar of this file, a few times the fp32 reference's distance from fp64
    e = rel_l2(xg.grad.cpu().numpy(), xc.grad.numpy())
    print(f"[{tag}] rel-L2 of dL/dx, HIP fp32 vs CPU fp32: {e:.2e}")
    assert e < 5e-2
    assert float(model.enc1[0].weight.grad.abs().max()) > 0


@pytest.mark.parametrize("tag", ["dc_c1", "plain_c3"])
def test_eval_mode_autograd_frozen_batchnorm_fp32(tag):
    """model.eval() with gradients enabled (fine-tuning with frozen BatchNorm statistics; the reference module supports it
    through plain autograd): probabilities equal the no_grad eval forward, parameter and input gradients match the same
    module on the CPU (ATen) within the bar of the train-mode test, running statistics stay untouched."""
    from utils.metrics_DC import focal_dice_loss
    model, g = build_model(tag, "eval")
    x, t = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_t"])
    cpu = build_model(tag, "eval")[0].eval()
    xc = x.clone().requires_grad_(True)
    pc = cpu(xc)
    focal_dice_loss(pc, t, alpha=1.0, gamma=2.0, ratio=0.3).backward()
    model = model.cuda().eval()
    before = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        p_ng = model(x.cuda()).cpu()
    xg = x.cuda().requires_grad_(True)
    p = model(xg)
    focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3).backward()
    assert float((p.detach().cpu() - pc.detach()).abs().max()) < 1e-5
    assert float((p.detach().cpu() - p_ng).abs().max()) < 1e-5        # training-path kernels vs folded-BatchNorm inference kernels
    for k, v in model.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k
    worst = 0.0
    for (k, pg), pcpu in zip(model.named_parameters(), cpu.parameters()):
        ref = pcpu.grad.double()
        e = float((pg.grad.cpu().double() - ref).norm()) / max(float(ref.norm()), 1e-12)
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    e = rel_l2(xg.grad.cpu().numpy(), xc.grad.numpy())
    print(f"[{tag}] eval-mode autograd: worst parameter-gradient rel-L2 {worst:.2e}, dL/dx rel-L2 {e:.2e}")
    assert e < 2e-3


@pytest.mark.parametrize("tag", ["dc_c1"])
def test_bf16_path_close_to_reference(tag):
    from utils.metrics_DC import focal_dice_loss
    model, g = build_model(tag, "eval")
    model = model.cuda()
    model.set_compute_dtype("bf16")
    model.eval()
    with torch.no_grad():
        p = model(torch.from_numpy(g["eval_x"]).cuda()).cpu()
    assert float((p - torch.from_numpy(g["eval_probs"])).abs().max()) < 3e-2
    # training step on a better-conditioned input than the 32x32 golden (whose 2x2 bottleneck gives
    # BatchNorm 8 samples per channel: even the fp32 reference is ~1e-2 off fp64 there)
    model.train()
    xc, tc = recipe.seeded_input(21, (4, 1, 128, 128)), recipe.seeded_target(22, (4, 1, 128, 128))
    p = model(xc.cuda())
    loss = focal_dice_loss(p, tc.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    cpu_model, _ = build_model(tag, "train")
    sd = {k: v.detach().clone() for k, v in cpu_model.state_dict().items()}
    loss_ref, p_ref, g32 = otc.train_step_grads(xc, tc, sd, dict(cpu_model.DILATIONS))
    sd = {k: v.detach().clone() for k, v in cpu_model.state_dict().items()}
    loss_emu, p_emu, gemu = otc.train_step_grads(xc, tc, sd, dict(cpu_model.DILATIONS), emulate_bf16=True)
    assert abs(loss.item() - float(loss_ref)) < 2e-2 * float(loss_ref)
    assert abs(loss.item() - float(loss_emu)) < 2e-3 * float(loss_emu)
    assert float((p.detach().cpu() - p_emu).abs().max()) < 2e-2
    # Gradients: the HIP bf16 path must agree with the bf16-STORAGE emulation of the reference (same
    # rounding points, fp32 accumulate); against pure fp32 only a loose bound holds because the forward
    # rounding itself moves deep-layer gradients at random init (oracle/unetdc_torch_cpu.py:_StoreBF16).
    rows = []
    for k, prm in model.named_parameters():
        assert torch.isfinite(prm.grad).all()
        if prm.numel() < 4096 or k.endswith(".bias"):
            continue
        a = prm.grad.cpu().double().reshape(-1)
        be, bf = gemu[k].double().reshape(-1), g32[k].double().reshape(-1)
        rows.append((k, float(a @ be / (a.norm() * be.norm())), float(a @ bf / (a.norm() * bf.norm()))))
    print(f"[{tag}] bf16 HIP gradient cosine vs (bf16-emulating oracle / fp32 oracle): "
          + ", ".join(f"{k}={ce:.4f}/{cf:.4f}" for k, ce, cf in rows))
    for k, ce, cf in rows:
        assert ce > 0.93 and cf > 0.80, (k, ce, cf)


def test_full_size_eval_mask_fp32():
    """BASELINE config 1: bs 8, 512x512, fp32 forward; mask bit-exact vs the CPU path (guard band)."""
    torch.manual_seed(11)
    from models.model_2 import UNetDC
    model = UNetDC(in_channels=1, out_channels=1)
    recipe.perturb_bn(model.state_dict(), 12)
    model.eval()
    x = recipe.seeded_input(13, (8, 1, 512, 512))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        _, z_cpu = otc.unet_forward(x, sd, dict(model.DILATIONS), train=False, return_logits=True)
    shift = recipe.LOGIT_THRESH - float(z_cpu.median())
    with torch.no_grad():
        model.out_conv.bias += shift
    z_ref = (z_cpu + shift).double()
    model = model.cuda()
    with torch.no_grad():
        p = model(x.cuda()).cpu()
    z = logit(p)
    err = float((z - z_ref).abs().max())
    assert err < 1e-3, err
    mask, mask_ref = (p > 0.3).numpy(), (z_ref > recipe.LOGIT_THRESH).numpy()
    dist = (z_ref - recipe.LOGIT_THRESH).abs().numpy()
    guard = dist > 1e-5
    assert 0.3 < mask_ref.mean() < 0.7
    assert np.array_equal(mask[guard], mask_ref[guard])
    flips = np.logical_and(~guard, mask != mask_ref)
    assert np.all(dist[flips] <= err + 1e-7)
    print(f"[full-size] max|dz|={err:.2e}, {int((~guard).sum())} guard-band pixels of {mask.size}, "
          f"{int(flips.sum())} flips inside the band")


def test_full_size_eval_mask_bf16_agreement_with_fp32_reference():
    """The THROUGHPUT configuration against north_star's mask criterion, stated as measured: bs 8, 512 x 512 x 1, calibrated
    50/50 mask (same recipe as the fp32 test above), bf16 eval forward on the device vs the fp32 CPU path.  bf16 storage cannot
    be bit-exact after the threshold -- the fp32 path is (test above) -- so this test PRINTS and bounds what it is: the fraction
    of pixels whose mask equals the fp32 reference's, the worst |dz| among the pixels that flip, and the same two figures
    against the bf16-storage evaluation of the reference (oracle emulate_bf16=True), which shows how much of the disagreement
    is bf16 storage itself rather than this implementation.  A flipped pixel must lie within the measured logit error of the
    threshold: no flip may come from anything but rounding."""
    torch.manual_seed(11)
    from models.model_2 import UNetDC
    model = UNetDC(in_channels=1, out_channels=1)
    recipe.perturb_bn(model.state_dict(), 12)
    model.eval()
    x = recipe.seeded_input(13, (8, 1, 512, 512))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, z32 = otc.unet_forward(x, sd, dict(model.DILATIONS), train=False, return_logits=True)
        _, zem = otc.unet_forward(x, sd, dict(model.DILATIONS), train=False, return_logits=True, emulate_bf16=True)
    shift = recipe.LOGIT_THRESH - float(z32.median())
    with torch.no_grad():
        model.out_conv.bias += shift
    z32, zem = (z32 + shift).double(), (zem + shift).double()
    model = model.cuda()
    model.set_compute_dtype("bf16")
    with torch.no_grad():
        p = model(x.cuda()).cpu()
    z = logit(p)
    mask = (p > 0.3).numpy()
    rows = {}
    for tag, zr in (("fp32 reference", z32), ("bf16-storage reference", zem)):
        mref = (zr > recipe.LOGIT_THRESH).numpy()
        flips = mask != mref
        dz = (z - zr).abs().numpy()
        dist = (zr - recipe.LOGIT_THRESH).abs().numpy()
        rows[tag] = (1.0 - float(flips.mean()), float(dz.max()), float(dz.mean()), float(dist[flips].max()) if flips.any() else 0.0,
                     float(dz[flips].max()) if flips.any() else 0.0)
        # a flip is only legitimate where the reference logit sits closer to the threshold than the logit error there
        assert np.all(dist[flips] <= dz[flips] + 1e-7), tag
    a32, aem = rows["fp32 reference"], rows["bf16-storage reference"]
    print("[full-size bf16 eval mask] " + "; ".join(
        f"vs {k}: {v[0] * 100:.3f} % of 2097152 pixels equal, max|dz| {v[1]:.3e}, mean|dz| {v[2]:.3e}, farthest flipped pixel "
        f"{v[3]:.3e} from the threshold, worst |dz| among flips {v[4]:.3e}" for k, v in rows.items()))
    if os.path.isdir("gpurun_out"):                        # (the figures DESIGN.md section 2 quotes)
        with open("gpurun_out/bf16_eval_mask_agreement.txt", "w") as f:
            for k, v in rows.items():
                f.write(f"vs {k}: equal {v[0] * 100:.4f} % of 2097152 pixels; max|dz| {v[1]:.4e}; mean|dz| {v[2]:.4e}; farthest flipped "
                        f"pixel {v[3]:.4e} from the threshold; worst |dz| among flips {v[4]:.4e}\n")
    assert 0.3 < float((z32 > recipe.LOGIT_THRESH).double().mean()) < 0.7
    # bands = a few times what was MEASURED on MI355X (round 4: 99.575 % / 99.469 % of the pixels equal, max |dz| 5.0e-3 /
    # 6.5e-3, mean |dz| 8.0e-4 / 1.0e-3): a 4x regression of the bf16 path fails here (the fp32 path's bar is 1e-3, above)
    assert a32[0] > 0.99 and aem[0] > 0.99, (a32, aem)
    assert a32[1] < 2e-2 and aem[1] < 2e-2, (a32, aem)    # max |dz|: bf16 storage, 23 layers deep
    assert a32[2] < 3e-3 and aem[2] < 3e-3, (a32, aem)    # mean |dz|


def test_full_size_bf16_gradients_vs_bf16_emulating_oracle():
    """Headline configuration (bs 8, 512x512x1, bf16 storage, fp32 accumulate) on the same batch as
    two CPU evaluations of the reference: plain fp32, and fp32 with the SAME bf16 storage points
    (weights / conv outputs / activations; oracle/unetdc_torch_cpu.py:_StoreBF16).

    At random init the deep-layer gradients are ill-conditioned with respect to 1e-3-level
    perturbations of the activations: the bf16-storage evaluation of the reference itself is only
    0.90-0.93 cosine from fp32 in enc2..bottleneck, and two bf16 evaluations that differ only in
    fp32 summation order (which flips ~0.1 % of the bf16 roundings) are 0.96-0.98 from each other.
    Parity criterion for the bf16 path: (a) loss/probabilities match the bf16-storage oracle tightly,
    (b) the HIP gradients are as close to fp32 as the bf16-storage oracle is (per tensor, -0.03),
    (c) HIP vs bf16-storage oracle cosine >= 0.94 everywhere, >= 0.999 in the last decoder block."""
    from models.model_2 import UNetDC
    from utils.metrics_DC import focal_dice_loss
    torch.manual_seed(21)
    model = UNetDC(1, 1)
    x = recipe.seeded_input(22, (8, 1, 512, 512))
    t = recipe.seeded_target(23, (8, 1, 512, 512), frac=0.1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss32, _, g32 = otc.train_step_grads(x, t, sd, dict(model.DILATIONS))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss_emu, p_emu, gemu = otc.train_step_grads(x, t, sd, dict(model.DILATIONS), emulate_bf16=True)
    model = model.cuda().train()
    model.set_compute_dtype("bf16")
    p = model(x.cuda())
    loss = focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    assert abs(loss.item() - float(loss_emu)) < 2e-3 * float(loss_emu)
    assert abs(loss.item() - float(loss32)) < 1e-2 * float(loss32)
    assert float((p.detach().cpu() - p_emu).abs().max()) < 3e-2

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return float(a @ b / (a.norm() * b.norm()))
    rows = []
    for k, prm in model.named_parameters():
        if prm.numel() < 4096 or k.endswith(".bias"):
            continue
        g = prm.grad.cpu()
        rows.append((k, cos(g, gemu[k]), cos(g, g32[k]), cos(gemu[k], g32[k])))
    print("[full-size bf16] per tensor cos(HIP,bf16-oracle) / cos(HIP,fp32) / cos(bf16-oracle,fp32): "
          + ", ".join(f"{k}={a:.4f}/{b:.4f}/{c:.4f}" for k, a, b, c in rows))
    for k, c_he, c_hf, c_ef in rows:
        assert c_hf >= c_ef - 0.03, (k, c_hf, c_ef)
        assert c_he > (0.999 if k.startswith("dec1") else 0.94), (k, c_he)


def test_full_size_train_step_fp32_vs_cpu_port():
    """BASELINE configs[2]'s shape in the PARITY dtype: 8 x 1 x 512 x 512, fp32 forward + Focal/Dice loss + backward on the HIP
    path vs the CPU port of the reference (oracle/unetdc_torch_cpu.py: the ATen ops the reference calls), same weights, same
    batch.  Bars: loss within 1e-5 relative, probabilities within 1e-4; every weight gradient's rel-L2 distance from the CPU
    result <= 4x the CPU path's OWN distance from a rerun on a 1-ulp-perturbed input (the conditioning yardstick of
    test_train_step_fp32_matches_golden_and_fp64: ReLU / max-pool ties that flip under any fp32 evaluation order), floor 2e-5."""
    from models.model_2 import UNetDC
    from utils.metrics_DC import focal_dice_loss
    torch.manual_seed(31)
    model = UNetDC(1, 1)
    dil = dict(model.DILATIONS)
    x = recipe.seeded_input(32, (8, 1, 512, 512))
    t = recipe.seeded_target(33, (8, 1, 512, 512), frac=0.1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss_ref, p_ref, g_ref = otc.train_step_grads(x, t, {k: v.clone() for k, v in sd.items()}, dil)
    torch.manual_seed(0)
    xs = x * (1 + (torch.rand_like(x) - 0.5) * 2.4e-7)                   # one 1-ulp perturbation of the input
    _, _, g_pert = otc.train_step_grads(xs, t, {k: v.clone() for k, v in sd.items()}, dil)
    model = model.cuda().train()                                          # compute dtype "f32" is the module's default
    p = model(x.cuda())
    loss = focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    assert abs(loss.item() - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)), (loss.item(), float(loss_ref))
    assert float((p.detach().cpu() - p_ref).abs().max()) < 1e-4
    worst, rows = 0.0, []
    for k, prm in model.named_parameters():
        ref = g_ref[k].double()
        n = float(ref.norm())
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            assert float(prm.grad.abs().max()) < 1e-4, k                  # structural zero in front of train-mode BatchNorm
            continue
        e_hip = float((prm.grad.cpu().double() - ref).norm()) / max(n, 1e-30)
        e_self = float((g_pert[k].double() - ref).norm()) / max(n, 1e-30)
        rows.append((k, e_hip, e_self))
        worst = max(worst, e_hip / max(e_self, 5e-6))
        assert e_hip <= max(4.0 * e_self, 2e-5), (k, e_hip, e_self)
    top = sorted(rows, key=lambda r: -r[1])[:4]
    print(f"[full-size fp32 train step] loss {loss.item():.7f} vs CPU {float(loss_ref):.7f}; worst gradient error ratio "
          f"HIP / CPU-self-noise = {worst:.2f}; largest rel-L2: " + ", ".join(f"{k}={a:.2e} (cpu noise {b:.2e})" for k, a, b in top))


def test_full_size_train_step_bf16_properties():
    """BASELINE headline config (bs 8, 512x512x1, bf16 fwd+bwd): size-independent properties --
    bitwise run-to-run determinism (two-stage reductions, no atomics), finite gradients, and the
    structural zero of conv-bias gradients in front of train-mode BatchNorm."""
    from models.model_2 import UNetDC
    from utils.metrics_DC import focal_dice_loss
    torch.manual_seed(5)
    model = UNetDC(1, 1).cuda().train()
    model.set_compute_dtype("bf16")
    x = recipe.seeded_input(6, (8, 1, 512, 512)).cuda()
    t = recipe.seeded_target(7, (8, 1, 512, 512)).cuda()
    snaps = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss = focal_dice_loss(model(x), t, alpha=1.0, gamma=2.0, ratio=0.3)
        loss.backward()
        snaps.append((loss.item(), [p.grad.clone() for p in model.parameters()]))
    assert snaps[0][0] == snaps[1][0]
    for a, b in zip(snaps[0][1], snaps[1][1]):
        assert torch.equal(a, b)
        assert torch.isfinite(a).all()
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    for k, p in model.named_parameters():
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            assert float(p.grad.abs().max()) < 1e-3 * gmax, k


def test_eval_after_fused_adam_step_uses_updated_weights():
    """torch.optim.Adam(fused=True) updates parameters without bumping their version counters: the packed bf16/fp32
    weight images must still follow.  One fused step, then an eval forward on the HIP path == the CPU port evaluated
    with the UPDATED state dict (and differs measurably from the pre-step output)."""
    model, g = build_model("dc_c1", "train")
    x, t = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_t"])
    from utils.metrics_DC import focal_dice_loss
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
    with torch.no_grad():
        model.eval()
        p_before = model(x.cuda()).cpu()
        model.train()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        focal_dice_loss(model(x.cuda()), t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3).backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        p_after = model(x.cuda()).cpu()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    p_ref = otc.unet_forward(x, sd, dict(model.DILATIONS), train=False)
    assert float((p_after - p_ref).abs().max()) < 1e-4
    assert float((p_after - p_before).abs().max()) > 1e-3          # the two Adam steps did change the output


@pytest.mark.parametrize("shape", [(2, 3, 96, 160), (3, 1, 48, 80)])
def test_non_power_of_two_maps_fp32_train_step(shape):
    """H, W divisible by 16 but not powers of two (the reference accepts any such size): every kernel takes its
    division-based pixel decode, the pooled levels get odd row counts (3 x 5 at the bottleneck), M is not a multiple
    of the 256-pixel block tile.  fp32 HIP step vs the CPU port: loss, probabilities, gradients."""
    from models.model_2 import UNetDC
    from utils.metrics_DC import focal_dice_loss
    n, c, h, w = shape
    torch.manual_seed(11)
    model = UNetDC(c, 1).train()
    g = torch.Generator().manual_seed(12)
    x = torch.rand(n, c, h, w, generator=g)
    t = (torch.rand(n, 1, h, w, generator=g) > 0.7).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss_ref, p_ref, g_ref = otc.train_step_grads(x, t, sd, dict(model.DILATIONS))
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    _, _, g64 = otc.train_step_grads(x.double(), t.double(), sd64, dict(model.DILATIONS))
    model = model.cuda()
    p = model(x.cuda())
    loss = focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    assert abs(loss.item() - float(loss_ref)) < 2e-5
    assert float((p.detach().cpu() - p_ref).abs().max()) < 1e-4
    # Gradients, per tensor: at random init this step is chaotic at the 1e-3 level (a ReLU / max-pool decision within
    # rounding of a tie flips under ANY fp32 evaluation order and moves the gradients by a finite amount), so the
    # yardstick of each tensor is the error of the CPU fp32 evaluation of THAT tensor against fp64 -- as evaluated, and
    # under six 1-ulp perturbations of the input (same protocol as test_train_step_fp32_matches_golden_and_fp64).
    def rel(a, k):
        return float((a.double() - g64[k]).norm()) / max(float(g64[k].norm()), 1e-30)
    yard = {k: rel(g_ref[k], k) for k in g64}
    torch.manual_seed(0)
    for _ in range(6):
        xs = x * (1 + (torch.rand_like(x) - 0.5) * 2.4e-7)
        _, _, gn = otc.train_step_grads(xs, t, {k: v.clone() for k, v in sd.items()}, dict(model.DILATIONS))
        for k, v in gn.items():
            yard[k] = max(yard[k], rel(v, k))
    worst = 0.0
    for k, prm in model.named_parameters():
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            assert float(prm.grad.abs().max()) < 1e-4, k       # exact value is 0 in front of train-mode BatchNorm
            continue
        e_hip = rel(prm.grad.cpu(), k)
        worst = max(worst, e_hip / max(yard[k], 1e-7))
        assert e_hip < max(4.0 * yard[k], 2e-5), (k, e_hip, yard[k])
    print(f"[np2 {shape}] worst per-tensor gradient error ratio HIP / fp32 reference (both vs fp64) = {worst:.2f}")


def test_two_live_forwards_and_a_validation_forward_before_the_backwards():
    """The reference module is plain autograd (models/model_2.py:56-80): several forwards may be alive before any backward, and
    an eval forward may run in between.  On the HIP path an engine whose activations belong to a live graph is busy and the
    next forward of that shape gets its own engine: both backwards give bit for bit the gradients of the one-at-a-time order.
    What is still refused, loudly: a SECOND backward through a retained graph after another forward has re-used the buffers; an
    in-place edit of the returned probabilities is caught by autograd's saved-tensor version check."""
    from unet_dc_segmentation_amd._lib import UnetdcError
    from utils.metrics_DC import focal_dice_loss
    model, g = build_model("dc_c1", "train")
    model = model.cuda().train()
    x, t = torch.from_numpy(g["train_x"]).cuda(), torch.from_numpy(g["train_t"]).cuda()
    loss_of = lambda p: focal_dice_loss(p, t, alpha=1.0, gamma=2.0, ratio=0.3)       # noqa: E731
    grads = lambda: {k: q.grad.clone() for k, q in model.named_parameters()}        # noqa: E731
    loss_of(model(x)).backward()
    g1 = grads()
    model.zero_grad(set_to_none=True)
    loss_of(model(x * 0.5)).backward()
    g2 = grads()
    model.zero_grad(set_to_none=True)
    assert len(model._engines[(x.device, tuple(x.shape))]) == 1                      # one engine so far
    p1 = model(x)
    p2 = model(x * 0.5)                                  # the first forward's activations stay where they are
    with torch.no_grad():
        model.eval()
        pe = model(x)                                    # a validation forward in between takes a third engine
        model.train()
    assert torch.isfinite(pe).all()
    assert len(model._engines[(x.device, tuple(x.shape))]) == 3
    loss_of(p2).backward()
    assert all(torch.equal(q.grad, g2[k]) for k, q in model.named_parameters())
    model.zero_grad(set_to_none=True)
    loss_of(p1).backward()
    assert all(torch.equal(q.grad, g1[k]) for k, q in model.named_parameters())
    model.zero_grad(set_to_none=True)
    # every engine is free again: the next forwards allocate nothing
    loss_of(model(x)).backward()
    assert len(model._engines[(x.device, tuple(x.shape))]) == 3
    assert all(torch.equal(q.grad, g1[k]) for k, q in model.named_parameters())
    model.zero_grad(set_to_none=True)
    # retained graph: a second backward is fine while the buffers are untouched ...
    p4 = model(x * 0.5)
    loss = loss_of(p4)
    loss.backward(retain_graph=True)
    assert all(torch.equal(q.grad, g2[k]) for k, q in model.named_parameters())
    model.zero_grad(set_to_none=True)
    loss.backward(retain_graph=True)
    assert all(torch.equal(q.grad, g2[k]) for k, q in model.named_parameters())
    model.zero_grad(set_to_none=True)
    # ... and refused once a later forward has taken them (the first backward released the engine)
    for _ in range(3):
        with torch.no_grad():
            model(x)
    with pytest.raises(UnetdcError, match="overwritten by a later forward"):
        loss.backward()
    # an in-place edit of the returned probabilities (read by the head backward) between forward and backward is caught by
    # autograd's saved-tensor version check, as for any ATen module
    model.zero_grad(set_to_none=True)
    p5 = model(x)
    loss5 = loss_of(p5)
    with torch.no_grad():
        p5.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss5.backward()


def test_1024_tiles_bf16_train_step():
    """BASELINE configs[4] per GPU: 4 x 1 x 1024 x 1024, bf16 storage / fp32 accumulate.  Bitwise run-to-run
    determinism, finite gradients, and loss / probabilities against the bf16-storage evaluation of the CPU port."""
    from models.model_2 import UNetDC
    from utils.metrics_DC import focal_dice_loss
    torch.manual_seed(31)
    model = UNetDC(1, 1)
    x = recipe.seeded_input(32, (4, 1, 1024, 1024))
    t = recipe.seeded_target(33, (4, 1, 1024, 1024), frac=0.1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # the CPU evaluation (most of this test's time) runs on the first two tiles; the bs-4 step is checked for determinism
    loss_emu, p_emu, gemu = otc.train_step_grads(x[:2], t[:2], sd, dict(model.DILATIONS), emulate_bf16=True)
    model = model.cuda().train()
    model.set_compute_dtype("bf16")
    xc, tc = x.cuda(), t.cuda()
    snaps = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        p = model(xc)
        loss = focal_dice_loss(p, tc, alpha=1.0, gamma=2.0, ratio=0.3)
        loss.backward()
        snaps.append((loss.item(), [q.grad.clone() for q in model.parameters()]))
    assert snaps[0][0] == snaps[1][0]
    for a, b in zip(*[s[1] for s in snaps]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    # VALUES of the bs-4 step (the CPU evaluation below covers a bs-2 sub-batch only): against the fp32 HIP path on the same
    # batch and weights -- the parity-pinned dtype (test_1024_tile_eval_mask_fp32, test_full_size_train_step_fp32_vs_cpu_port)
    m32 = UNetDC(1, 1)
    m32.load_state_dict(sd)
    m32 = m32.cuda().train()
    p32 = m32(xc)
    loss32 = focal_dice_loss(p32, tc, alpha=1.0, gamma=2.0, ratio=0.3)
    assert abs(snaps[0][0] - loss32.item()) < 1e-2 * loss32.item(), (snaps[0][0], loss32.item())
    assert float((p.detach() - p32.detach()).abs().max()) < 3e-2
    del m32, p32, loss32
    model.zero_grad(set_to_none=True)
    p = model(xc[:2])
    loss = focal_dice_loss(p, tc[:2], alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    snaps = [(loss.item(), None)]
    assert abs(snaps[0][0] - float(loss_emu)) < 2e-3 * float(loss_emu)
    assert float((p.detach().cpu() - p_emu).abs().max()) < 3e-2
    for k, prm in model.named_parameters():
        if k.startswith("dec1") and k.endswith(".weight") and prm.numel() >= 4096:
            a, b = prm.grad.cpu().double().reshape(-1), gemu[k].double().reshape(-1)
            assert float(a @ b / (a.norm() * b.norm())) > 0.999, k


def test_1024_tile_eval_mask_fp32():
    """One 1024 x 1024 tile, fp32 forward: pre-sigmoid within 1e-3 and mask equal to the CPU path outside the guard
    band (same protocol as test_full_size_eval_mask_fp32; the 64 x 64 bottleneck map exercises d = 16 with all nine
    taps in bounds for the interior pixels)."""
    from models.model_2 import UNetDC
    torch.manual_seed(41)
    model = UNetDC(in_channels=1, out_channels=1)
    recipe.perturb_bn(model.state_dict(), 42)
    model.eval()
    x = recipe.seeded_input(43, (1, 1, 1024, 1024))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, z_cpu = otc.unet_forward(x, sd, dict(model.DILATIONS), train=False, return_logits=True)
    shift = recipe.LOGIT_THRESH - float(z_cpu.median())
    with torch.no_grad():
        model.out_conv.bias += shift
    z_ref = (z_cpu + shift).double()
    model = model.cuda()
    with torch.no_grad():
        p = model(x.cuda()).cpu()
    z = logit(p)
    err = float((z - z_ref).abs().max())
    assert err < 1e-3, err
    mask, mask_ref = (p > 0.3).numpy(), (z_ref > recipe.LOGIT_THRESH).numpy()
    dist = (z_ref - recipe.LOGIT_THRESH).abs().numpy()
    guard = dist > 1e-5
    assert 0.3 < mask_ref.mean() < 0.7
    assert np.array_equal(mask[guard], mask_ref[guard])
    flips = np.logical_and(~guard, mask != mask_ref)
    assert np.all(dist[flips] <= err + 1e-7)


def test_quantify_cli_on_gpu_512(tmp_path):
    """BASELINE configs[0] flow on the HIP path: quantify_droplets_batch.main() on 4 synthetic 512 x 512 PNGs, fp32.
    The masks it writes must equal the CPU path's (the same entry point with DEVICE = "cpu") on every pixel outside
    the guard band; droplet tables follow from the masks."""
    import pandas as pd
    from PIL import Image
    import quantify_droplets_batch as q
    from models.model_2 import UNetDC
    rng = np.random.default_rng(0)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    yy, xx = np.mgrid[0:512, 0:512]
    for i in range(4):
        img = (rng.random((512, 512, 3)) * 60).astype(np.uint8)
        for _ in range(40):
            cy, cx, r = rng.integers(8, 504), rng.integers(8, 504), rng.integers(2, 12)
            img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 230
        Image.fromarray(img).save(img_dir / f"im{i}.png")
    torch.manual_seed(0)
    model = UNetDC(3, 1)
    recipe.perturb_bn(model.state_dict(), 5)
    # calibrate the head bias so that the mask is about half ones (random init is all-ones at 0.3)
    xs = torch.stack([q.preprocess(img_dir / f"im{i}.png", 15)[0] for i in range(4)]).cpu()     # GPU preprocessing: bit-exact vs the CPU path (test_gpu_preprocess.py)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, z = otc.unet_forward(xs, sd, dict(model.DILATIONS), train=False, return_logits=True)
        model.out_conv.bias += recipe.LOGIT_THRESH - float(z.median())
        z = z + (recipe.LOGIT_THRESH - float(z.median()))
    ckpt = tmp_path / "best_UNetDC_focal_model.pth"
    torch.save(model.state_dict(), ckpt)
    assert q.DEVICE == "cuda"
    out = q.main(["--img_dir", str(img_dir), "--ckpt_path", str(ckpt), "--out_dir", str(tmp_path / "gpu"),
                  "--batch", "4", "--prob_thresh", "0.3", "--skip_excel", "--skip_histogram",
                  "--background_radius", "15", "--px_per_micron", "3.45"])
    guard = ((z[:, 0].double() - recipe.LOGIT_THRESH).abs() > 1e-5).numpy()
    ref = (z[:, 0].double() > recipe.LOGIT_THRESH).numpy()
    for i in range(4):
        m = np.array(Image.open(out / "predicted_masks" / f"im{i}_pred.png")) > 0
        assert m.shape == (512, 512) and 0.2 < m.mean() < 0.8
        assert np.array_equal(m[guard[i]], ref[i][guard[i]])
    summary = pd.read_csv(out / "summary_per_image.csv")
    assert list(summary.columns) == ["filename", "droplet_count", "total_area_px"] and len(summary) == 4
    drops = pd.read_csv(out / "all_droplets.csv")
    for i in range(4):
        m = np.array(Image.open(out / "predicted_masks" / f"im{i}_pred.png")) > 0
        d = q.quantify(m.astype(np.uint8), 1, 3.45)             # the scipy restatement of reference :81-95
        mine = drops[drops["filename"] == f"im{i}.png"]
        assert len(mine) == len(d) and abs(mine["area"].sum() - d["area"].sum()) < 1e-9
        assert int(summary[summary["filename"] == f"im{i}.png"]["droplet_count"].iloc[0]) == len(d)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_adam_matches_torch_adam_and_refreshes_packed_weights(dtype):
    """unet_dc_segmentation_amd.optim.FusedAdam (csrc/optim.hip: Adam + weight re-pack in one kernel) against
    torch.optim.Adam over 3 steps on identical gradients: parameters and both moments equal to a few ulps (same update
    rule, one fma of difference), then an eval forward must use the UPDATED weights although no pack launch ran
    (the optimizer wrote the packed images itself)."""
    from unet_dc_segmentation_amd.optim import FusedAdam
    from utils.metrics_DC import focal_dice_loss
    model_a, g = build_model("dc_c1", "train")
    model_b, _ = build_model("dc_c1", "train")
    model_a, model_b = model_a.cuda().train(), model_b.cuda().train()
    model_a.set_compute_dtype(dtype)
    model_b.set_compute_dtype(dtype)
    x, t = torch.from_numpy(g["train_x"]).cuda(), torch.from_numpy(g["train_t"]).cuda()
    opt_a = FusedAdam(model_a, lr=1e-2)
    opt_b = torch.optim.Adam(model_b.parameters(), lr=1e-2)
    for k in range(3):
        for m, o in ((model_a, opt_a), (model_b, opt_b)):
            o.zero_grad(set_to_none=True)
            focal_dice_loss(m(x * (1 - 0.1 * k)), t, alpha=1.0, gamma=2.0, ratio=0.3).backward()
        with torch.no_grad():                                # feed B's optimizer the same gradients: isolates the update rule
            for pa, pb in zip(model_a.parameters(), model_b.parameters()):
                pb.grad.copy_(pa.grad)
        opt_a.step()
        opt_b.step()
        for (k_, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
            scale = float(pb.detach().abs().max())
            assert float((pa - pb).abs().max()) <= 4e-6 * scale + 1e-9, (k, k_)
            sa, sb = opt_a.state[pa], opt_b.state[pb]
            assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 2e-6 * float(sb["exp_avg"].abs().max()) + 1e-12
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 2e-6 * float(sb["exp_avg_sq"].abs().max()) + 1e-20
    assert int(float(opt_a.state[next(model_a.parameters())]["step"])) == 3
    # the packed images follow without a re-pack launch
    model_a.eval()
    with torch.no_grad():
        p_after = model_a(x).cpu()
    sd = {k: v.detach().cpu().clone() for k, v in model_a.state_dict().items()}
    p_ref = otc.unet_forward(x.cpu(), sd, dict(model_a.DILATIONS), train=False, emulate_bf16=(dtype == "bf16"))
    assert float((p_after - p_ref).abs().max()) < (1e-4 if dtype == "f32" else 3e-2)
    # state_dict round trip into torch.optim.Adam
    opt_c = torch.optim.Adam(model_a.parameters(), lr=1e-2)
    opt_c.load_state_dict(opt_a.state_dict())
    assert torch.equal(opt_c.state[next(model_a.parameters())]["exp_avg"], opt_a.state[next(model_a.parameters())]["exp_avg"])
