"""CPU: pin the oracle (oracle/*.py) against fixtures generated from the LIVE reference
(tools/make_goldens.py).  The reference ships no tests/golden vectors of its own for this path
(SURVEY.md section 4), so these generated vectors are the pin."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import unetdc_numpy as onp
from oracle import unetdc_torch_cpu as otc
from tests.helpers import build_model, load_golden, rel_l2, sd_numpy


@pytest.fixture(scope="module")
def ops():
    return load_golden("ops")


@pytest.mark.parametrize("d", [1, 2, 4, 16])
def test_conv3x3_dilated_fwd_bwd(ops, d):
    x, w, b = ops[f"conv_d{d}_x"], ops[f"conv_d{d}_w"], ops[f"conv_d{d}_b"]
    y = onp.conv3x3(x, w, b, d)
    np.testing.assert_allclose(y, ops[f"conv_d{d}_y"], rtol=0, atol=2e-5)
    gx, gw, gb = onp.conv3x3_bwd(x, w, d, ops[f"conv_d{d}_gy"])
    np.testing.assert_allclose(gx, ops[f"conv_d{d}_gx"], atol=2e-5)
    np.testing.assert_allclose(gw, ops[f"conv_d{d}_gw"], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(gb, ops[f"conv_d{d}_gb"], rtol=1e-5, atol=2e-4)


def test_batchnorm_train_eval(ops):
    x, gam, bet = ops["bn_x"], ops["bn_gamma"], ops["bn_beta"]
    y, cache = onp.bn_train(x, gam, bet)
    np.testing.assert_allclose(y, ops["bn_y"], atol=2e-6)
    gx, gg, gb = onp.bn_train_bwd(ops["bn_gy"], cache, gam)
    np.testing.assert_allclose(gx, ops["bn_gx"], atol=5e-6)
    np.testing.assert_allclose(gg, ops["bn_gg"], atol=2e-5)
    np.testing.assert_allclose(gb, ops["bn_gb"], atol=2e-5)
    cnt = x.shape[0] * x.shape[2] * x.shape[3]
    rm, rv = onp.bn_running_update(ops["bn_rm0"], ops["bn_rv0"], cache[2], cache[3], cnt)
    np.testing.assert_allclose(rm, ops["bn_rm1"], atol=1e-6)
    np.testing.assert_allclose(rv, ops["bn_rv1"], atol=1e-6)      # unbiased variance in running_var
    ye = onp.bn_eval(x, gam, bet, ops["bn_rm1"], ops["bn_rv1"])
    np.testing.assert_allclose(ye, ops["bn_y_eval"], atol=2e-6)


def test_maxpool_with_ties(ops):
    y, arg = onp.maxpool2(ops["pool_x"])
    np.testing.assert_array_equal(y, ops["pool_y"])
    np.testing.assert_array_equal(onp.maxpool2_bwd(ops["pool_gy"], arg), ops["pool_gx"])


def test_conv_transpose(ops):
    y = onp.convT2x2(ops["ct_x"], ops["ct_w"], ops["ct_b"])
    np.testing.assert_allclose(y, ops["ct_y"], atol=2e-6)
    gx, gw, gb = onp.convT2x2_bwd(ops["ct_x"], ops["ct_w"], ops["ct_gy"])
    np.testing.assert_allclose(gx, ops["ct_gx"], atol=5e-6)
    np.testing.assert_allclose(gw, ops["ct_gw"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gb, ops["ct_gb"], rtol=1e-5, atol=2e-5)


def test_cat_order_and_head(ops):
    np.testing.assert_array_equal(np.concatenate([ops["cat_a"], ops["cat_b"]], 1), ops["cat_y"])
    p = onp.sigmoid(onp.conv1x1(ops["head_x"], ops["head_w"], ops["head_b"]))
    np.testing.assert_allclose(p, ops["head_p"], atol=1e-6)


def test_losses(ops):
    p, t = ops["loss_p"], ops["loss_t"]
    assert abs(onp.focal_dice_loss(p, t) - float(ops["loss_val"])) < 1e-6
    np.testing.assert_allclose(onp.focal_dice_loss_bwd(p.astype(np.float64), t.astype(np.float64)),
                               ops["loss_gp"], atol=1e-6)
    # known answer: p = 0.5, t = 0 -> 0.3*0.25*ln2 + 0.7*(1 - ~0) = 0.751986
    assert abs(float(ops["loss_known"]) - 0.751986) < 1e-5
    assert abs(onp.focal_dice_loss(np.full((1, 1, 4, 4), 0.5), np.zeros((1, 1, 4, 4))) - 0.751986) < 1e-5
    # torch port of the loss and the product's host-side loss agree with the reference's values
    from utils import metrics_DC as prod
    pt, tt = torch.from_numpy(p), torch.from_numpy(t)
    assert abs(float(otc.focal_dice_loss(pt, tt)) - float(ops["loss_val"])) < 1e-6
    assert abs(float(prod.focal_dice_loss(pt, tt)) - float(ops["loss_val"])) < 1e-6
    assert abs(float(prod.dice_loss(pt, tt)) - float(ops["loss_dice"])) < 1e-6
    assert abs(float(prod.combined_loss(pt, tt)) - float(ops["loss_combined"])) < 1e-6
    assert abs(float(prod.dice_coef(tt, pt)) - float(ops["loss_dicecoef"])) < 1e-6
    assert abs(float(prod.FocalLoss(reduction="sum")(pt, tt)) / pt.numel()
               - float(prod.FocalLoss()(pt, tt))) < 1e-6


@pytest.mark.parametrize("tag", ["dc_c1", "dc_c3", "plain_c3"])
def test_seeded_init_matches_reference(tag):
    """Same seed => the drop-in module's state dict equals the reference's (136 keys, values)."""
    model, g = build_model(tag, "init")
    keys, sums = recipe.sd_checksums(model.state_dict())
    assert keys == [str(k) for k in g["sd_keys"]] and len(keys) == 136
    np.testing.assert_array_equal(sums, g["init_checksums"])
    assert [k for k, _ in model.named_parameters()] == [str(k) for k in g["param_names"]]
    model, g = build_model(tag, "eval")
    np.testing.assert_array_equal(recipe.sd_checksums(model.state_dict())[1], g["eval_checksums"])


@pytest.mark.parametrize("tag", ["dc_c1", "dc_c3", "plain_c3"])
def test_e2e_eval_numpy_and_torch_port(tag):
    model, g = build_model(tag, "eval")
    dil = dict(model.DILATIONS)
    sd = sd_numpy(model)
    z_ref, mask_ref = g["eval_z"], g["eval_mask"].astype(bool)
    guard = np.abs(z_ref - recipe.LOGIT_THRESH) > 1e-5
    # numpy oracle (fp32 and fp64)
    for dt, tol in ((np.float32, 2e-5), (np.float64, 2e-5)):
        o = onp.UNetOracle(sd, dil, dtype=dt)
        p, z = o.forward(g["eval_x"], train=False, return_logits=True)
        assert np.abs(z - z_ref).max() < tol
        assert np.array_equal((p > 0.3)[guard], mask_ref[guard])
    # torch-CPU port
    sdt = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        p, z = otc.unet_forward(torch.from_numpy(g["eval_x"]), sdt, dil, train=False, return_logits=True)
    assert float((z - torch.from_numpy(z_ref)).abs().max()) < 1e-5
    assert np.array_equal((p > 0.3).numpy(), mask_ref)          # same ATen ops: exact incl. guard band
    # the drop-in module's own CPU path (BASELINE config 0 plumbing)
    model.eval()
    with torch.no_grad():
        p2 = model(torch.from_numpy(g["eval_x"]))
    np.testing.assert_array_equal(p2.numpy(), g["eval_probs"])


@pytest.mark.parametrize("tag", ["dc_c1", "plain_c3"])
def test_e2e_train_grads_numpy_and_torch_port(tag):
    model, g = build_model(tag, "train")
    dil = dict(model.DILATIONS)
    names = [str(k) for k in g["param_names"]]
    x, t = g["train_x"], g["train_t"]
    # torch port
    sdt = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss, p, grads = otc.train_step_grads(torch.from_numpy(x), torch.from_numpy(t), sdt, dil)
    assert abs(float(loss) - float(g["train_loss"])) < 1e-6
    np.testing.assert_allclose(p.numpy(), g["train_probs"], atol=1e-6)
    for i, k in enumerate(names):
        assert abs(float(grads[k].double().norm()) - g["grad_norms"][i]) <= 1e-4 * g["grad_norms"][i] + 1e-7, k
    run = np.concatenate([sdt[k].numpy().reshape(-1)[:8] for k in sorted(sdt)
                          if k.endswith("running_mean") or k.endswith("running_var")])
    np.testing.assert_allclose(run, g["running_after"], atol=1e-6)
    # numpy oracle in fp64: loss, dL/dp, every parameter gradient
    o = onp.UNetOracle(sd_numpy(model), dil, dtype=np.float64)
    pn = o.forward(x, train=True)
    assert abs(onp.focal_dice_loss(pn, t.astype(np.float64)) - float(g["train_loss"])) < 1e-5
    gp = onp.focal_dice_loss_bwd(pn, t.astype(np.float64))
    np.testing.assert_allclose(gp, g["train_dprobs"], atol=1e-7, rtol=1e-4)
    gn = o.backward(gp)
    for i, k in enumerate(names):
        probe = recipe.grad_probe(torch.from_numpy(np.ascontiguousarray(gn[k]))).numpy()
        ref = g["grad_probes"][i][:probe.size]
        scale = max(g["grad_norms"][i] / np.sqrt(gn[k].size), 1e-8)
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            # conv bias before train-mode BN: true gradient is 0, the reference holds fp32 noise
            assert np.abs(probe).max() < 1e-5, k
            continue
        # fp64 oracle vs the fp32 reference: the reference's own rounding is 4e-3..9e-3 relative on the
        # deepest gradients here (BatchNorm over 8 samples at the 2x2 bottleneck is ill-conditioned;
        # measured with torch fp64 autograd, which this oracle matches to 1e-14)
        assert np.abs(probe - ref).max() < 8e-2 * scale + 1e-6, (k, np.abs(probe - ref).max(), scale)
        assert abs(np.linalg.norm(gn[k]) - g["grad_norms"][i]) <= 2e-2 * g["grad_norms"][i] + 1e-7, k
    rm, rv = o.new_running["enc1.1.running_mean"], o.new_running["enc1.1.running_var"]
    np.testing.assert_allclose(rm, sdt["enc1.1.running_mean"].numpy(), atol=1e-6)
    np.testing.assert_allclose(rv, sdt["enc1.1.running_var"].numpy(), atol=1e-6)
