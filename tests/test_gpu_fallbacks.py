"""The A/B switches select older or alternative kernels inside the same library (first-generation register-staged
implicit GEMM and weight gradient, halo-patch convolutions, three-segment tap-fused wgrad, VALU first layer).  They are also the fallbacks for shapes the fast kernels do not take (>= 2 GiB tensors, ragged
pixel counts), so the operator parity tests are re-run under each switch set -- in a child process, because the
switches are read once per process."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCH_SETS = {
    # the >= 2 GiB / ragged-shape fallbacks: first-generation register-staged implicit GEMM and weight gradient, VALU first layer
    "first_generation": {"UNETDC_IGEMM": "legacy", "UNETDC_WGRAD": "legacy", "UNETDC_FIRST": "valu"},
    # every fusion off: stand-alone BatchNorm-backward reduction, column sums, loss; three-segment weight gradient
    "unfused_epilogues": {"UNETDC_WGRAD_RING": "0", "UNETDC_FUSE_BNBWD": "0", "UNETDC_FUSED_LOSS": "0"},
    # round-2 kernels off: halo-patch / per-tap convolutions instead of the persistent lattice kernel (the halo-patch kernel
    # is also the fp32 path), quadrant ring and per-tap weight gradients instead of the tap-split ring and the valid-rectangle
    # kernel, per-pixel first-layer wgrad
    "round1_kernels": {"UNETDC_LATTICE": "0", "UNETDC_WGRAD_SPLIT": "0", "UNETDC_WGRAD_RECT": "0", "UNETDC_FIRST_ROWS": "0"},
}


def test_operator_parity_under_switches():
    """The three switch sets run as three child processes AT THE SAME TIME (the switches are read once per process; four
    processes on the card, well inside the pool's limit of six): sequentially they were 1.5 minutes of the GPU suite."""
    procs = {}
    for name in sorted(SWITCH_SETS):
        env = dict(os.environ, UNETDC_TEST_THIN="1", OMP_NUM_THREADS="4", **SWITCH_SETS[name])   # thinned shape lists: tests/test_gpu_ops.py
        sel = "conv3x3_fwd_dgrad_wgrad or first_conv or conv_transpose or fused_bn_backward_statistics"
        if name == "unfused_epilogues":
            sel = "wgrad_tap_fused or conv3x3_fwd_dgrad_wgrad"
        if name == "round1_kernels":
            sel = "(" + sel + " or wgrad_tap_fused) and not f32"
        cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-m", "gpu", "-x", "-q",
               "-k", sel, "-p", "no:cacheprovider"]
        procs[name] = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for name, pr in procs.items():
        out, _ = pr.communicate(timeout=900)
        assert pr.returncode == 0, name + ":\n" + out[-3000:]
        assert " passed" in out, name


def test_train_step_under_unfused_switches_matches_default():
    """One bf16 training step at a small size: gradients with every fusion switched off == the default path within bf16
    rounding (same kernels' math, different launch structure).  (The bit-identical round-3/4 fusions -- head-input gradient recomputed,
    first stage's BatchNorm backward on load of its weight gradient, fused column sums -- have no switch any more: their operator
    tests in test_gpu_ops.py compare them with the stored / two-pass forms.)"""
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from models.model_2 import UNetDC
from utils.metrics_DC import focal_dice_loss
from oracle import recipe
torch.manual_seed(3)
m = UNetDC(1, 1).cuda().train(); m.set_compute_dtype("bf16")
x = recipe.seeded_input(6, (2, 1, 128, 128)).cuda(); t = recipe.seeded_target(7, (2, 1, 128, 128)).cuda()
focal_dice_loss(m(x), t, alpha=1.0, gamma=2.0, ratio=0.3).backward()
torch.save({k: p.grad.cpu() for k, p in m.named_parameters()}, sys.argv[1])
''' % ROOT
    outs = []
    # last arm: the per-tap kernels everywhere they can stand in (UNETDC_IGEMM=dma) -- the input-normalising forward of the
    # 64-channel blocks exists in the lattice kernel only and must still be the one that runs (a kernel that ignored in_scale
    # would feed the raw pre-BatchNorm tensor into the second convolution: cosine far below the bar)
    arms = ({}, dict(SWITCH_SETS["unfused_epilogues"], UNETDC_FUSE_HEAD_BN="0"), {"UNETDC_IGEMM": "dma"})
    for i, extra in enumerate(arms):
        path = os.path.join("/tmp", f"unetdc_fallback_grads_{os.getpid()}_{i}.pt")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **extra), cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(torch.load(path, weights_only=True))
        os.remove(path)
    for other, bar in ((outs[1], 0.98), (outs[2], 0.95)):    # (other convolution kernels: other summation orders)
        for k in outs[0]:
            a, b = outs[0][k].double(), other[k].double()
            if k.endswith(".0.bias") or k.endswith(".3.bias"):
                continue                                        # structural zeros in front of train-mode BatchNorm
            cos = float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30))
            assert cos > bar, (k, cos)      # bf16 storage rounding makes the step chaotic at the 1e-2 level
