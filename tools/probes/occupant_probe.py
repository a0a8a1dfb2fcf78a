#!/usr/bin/env python3
"""What does a communication kernel that sits on some CUs do to a training step built from persistent compute grids?
(round-3 verdict item 6c; one-GPU stand-in for the ring all-reduce kernels of a multi-GPU run)

    python3 tools/probes/occupant_probe.py            (needs tools/probes/liboccupant.so, see occupant.hip)

Per step, `bursts` occupant launches of `us` microseconds each are enqueued on a SIDE stream in front of the step (they start
with the step and run back to back): nwg workgroups x 256 threads x 48 KB LDS that hold their slots and do nothing.
Reported: ms per step (mean of 20 steps after 5 warm-up steps) for each (nwg, bursts x us).
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from models.model_2 import UNetDC
from utils.metrics_DC import focal_dice_loss
from unet_dc_segmentation_amd.optim import FusedAdam

occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboccupant.so"))
occ.occupant_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = UNetDC(1, 1).to(dev).train()
model.set_compute_dtype("bf16")
opt = FusedAdam(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 1, 512, 512, generator=g).to(dev)
t = (torch.rand(8, 1, 512, 512, generator=g) < 0.3).float().to(dev)
side = torch.cuda.Stream(device=dev)
GHZ = 2.0                                                    # s_memtime ticks per ns, roughly (the bursts need not be exact)


def step():
    opt.zero_grad(set_to_none=True)
    loss = focal_dice_loss(model(x), t, alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    opt.step()


def measure(nwg, bursts, us, threads=256, lds=48 * 1024):
    def one():
        if nwg:
            side.wait_stream(torch.cuda.current_stream())   # the bursts start with the step
            for _ in range(bursts):
                rc = occ.occupant_launch(nwg, threads, lds, int(us * 1e3 * GHZ), ctypes.c_void_p(side.cuda_stream))
                assert rc == 0, rc
        step()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


step(); step()
torch.cuda.synchronize()
base = measure(0, 0, 0)
print(f"no occupant                                   {base:7.3f} ms/step", flush=True)
for bursts, us in ((7, 150), (40, 250)):
    for nwg in (8, 16, 32, 64, 128):
        ms = measure(nwg, bursts, us)
        print(f"{nwg:4d} workgroups x 256 threads, {bursts:2d} x {us} us per step   {ms:7.3f} ms/step  ({ms - base:+.3f})", flush=True)
print(f"no occupant (again)                           {measure(0, 0, 0):7.3f} ms/step", flush=True)
