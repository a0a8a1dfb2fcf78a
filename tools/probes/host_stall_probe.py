"""Where does the HOST spend a 30-50 ms pause once every few dozen training steps?  (round-4 driver run: one 48.9 ms device
step at timed step 17; round 5, collector disabled: the same step index shows 28.5 ms of host time and no device stall.)

    python tools/probes/host_stall_probe.py [steps=160] [mode]

Per step: host time of each phase (zero_grad, forward, loss, backward, optimizer) and of every C-ABI call (perf_counter around
the ctypes call: a launch that blocks inside the HIP runtime shows up under its entry point), the device time of the step (events
created up front), and the cumulative number of kernel launches / kernel-argument bytes is not known to Python -- the C-ABI call
count is printed instead.  Prints every step whose host time exceeds 4x the median with its slowest calls, and the distances
between such steps (a fixed period in launches points at a ring buffer of the runtime: kernel arguments, signals, AQL packets).
mode = "sync": torch.cuda.synchronize() after every step (host never runs ahead: does the pause still happen, and is it then a
device pause as well?).
mode = "events": the round-4 bench condition -- _lib.start_timing() brackets every implicit-GEMM call with a pair of timing events
created inside the loop (92 per step, all kept alive), no per-call host timing: does the DEVICE stall when the launch queue's
back-pressure first sets in?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gc

import torch

from models.model_2 import UNetDC
from unet_dc_segmentation_amd import _lib
from unet_dc_segmentation_amd.optim import FusedAdam
from utils.metrics_DC import focal_dice_loss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
mode = sys.argv[2] if len(sys.argv) > 2 else "free"
torch.manual_seed(0)
m = UNetDC(1, 1).cuda().train()
m.set_compute_dtype("bf16")
opt = FusedAdam(m, lr=1e-3)
x = torch.rand(8, 1, 512, 512, device="cuda")
t = (torch.rand(8, 1, 512, 512, device="cuda") > 0.7).float()

calls = []            # (name, host seconds) of the current step
ncalls = [0]
lib = _lib.load()
orig_call = _lib.call


def timed_call(name, *args):
    t0 = time.perf_counter()
    rc = getattr(lib, name)(*args)
    dt = time.perf_counter() - t0
    calls.append((name, dt))
    ncalls[0] += 1
    _lib.check(rc, name)


import unet_dc_segmentation_amd.engine as eng_mod
import unet_dc_segmentation_amd.loss as loss_mod

if mode != "events":
    _lib.call = eng_mod.call = loss_mod.call = timed_call


def step(rec):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    p = m(x)
    t2 = time.perf_counter()
    loss = focal_dice_loss(p, t, alpha=1.0, gamma=2.0, ratio=0.3)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    rec.update(zero=t1 - t0, fwd=t2 - t1, loss=t3 - t2, bwd=t4 - t3, opt=t5 - t4, total=t5 - t0)


for _ in range(6):
    step({})
torch.cuda.synchronize()
gc.collect()
gc.freeze()
gc.disable()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
recs = []
launch0 = ncalls[0]
if mode == "events":
    import bench
    _lib.start_timing(bench.IGEMM_CALLS)
marks[0].record()
for i in range(steps):
    calls.clear()
    r = {"calls_before": ncalls[0] - launch0}
    step(r)
    r["slow"] = sorted(calls, key=lambda c: -c[1])[:3]
    r["ncalls"] = len(calls)
    marks[i + 1].record()
    if mode == "sync":
        torch.cuda.synchronize()
    recs.append(r)
torch.cuda.synchronize()
if mode == "events":
    _lib.stop_timing()
dev = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
tot = sorted(r["total"] for r in recs)
med = tot[len(tot) // 2]
dmed = sorted(dev)[len(dev) // 2]
print(f"mode {mode}: {steps} steps, host median {med * 1e3:.2f} ms/step, device median {dmed:.2f} ms/step, "
      f"{recs[0]['ncalls']} C-ABI calls per step")
if mode == "events":
    print("host ms per step:", [round(r["total"] * 1e3, 1) for r in recs])
    print("device ms per step:", [round(v, 1) for v in dev])
slow = [i for i, r in enumerate(recs) if r["total"] > 4 * med]
for i in slow:
    r = recs[i]
    print(f"step {i:4d}: host {r['total'] * 1e3:7.2f} ms (zero {r['zero'] * 1e3:.2f} fwd {r['fwd'] * 1e3:.2f} loss {r['loss'] * 1e3:.2f} "
          f"bwd {r['bwd'] * 1e3:.2f} opt {r['opt'] * 1e3:.2f}); device {dev[i]:.2f} ms (next {dev[i + 1] if i + 1 < steps else 0:.2f}); "
          f"C-ABI calls before it {r['calls_before']}; slowest calls: "
          + ", ".join(f"{n} {d * 1e3:.2f} ms" for n, d in r["slow"]))
print("slow host steps at", slow, "distances", [b - a for a, b in zip(slow, slow[1:])])
dslow = [i for i, d in enumerate(dev) if d > 1.5 * dmed]
print("slow DEVICE steps at", dslow, [round(dev[i], 2) for i in dslow])
