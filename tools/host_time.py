"""Host-side enqueue time of one training step vs the synchronised step time (is the step GPU-bound?):  python tools/host_time.py
Measured on the MI355X box: 3.7-4.2 ms of host work per step against 11.9 ms of GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.model_2 import UNetDC
from utils.metrics_DC import focal_dice_loss
from unet_dc_segmentation_amd.optim import FusedAdam
torch.manual_seed(0)
m = UNetDC(1, 1).cuda().train(); m.set_compute_dtype("bf16")
opt = FusedAdam(m, lr=1e-3)
x = torch.rand(8, 1, 512, 512, device="cuda"); t = (torch.rand(8, 1, 512, 512, device="cuda") > 0.7).float()
def step():
    opt.zero_grad(set_to_none=True)
    loss = focal_dice_loss(m(x), t, alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step, with sync {1e3*(t2-t0)/20:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
