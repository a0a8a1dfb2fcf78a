"""Where do the per-step device copies (rocprofv3: __amd_rocclr_copyBuffer) come from?  Runs three training steps under the
PyTorch profiler and prints every memcpy / memset activity of the LAST step with the Python frames that issued it.
GPU box, repo root:  python3 tools/find_copies.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from models.model_2 import UNetDC  # noqa: E402
from unet_dc_segmentation_amd.optim import FusedAdam  # noqa: E402
from utils.metrics_DC import focal_dice_loss  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = UNetDC(1, 1).to(dev).train()
model.set_compute_dtype("bf16")
opt = FusedAdam(model, lr=1e-3)
x, t = synthetic_batch(1, 8, 512, 512, 1)
x, t = x.to(dev), t.to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = focal_dice_loss(model(x), t, alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
n = 0
for ev in prof.events():
    name = ev.name.lower()
    if "memcpy" in name or "memset" in name or "copy_" in name or "aten::to" in name or "_to_copy" in name or "fill_" in name:
        n += 1
        stack = [f for f in (ev.stack or []) if "site-packages/torch" not in f][:4]
        print(f"{ev.name:40s} dev {ev.device_type} dur {ev.device_time_total if hasattr(ev, 'device_time_total') else 0:8.1f} us  shapes {ev.input_shapes}  {stack}")
print("events listed:", n)
