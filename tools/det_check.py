"""Run the bf16 train step twice on identical inputs and list the gradient tensors that differ (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.model_2 import UNetDC
from utils.metrics_DC import focal_dice_loss

bs, size = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(5)
model = UNetDC(1, 1).cuda().train()
model.set_compute_dtype(sys.argv[3] if len(sys.argv) > 3 else "bf16")
g = torch.Generator().manual_seed(6)
x = torch.rand(bs, 1, size, size, generator=g).cuda()
t = (torch.rand(bs, 1, size, size, generator=g) > 0.7).float().cuda()
snaps = []
for _ in range(3):
    model.zero_grad(set_to_none=True)
    loss = focal_dice_loss(model(x), t, alpha=1.0, gamma=2.0, ratio=0.3)
    loss.backward()
    snaps.append((loss.item(), {k: p.grad.clone() for k, p in model.named_parameters()}))
print("loss", [s[0] for s in snaps])
for k in snaps[0][1]:
    a, b, c = (s[1][k] for s in snaps)
    if not (torch.equal(a, b) and torch.equal(a, c)):
        nd = int((a != b).sum()), int((a != c).sum())
        print(f"DIFF {k:28s} n_diff={nd} of {a.numel()} max|d|={float((a - b).abs().max()):.3e} ref max={float(a.abs().max()):.3e}")
print("done")
