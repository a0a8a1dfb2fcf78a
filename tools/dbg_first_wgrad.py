import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_e2e import build_model, otc
from utils.metrics_DC import focal_dice_loss
tag = sys.argv[1] if len(sys.argv) > 1 else "plain_c3"
model, g = build_model(tag, "train")
dil = dict(model.DILATIONS)
x, t = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_t"])
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
_, p64, g64 = otc.train_step_grads(x.double(), t.double(), sd64, dil)
model = model.cuda().train()
p = model(x.cuda())
loss = focal_dice_loss(p, t.cuda(), alpha=1.0, gamma=2.0, ratio=0.3)
loss.backward()
print("x", tuple(x.shape))
for k in ("enc1.0.weight", "enc1.3.weight", "enc1.1.weight"):
    a = dict(model.named_parameters())[k].grad.cpu().double(); r = g64[k]
    d = (a - r)
    print(k, "rel", float(d.norm() / r.norm()), "max|d|", float(d.abs().max()), "max|ref|", float(r.abs().max()))
    if k == "enc1.0.weight":
        dd = d.abs()
        print(" per-tap max|d|:", [f"{float(dd[:, :, i // 3, i % 3].max()):.2e}" for i in range(9)])
        print(" per-ci  max|d|:", [f"{float(dd[:, c].max()):.2e}" for c in range(dd.shape[1])])
        print(" worst channels:", torch.topk(dd.amax(dim=(1, 2, 3)), 5))
