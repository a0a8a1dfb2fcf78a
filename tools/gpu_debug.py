import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tests import gpu_ops as G
from unet_dc_segmentation_amd import _lib
from unet_dc_segmentation_amd._lib import call
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)

# 1. apply kernel f32, identity affine, tiny
for dtype in ("f32", "bf16"):
    n, h, w, c = 1, 2, 2, 16
    y = torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) / 8 - 3
    yv = G.to_nhwc(y, dtype)
    print(dtype, "yv buffer\n", yv.float().cpu())
    av = G.empty_nhwc(n * h * w, c, dtype)
    sc = torch.ones(c, device="cuda"); sh = torch.zeros(c, device="cuda")
    call("unetdc_bn_relu_apply", yv.data_ptr(), yv.stride(0), sc.data_ptr(), sh.data_ptr(), av.data_ptr(), av.stride(0),
         None, 0, n, h, w, c, G.DT[dtype], G.stream())
    torch.cuda.synchronize()
    print(dtype, "av buffer\n", av.float().cpu())

# 2. first conv with delta weights: cout=8.. need cout in {8,..}: use 64, look at channel t for tap t
n, h, w, cin, cout = 1, 4, 6, 1, 64
x = torch.arange(h * w, dtype=torch.float32).reshape(1, 1, h, w)
wt = torch.zeros(cout, cin, 3, 3)
for t in range(9):
    wt[t, 0, t // 3, t % 3] = 1.0
ref = F.conv2d(x, wt, None, padding=1)
yv = G.empty_nhwc(n * h * w, cout, "f32")
xd, wd = x.cuda(), wt.cuda()
call("unetdc_conv3x3_first_fwd", xd.data_ptr(), wd.data_ptr(), None, None, None, yv.data_ptr(), yv.stride(0), None,
     n, h, w, cin, cout, 1, _lib.F32, G.stream())
torch.cuda.synchronize()
out = G.from_nhwc(yv, n, h, w)
for t in range(9):
    print("tap", t, "max err", float((out[0, t] - ref[0, t]).abs().max()))
print("out ch0\n", out[0, 0], "\nref ch0\n", ref[0, 0])
print("out ch9 (should be 0)\n", out[0, 9])
