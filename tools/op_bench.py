#!/usr/bin/env python3
"""Single-operator microbenchmark (for rocprofv3 --pmc runs): repeats one C-ABI conv call.

    python tools/op_bench.py fwd 8 512 512 64 64 1 bf16 [reps]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gpu_ops as G
from unet_dc_segmentation_amd import _lib

op, n, h, w, cin, cout, d = sys.argv[1], *map(int, sys.argv[2:8])
dtype = sys.argv[8] if len(sys.argv) > 8 else "bf16"
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
g = torch.Generator().manual_seed(0)
x = torch.randn(n * h * w, cin, generator=g).to(G.TD[dtype]).cuda()
dy = torch.randn(n * h * w, cout, generator=g).to(G.TD[dtype]).cuda()
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
wf, wd = G.pack_conv(wt, dtype)
y = torch.empty(n * h * w, cout, dtype=G.TD[dtype], device="cuda")
dx = torch.empty(n * h * w, cin, dtype=G.TD[dtype], device="cuda")
bias = torch.zeros(cout, device="cuda")


def run():
    if op == "fwd":
        G.conv3x3_fwd(x, wf, bias, n, h, w, cin, cout, d, dtype, y, stats=True)
    elif op == "dgrad":
        G.conv3x3_dgrad(dy, wd, dx, n, h, w, cin, cout, d, dtype)
    else:
        G.conv3x3_wgrad(x, dy, n, h, w, cin, cout, d, dtype)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * h * w * cin * cout * 9
print(f"{op} {n}x{h}x{w} {cin}->{cout} d={d} {dtype}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s "
      f"(UNETDC_IGEMM={os.environ.get('UNETDC_IGEMM', '')})")
