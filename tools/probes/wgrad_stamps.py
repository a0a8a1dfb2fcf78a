#!/usr/bin/env python3
"""Reads the cycle stamps of the diagnostic weight-gradient build (tools/probes/wgrad_stamps.sh):
    UNETDC_LIB=.../libunetdc_hip_stamps.so python3 tools/probes/wgrad_stamps.py [n h w cin cout d]"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tests import gpu_ops as G
from unet_dc_segmentation_amd import _lib

n, h, w, cin, cout, d = (list(map(int, sys.argv[1:7])) if len(sys.argv) > 6 else [8, 64, 64, 512, 512, 1])
g = torch.Generator().manual_seed(0)
x = torch.randn(n * h * w, cin, generator=g).bfloat16().cuda()
dy = torch.randn(n * h * w, cout, generator=g).bfloat16().cuda()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:                       # steady-state clock
    for _ in range(20):
        G.conv3x3_wgrad(x, dy, n, h, w, cin, cout, d, "bf16")
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
G.conv3x3_wgrad(x, dy, n, h, w, cin, cout, d, "bf16")
e1.record()
torch.cuda.synchronize()
print(f"wgrad {n}x{h}x{w} {cin}->{cout} d={d}: {e0.elapsed_time(e1) * 1e3:.1f} us (kernel + reduce), last kernel {_lib.load().unetdc_last_kernel().decode()}")
buf = np.zeros((512, 8, 5), dtype=np.uint64)
lib = _lib.load()
lib.unetdc_dbg_wgrad_stamps.argtypes = [ctypes.c_void_p]
rc = lib.unetdc_dbg_wgrad_stamps(buf.ctypes.data)
assert rc == 0, rc
used = buf[:, :, 4] > 0
steps = buf[:, :, 4][used].astype(np.float64)
print(f"waves with data: {int(used.sum())}, steps per wave: {steps.min():.0f}..{steps.max():.0f}")
names = ["counted vmcnt wait", "workgroup barrier", "(loop incl. prologue, total)", "barrier -> end of step"]
for k in (0, 1, 3):
    per = buf[:, :, k][used].astype(np.float64) / steps
    print(f"  {names[k]:30s} mean {per.mean():8.0f}  p10 {np.percentile(per, 10):8.0f}  p90 {np.percentile(per, 90):8.0f} cycles per step")
tot = buf[:, :, 2][used].astype(np.float64)
print(f"  loop incl. prologue: mean {tot.mean():.0f} cycles = {(tot / steps).mean():.0f} per step")
for hv in (0, 1):
    sel = used.copy(); sel[:, (1 - hv) * 4:(1 - hv) * 4 + 4] = False
    if sel.any():
        print(f"  half {hv}: vmcnt {np.mean(buf[:, :, 0][sel] / buf[:, :, 4][sel]):.0f}  barrier {np.mean(buf[:, :, 1][sel] / buf[:, :, 4][sel]):.0f}  work {np.mean(buf[:, :, 3][sel] / buf[:, :, 4][sel]):.0f}")
for wv in range(8):
    sel = used[:, wv]
    if sel.any():
        print(f"    wave {wv}: vmcnt {np.mean(buf[:, wv, 0][sel] / buf[:, wv, 4][sel]):6.0f}  barrier {np.mean(buf[:, wv, 1][sel] / buf[:, wv, 4][sel]):6.0f}  work {np.mean(buf[:, wv, 3][sel] / buf[:, wv, 4][sel]):6.0f}")
