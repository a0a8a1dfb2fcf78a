#!/usr/bin/env python3
"""Single-operator microbenchmark (for rocprofv3 --pmc runs): repeats one C-ABI conv call.

    python tools/op_bench.py fwd 8 512 512 64 64 1 bf16 [reps]
    python tools/op_bench.py convt_fwd 8 256 256 128 64 1 bf16      (h, w = INPUT size; also convt_dgrad, convt_wgrad)
    python tools/op_bench.py fwd_bnin 8 512 512 64 64 1 bf16        (also dgrad_bnstats, wgrad_bnin: the level-1 forms)
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
ZERO = os.environ.get("ZERO") == "1"          # all-zero operands: same instruction stream, far less switching power
x = torch.randn(n * h * w, cin, generator=g).to(G.TD[dtype]).cuda()
dy = torch.randn(n * h * w, cout, generator=g).to(G.TD[dtype]).cuda()
if ZERO:
    x.zero_(); dy.zero_()
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
if ZERO:
    wt.zero_()
wf, wd = G.pack_conv(wt, dtype)
y = torch.empty(n * h * w, cout, dtype=G.TD[dtype], device="cuda")
dx = torch.empty(n * h * w, cin, dtype=G.TD[dtype], device="cuda")
bias = torch.zeros(cout, device="cuda")
DTI = G.DT[dtype]
is_t = op.startswith("convt")
if is_t:
    wtt = torch.randn(cin, cout, 2, 2, generator=g) / (2 * cin ** 0.5)
    twf, twd = G.pack_convT(wtt, dtype)
    up = torch.empty(n * 4 * h * w, 2 * cout, dtype=G.TD[dtype], device="cuda")       # concat buffer, first half written
    dup = torch.randn(n * 4 * h * w, 2 * cout, generator=g).to(G.TD[dtype]).cuda()
    tdx = torch.empty(n * h * w, cin, dtype=G.TD[dtype], device="cuda")
    tdw = torch.empty(cin, cout, 2, 2, device="cuda")
    tws_bytes = _lib.load().unetdc_convT2x2_wgrad_workspace(n, h, w, cin, cout, G.DT[dtype])
    tws = G.workspace(tws_bytes)


if op in ("fwd_bnin", "dgrad_bnstats", "wgrad_bnin"):
    import ctypes
    f32 = dict(device="cuda", dtype=torch.float32)
    sc_in, sh_in = torch.rand(cin, **f32) + 0.5, torch.randn(cin, **f32) * 0.3
    mu_in, rs_in = torch.randn(cin, **f32) * 0.1, torch.rand(cin, **f32) + 0.5
    rows = _lib.load().unetdc_conv3x3_stats_rows(n * h * w, max(cin, cout))
    stats = torch.empty((rows + 64) * 3 * max(cin, cout), **f32)
    yprev = torch.randn(n * h * w, cin, generator=g).to(G.TD[dtype]).cuda()
    npart = ctypes.c_int(0)
    wws_bytes = _lib.load().unetdc_conv3x3_wgrad_workspace(n, h, w, cin, cout, DTI)
    wws = G.workspace(wws_bytes)
    dwt = torch.empty(cout, cin, 3, 3, device="cuda")
    # 128-channel n-blocks: the input-normalising forward also stores the normalised activation (what the engine asks of it)
    act = torch.empty(n * h * w, cin, dtype=G.TD[dtype], device="cuda") \
        if op == "fwd_bnin" and _lib.load().unetdc_conv3x3_bnin_supported(n, h, w, cin, cout, d, DTI) == 2 else None


def run():
    if op == "fwd_bnin":
        _lib.call("unetdc_conv3x3_fwd_bnin", x.data_ptr(), cin, sc_in.data_ptr(), sh_in.data_ptr(), wf.data_ptr(), bias.data_ptr(),
                  y.data_ptr(), cout, stats.data_ptr(), None, None if act is None else act.data_ptr(), 0 if act is None else cin,
                  n, h, w, cin, cout, d, DTI, G.stream())
    elif op == "dgrad_bnstats":
        _lib.call("unetdc_conv3x3_dgrad_bnstats", dy.data_ptr(), cout, wd.data_ptr(), dx.data_ptr(), cin, yprev.data_ptr(), cin,
                  sc_in.data_ptr(), sh_in.data_ptr(), mu_in.data_ptr(), rs_in.data_ptr(), stats.data_ptr(), stats.numel(),
                  ctypes.byref(npart), n, h, w, cin, cout, d, DTI, G.stream())
    elif op == "wgrad_bnin":
        _lib.call("unetdc_conv3x3_wgrad_bnin", x.data_ptr(), cin, sc_in.data_ptr(), sh_in.data_ptr(), dy.data_ptr(), cout,
                  dwt.data_ptr(), wws.data_ptr(), wws_bytes, n, h, w, cin, cout, d, DTI, G.stream())
    elif op == "fwd":
        G.conv3x3_fwd(x, wf, bias, n, h, w, cin, cout, d, dtype, y, stats=True)
    elif op == "convt_fwd":
        _lib.call("unetdc_convT2x2_fwd", x.data_ptr(), cin, twf.data_ptr(), bias.data_ptr(), up.data_ptr(), 2 * cout,
                  n, h, w, cin, cout, DTI, G.stream())
    elif op == "convt_dgrad":
        _lib.call("unetdc_convT2x2_dgrad", dup.data_ptr(), 2 * cout, twd.data_ptr(), tdx.data_ptr(), cin,
                  n, h, w, cin, cout, DTI, G.stream())
    elif op == "convt_wgrad":
        _lib.call("unetdc_convT2x2_wgrad", x.data_ptr(), cin, dup.data_ptr(), 2 * cout, tdw.data_ptr(), tws.data_ptr(),
                  tws_bytes, n, h, w, cin, cout, DTI, G.stream())
    elif op == "dgrad":
        G.conv3x3_dgrad(dy, wd, dx, n, h, w, cin, cout, d, dtype)
    else:
        G.conv3x3_wgrad(x, dy, n, h, w, cin, cout, d, dtype)


for _ in range(3):
    run()
torch.cuda.synchronize()
# steady state: the first milliseconds after idle run at a lower clock (the same launch reads 131.7 us in a cold 30-launch loop and
# 111.9 us after 2 s of back-to-back launches, profiles/r04_power_probes.txt; MI355X_MICROARCH.md, DVFS item 6)
import time
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < float(os.environ.get("OPB_WARM_S", "1.5")):
    for _ in range(50):
        run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * h * w * cin * cout * (4 if is_t else 9)
print(f"{op} {n}x{h}x{w} {cin}->{cout} d={d} {dtype}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s "
      f"(UNETDC_IGEMM={os.environ.get('UNETDC_IGEMM', '')})")
