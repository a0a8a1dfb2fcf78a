#!/usr/bin/env python3
"""Can an HBM-bound pass run BESIDE an MFMA kernel on a partitioned chip (CU-masked HIP streams) without costing the MFMA kernel
what the partition takes away?  (The MFMA kernels are held by power, r04_wgrad_stamps.txt: fewer CUs at a higher clock may deliver the
same FLOP/s.)  Probe library: the product sources with the persistent-grid CU count read from UNETDC_CUS (tools/probes/cu_mask_probe.sh).

    UNETDC_LIB=.../libunetdc_hip_probe.so python3 tools/probes/cu_mask_probe.py
"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import gpu_ops as G
from unet_dc_segmentation_amd import _lib

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]


def masked_stream(nibble):
    words = (ctypes.c_uint32 * 8)(*([int(f"{nibble:x}" * 8, 16)] * 8))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return s.value


torch.zeros(1, device="cuda")
n, h, w, cin, cout, d = 8, 256, 256, 128, 128, 1
g = torch.Generator().manual_seed(0)
x = torch.randn(n * h * w, cin, generator=g).bfloat16().cuda()
wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
wf, wd = G.pack_conv(wt, "bf16")
y = torch.empty(n * h * w, cout, dtype=torch.bfloat16, device="cuda")
bias = torch.zeros(cout, device="cuda")
rows = _lib.load().unetdc_conv3x3_stats_rows(n * h * w, cout)
stats = torch.empty((rows + 64) * 2 * cout, device="cuda")
# HBM-bound pass: BatchNorm apply + ReLU on a level-1 tensor (8 x 512 x 512 x 64 bf16 = 268 MB in, 268 MB out)
N1, H1, W1, C1 = 8, 512, 512, 64
ya = torch.randn(N1 * H1 * W1, C1, generator=g).bfloat16().cuda()
aa = torch.empty_like(ya)
sc, sh = (torch.rand(C1) + 0.5).cuda(), torch.randn(C1).cuda()
BF = G.DT["bf16"]
conv_fl = 2.0 * n * h * w * cin * cout * 9
app_bytes = 2.0 * ya.numel() * 2


def conv(s):
    _lib.call("unetdc_conv3x3_fwd", x.data_ptr(), cin, wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), cout,
              stats.data_ptr(), None, n, h, w, cin, cout, d, BF, s)


def app(s):
    _lib.call("unetdc_bn_relu_apply", ya.data_ptr(), C1, sc.data_ptr(), sh.data_ptr(), aa.data_ptr(), C1, None, 0, N1, H1, W1, C1,
              BF, s)


def run(label, cus, sconv, sapp, nconv, napp):
    os.environ["UNETDC_CUS"] = str(cus)
    streams = [s for s in (sconv, sapp) if s is not None]
    def loop(k):
        for i in range(k):
            if sconv is not None:
                for _ in range(nconv):
                    conv(sconv)
            if sapp is not None:
                for _ in range(napp):
                    app(sapp)
        for s in streams:
            hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        loop(10)
    t0 = time.perf_counter()
    K = 60
    loop(K)
    dt = time.perf_counter() - t0
    msg = f"{label:58s} {dt / K * 1e6:8.1f} us per round"
    if sconv is not None:
        msg += f" | conv {nconv * K * conv_fl / dt / 1e12:7.1f} TFLOP/s"
    if sapp is not None:
        msg += f" | apply {napp * K * app_bytes / dt / 1e12:5.2f} TB/s"
    print(msg, flush=True)


full = masked_stream(0xF)
big, small = masked_stream(0x7), masked_stream(0x8)          # 3 of every 4 CUs / the fourth
half_a, half_b = masked_stream(0x3), masked_stream(0xC)
run("conv alone, all 256 CUs", 256, full, None, 1, 0)
run("apply alone, all 256 CUs", 256, None, full, 0, 1)
run("conv then apply on ONE stream (as in the step)", 256, full, full, 1, 1)
run("conv alone on 192 CUs (grid for 192)", 192, big, None, 1, 0)
run("apply alone on 64 CUs", 256, None, small, 0, 1)
run("conv on 192 CUs BESIDE apply on 64 CUs", 192, big, small, 1, 1)
run("conv alone on 128 CUs (grid for 128)", 128, half_a, None, 1, 0)
run("apply alone on 128 CUs", 256, None, half_b, 0, 1)
run("conv on 128 CUs BESIDE apply on 128 CUs", 128, half_a, half_b, 1, 1)
run("conv BESIDE apply, both unmasked streams", 256, full, masked_stream(0xF), 1, 1)
