#!/usr/bin/env python3
"""Composed decoder up-path vs the explicit chain, per decoder level of the bs-8 512 x 512 network (GPU box, repo root):

    python tools/upcomp_bench.py [levels, e.g. 1,2,3,4] [reps]

forward : unetdc_convT2x2_fwd + unetdc_conv3x3_fwd over the [pixels, 2C] concat (statistics mode)   vs   unetdc_upcomp_fwd
(dgrad / wgrad lines appear as those entry points exist in the library)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests import gpu_ops as G
from unet_dc_segmentation_amd import _lib
from unet_dc_segmentation_amd._lib import call

levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = 8
BF = G.DT["bf16"]
lib = _lib.load()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:                       # steady-state clocks
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lvl in levels:
    c = 64 << (lvl - 1)
    hlo = wlo = 256 >> (lvl - 1)
    g = torch.Generator().manual_seed(lvl)
    bf = torch.bfloat16
    plo, phi = N * hlo * wlo, N * 4 * hlo * wlo
    h = torch.randn(plo, 2 * c, generator=g).to(bf).cuda()
    cat = torch.randn(phi, 2 * c, generator=g).to(bf).cuda()
    y = torch.empty(phi, c, dtype=bf, device="cuda")
    w3 = torch.randn(c, 2 * c, 3, 3, generator=g) / (3 * (2 * c) ** 0.5)
    wt = torch.randn(2 * c, c, 2, 2, generator=g) / (2 * (2 * c) ** 0.5)
    b3, bt = torch.randn(c, generator=g).cuda(), torch.randn(c, generator=g).cuda()
    w3f, w3d = G.pack_conv(w3, "bf16")
    wtf, wtd = G.pack_convT(wt, "bf16")
    rows = lib.unetdc_conv3x3_stats_rows(phi, c)
    st = torch.empty((rows + 64) * 3 * 2 * c, device="cuda")
    live = ctypes.c_int(0)
    wc_f = torch.empty(16 * c * 2 * c, dtype=bf, device="cuda")
    wc_d = torch.empty_like(wc_f)
    ws_f = torch.empty(9 * c * c, dtype=bf, device="cuda")
    ws_d = torch.empty_like(ws_f)
    btab = torch.empty(10 * c, device="cuda")
    w3m = w3.cuda().contiguous()

    def compose():
        call("unetdc_upcomp_compose", w3f.data_ptr(), w3d.data_ptr(), wtd.data_ptr(), w3m.data_ptr(), b3.data_ptr(), bt.data_ptr(),
             wc_f.data_ptr(), wc_d.data_ptr(), ws_f.data_ptr(), ws_d.data_ptr(), btab.data_ptr(), c, BF, G.stream())

    def explicit_fwd():
        call("unetdc_convT2x2_fwd", h.data_ptr(), 2 * c, wtf.data_ptr(), bt.data_ptr(), cat.data_ptr(), 2 * c, N, hlo, wlo, 2 * c, c,
             BF, G.stream())
        call("unetdc_conv3x3_fwd", cat.data_ptr(), 2 * c, w3f.data_ptr(), b3.data_ptr(), None, None, y.data_ptr(), c, st.data_ptr(),
             ctypes.byref(live), N, 2 * hlo, 2 * wlo, 2 * c, c, 1, BF, G.stream())

    def composed_fwd():
        call("unetdc_upcomp_fwd", cat.data_ptr() + 2 * c, 2 * c, ws_f.data_ptr(), btab.data_ptr(), h.data_ptr(), 2 * c,
             wc_f.data_ptr(), y.data_ptr(), c, st.data_ptr(), ctypes.byref(live), N, hlo, wlo, c, BF, G.stream())

    # backward, input side
    dy = torch.randn(phi, c, generator=g).to(bf).cuda()
    dcat = torch.empty(phi, 2 * c, dtype=bf, device="cuda")
    dh = torch.empty(plo, 2 * c, dtype=bf, device="cuda")
    yprev = torch.randn(plo, 2 * c, generator=g).to(bf).cuda()
    f32 = dict(device="cuda", dtype=torch.float32)
    sc, sh, mu, rs = torch.rand(2 * c, **f32) + 0.5, torch.randn(2 * c, **f32) * 0.3, torch.randn(2 * c, **f32) * 0.1, torch.rand(2 * c, **f32) + 0.5
    prow = lib.unetdc_conv3x3_stats_rows(plo, 2 * c)
    parts = torch.empty((prow + 64) * 3 * 2 * c, **f32)
    npart = ctypes.c_int(0)
    colsum = torch.empty(c, **f32)
    wsb = lib.unetdc_conv3x3_dgrad_colsum_workspace(N, 2 * hlo, 2 * wlo, 2 * c)
    ws = G.workspace(wsb)

    def explicit_dgrad():
        call("unetdc_conv3x3_dgrad_colsum", dy.data_ptr(), c, w3d.data_ptr(), dcat.data_ptr(), 2 * c, colsum.data_ptr(), 0, c, ws.data_ptr(),
             wsb, N, 2 * hlo, 2 * wlo, 2 * c, c, 1, BF, G.stream())
        call("unetdc_convT2x2_dgrad_bnstats", dcat.data_ptr(), 2 * c, wtd.data_ptr(), dh.data_ptr(), 2 * c, yprev.data_ptr(), 2 * c,
             sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), parts.data_ptr(), parts.numel(), ctypes.byref(npart), N, hlo, wlo,
             2 * c, c, BF, G.stream())

    def composed_dgrad():
        call("unetdc_conv3x3_dgrad", dy.data_ptr(), c, ws_d.data_ptr(), dcat.data_ptr() + 2 * c, 2 * c, N, 2 * hlo, 2 * wlo, c, c, 1, BF,
             G.stream())
        call("unetdc_upcomp_dgrad_bnstats", dy.data_ptr(), c, wc_d.data_ptr(), dh.data_ptr(), 2 * c, yprev.data_ptr(), 2 * c, sc.data_ptr(),
             sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), parts.data_ptr(), parts.numel(), ctypes.byref(npart), N, hlo, wlo, c, BF, G.stream())

    # backward, weight side
    dw3 = torch.empty(c, 2 * c, 3, 3, device="cuda")
    dwt = torch.empty(2 * c, c, 2, 2, device="cuda")
    dbt = torch.empty(c, device="cuda")
    total = torch.zeros(c, device="cuda")
    wb1 = lib.unetdc_conv3x3_wgrad_workspace(N, 2 * hlo, 2 * wlo, 2 * c, c, BF)
    wb2 = lib.unetdc_convT2x2_wgrad_workspace(N, hlo, wlo, 2 * c, c, BF)
    wb3 = lib.unetdc_upcomp_wgrad_workspace(N, hlo, wlo, c, BF)
    wws = G.workspace(max(wb1, wb2, wb3))

    def explicit_wgrad():
        call("unetdc_conv3x3_wgrad", cat.data_ptr(), 2 * c, dy.data_ptr(), c, dw3.data_ptr(), wws.data_ptr(), wb1, N, 2 * hlo, 2 * wlo,
             2 * c, c, 1, BF, G.stream())
        call("unetdc_convT2x2_wgrad", h.data_ptr(), 2 * c, dcat.data_ptr(), 2 * c, dwt.data_ptr(), wws.data_ptr(), wb2, N, hlo, wlo, 2 * c, c,
             BF, G.stream())

    def composed_wgrad():
        call("unetdc_upcomp_wgrad", h.data_ptr(), 2 * c, cat.data_ptr() + 2 * c, 2 * c, dy.data_ptr(), c, wtf.data_ptr(), w3d.data_ptr(),
             w3m.data_ptr(), bt.data_ptr(), total.data_ptr(), dw3.data_ptr(), dwt.data_ptr(), dbt.data_ptr(), wws.data_ptr(), wb3, N, hlo, wlo,
             c, BF, G.stream())

    compose()
    t_comp = timed(compose)
    t_exp, t_cmp = timed(explicit_fwd), timed(composed_fwd)
    d_exp, d_cmp = timed(explicit_dgrad), timed(composed_dgrad)
    w_exp, w_cmp = timed(explicit_wgrad), timed(composed_wgrad)
    print(f"level {lvl} (C = {c}, low-res {hlo} x {wlo}): compose {t_comp:6.1f} us | forward explicit {t_exp:7.1f} us, composed {t_cmp:7.1f} us "
          f"({t_exp - t_cmp:+.1f}) | dgrad explicit {d_exp:7.1f} us, composed {d_cmp:7.1f} us ({d_exp - d_cmp:+.1f}) | wgrad explicit "
          f"{w_exp:7.1f} us, composed {w_cmp:7.1f} us ({w_exp - w_cmp:+.1f})", flush=True)
