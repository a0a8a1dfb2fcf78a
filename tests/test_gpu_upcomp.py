"""GPU parity of the composed decoder up-path (round 5): conv3x3(W3[:, :C]) o convT2x2(WT) evaluated on the low-res tensor
(unetdc_upcomp_*) against the explicit chain the reference runs -- ConvTranspose2d, torch.cat, Conv2d
(/root/reference/models/model_2.py:67-69) -- in PyTorch-CPU fp32 on the same bf16-rounded operands.

Rounding points differ from the explicit bf16 path (no `up` tensor: the composed weights W' and the partial sum of the skip half
are rounded to bf16 instead), so the bar is the bf16 operator bar of tests/test_gpu_ops.py, not bit equality."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests import gpu_ops as G
    from unet_dc_segmentation_amd import _lib
    from unet_dc_segmentation_amd._lib import call


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def gen(seed):
    return torch.Generator().manual_seed(seed)


def composed_weights(w3, wt, b3, bt, c):
    """pack both layers, run unetdc_upcomp_compose; returns the device buffers"""
    bf = torch.bfloat16
    w3f, w3d = G.pack_conv(w3, "bf16")
    wtf, wtd = G.pack_convT(wt, "bf16")
    wc_f = torch.full((16 * c * 2 * c,), float("nan"), dtype=bf, device="cuda")
    wc_d = torch.full((16 * c * 2 * c,), float("nan"), dtype=bf, device="cuda")
    ws_f = torch.full((9 * c * c,), float("nan"), dtype=bf, device="cuda")
    ws_d = torch.full((9 * c * c,), float("nan"), dtype=bf, device="cuda")
    btab = torch.full((10 * c,), float("nan"), device="cuda")
    w3m, b3d, btd = w3.cuda().contiguous(), b3.cuda(), bt.cuda()
    call("unetdc_upcomp_compose", w3f.data_ptr(), w3d.data_ptr(), wtd.data_ptr(), w3m.data_ptr(), b3d.data_ptr(), btd.data_ptr(),
         wc_f.data_ptr(), wc_d.data_ptr(), ws_f.data_ptr(), ws_d.data_ptr(), btab.data_ptr(), c, G.DT["bf16"], G.stream())
    return dict(w3f=w3f, w3d=w3d, wtf=wtf, wtd=wtd, wc_f=wc_f, wc_d=wc_d, ws_f=ws_f, ws_d=ws_d, btab=btab, keep=(w3m, b3d, btd))


def reference_wprime(w3, wt, c):
    """W'[phase][t][co][ci] from the definition (fp64)"""
    w3u, wtd = w3[:, :c].double(), wt.double()                    # [co][c][ky][kx], [ci][c][a][b]
    out = torch.zeros(4, 4, c, 2 * c, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            for ky in range(3):
                for kx in range(3):
                    ty, a = ((py + ky + 1) >> 1) - py, (py + ky + 1) & 1
                    tx, b = ((px + kx + 1) >> 1) - px, (px + kx + 1) & 1
                    out[py * 2 + px, ty * 2 + tx] += torch.einsum("oc,ic->oi", w3u[:, :, ky, kx], wtd[:, :, a, b])
    return out


@pytest.mark.parametrize("c", [64, 128])
def test_composed_weights_and_bias_table(c):
    g = gen(11)
    w3 = G.quant(torch.randn(c, 2 * c, 3, 3, generator=g) / (3 * (2 * c) ** 0.5), "bf16")
    wt = G.quant(torch.randn(2 * c, c, 2, 2, generator=g) / (2 * (2 * c) ** 0.5), "bf16")
    b3, bt = torch.randn(c, generator=g), torch.randn(c, generator=g)
    W = composed_weights(w3, wt, b3, bt, c)
    ref = reference_wprime(w3, wt, c)
    got_f = W["wc_f"].float().cpu().reshape(4, 4, c, 2 * c)
    assert rel(got_f, ref) < 3e-3                                    # one bf16 rounding of the fp32-accumulated product
    got_d = W["wc_d"].float().cpu().reshape(4, 4, 2 * c, c)          # [phase][3 - t][ci][co]
    assert torch.equal(got_d.flip(1).transpose(2, 3), got_f)
    # skip-half slices: exact copies
    assert torch.equal(W["ws_f"].float().cpu().reshape(9, c, c), w3[:, c:].permute(2, 3, 0, 1).reshape(9, c, c))
    assert torch.equal(W["ws_d"].float().cpu().reshape(9, c, c), w3[:, c:].flip(2, 3).permute(2, 3, 1, 0).reshape(9, c, c))
    # bias table
    btap = torch.einsum("octk,c->tko", w3[:, :c].double().reshape(c, c, 3, 3), bt.double()).reshape(9, c)
    tab = W["btab"].cpu().double().reshape(10, c)
    np.testing.assert_allclose(tab[1:].numpy(), btap.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tab[0].numpy(), (b3.double() + btap.sum(0)).numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", [(2, 16, 32, 64), (1, 8, 64, 128), (3, 24, 32, 64), (8, 32, 32, 128), (2, 64, 64, 256)])
def test_composed_forward_matches_convtranspose_cat_conv(case):
    n, hlo, wlo, c = case
    lib = _lib.load()
    assert lib.unetdc_upcomp_supported(n, hlo, wlo, c, G.DT["bf16"]) == 1
    g = gen(12)
    h = G.quant(torch.randn(n, 2 * c, hlo, wlo, generator=g), "bf16")
    skip = G.quant(torch.randn(n, c, 2 * hlo, 2 * wlo, generator=g), "bf16")
    w3 = G.quant(torch.randn(c, 2 * c, 3, 3, generator=g) / (3 * (2 * c) ** 0.5), "bf16")
    wt = G.quant(torch.randn(2 * c, c, 2, 2, generator=g) / (2 * (2 * c) ** 0.5), "bf16")
    b3, bt = torch.randn(c, generator=g) * 0.5, torch.randn(c, generator=g)             # bT != 0: the border classes matter
    up = F.conv_transpose2d(h, wt, bt, stride=2)
    y_ref = F.conv2d(torch.cat([up, skip], 1), w3, b3, padding=1)
    W = composed_weights(w3, wt, b3, bt, c)
    npix = n * 4 * hlo * wlo
    hv = G.to_nhwc(h, "bf16", ld=2 * c + 64, off=64)
    sv = G.to_nhwc(skip, "bf16", ld=2 * c, off=c)                    # the second half of a concat buffer
    yv = G.empty_nhwc(npix, c, "bf16", ld=c + 64, off=0)
    rows = lib.unetdc_conv3x3_stats_rows(npix, c)
    st = torch.full(((rows + 64) * 2 * c,), float("nan"), device="cuda")
    live = ctypes.c_int(-1)
    call("unetdc_upcomp_fwd", sv.data_ptr(), sv.stride(0), W["ws_f"].data_ptr(), W["btab"].data_ptr(), hv.data_ptr(), hv.stride(0),
         W["wc_f"].data_ptr(), yv.data_ptr(), yv.stride(0), st.data_ptr(), ctypes.byref(live), n, hlo, wlo, c, G.DT["bf16"], G.stream())
    y = G.from_nhwc(yv, n, 2 * hlo, 2 * wlo)
    e = rel(y, y_ref)
    # border pixels on their own (the ConvT bias does not reach them through the taps that leave the image)
    border = torch.ones_like(y_ref, dtype=torch.bool)
    border[:, :, 1:-1, 1:-1] = False
    eb = rel(y[border], y_ref[border])
    print(f"[upcomp fwd {case}] rel-L2 {e:.2e}, border pixels {eb:.2e}")
    assert e < 8e-3 and eb < 8e-3, (e, eb)
    # statistics rows: sums of the values as stored
    assert 1 <= live.value <= rows
    sta = st.cpu()[: rows * 2 * c].reshape(rows, 2, c)
    assert torch.equal(sta[live.value:], torch.zeros_like(sta[live.value:]))
    stc = sta.double().sum(0)
    yq = y.double()
    np.testing.assert_allclose(stc[0].numpy(), yq.sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(stc[1].numpy(), (yq * yq).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=2e-2)
    # run-to-run: bitwise
    yv2 = G.empty_nhwc(npix, c, "bf16", ld=c + 64, off=0)
    st2 = torch.full_like(st, float("nan"))
    call("unetdc_upcomp_fwd", sv.data_ptr(), sv.stride(0), W["ws_f"].data_ptr(), W["btab"].data_ptr(), hv.data_ptr(), hv.stride(0),
         W["wc_f"].data_ptr(), yv2.data_ptr(), yv2.stride(0), st2.data_ptr(), ctypes.byref(live), n, hlo, wlo, c, G.DT["bf16"], G.stream())
    assert torch.equal(yv, yv2) and torch.equal(st[: rows * 2 * c], st2[: rows * 2 * c])


@pytest.mark.parametrize("case", [(2, 16, 32, 64), (1, 8, 64, 128), (3, 24, 32, 64), (8, 32, 32, 128), (2, 64, 64, 256)])
def test_composed_input_gradient_and_fused_bn_backward_sums(case):
    """unetdc_upcomp_dgrad_bnstats: dL/dh through conv3x3(up half) o convT2x2 vs autograd of the explicit chain; the skip half of the
    concat gradient through the ordinary dgrad with the sliced image; the fused BatchNorm-backward sums from the gradient as stored."""
    n, hlo, wlo, c = case
    g = gen(13)
    w3 = G.quant(torch.randn(c, 2 * c, 3, 3, generator=g) / (3 * (2 * c) ** 0.5), "bf16")
    wt = G.quant(torch.randn(2 * c, c, 2, 2, generator=g) / (2 * (2 * c) ** 0.5), "bf16")
    b3, bt = torch.randn(c, generator=g), torch.randn(c, generator=g)
    dy = G.quant(torch.randn(n, c, 2 * hlo, 2 * wlo, generator=g), "bf16")
    yprev = G.quant(torch.randn(n, 2 * c, hlo, wlo, generator=g) * 1.5 + 0.3, "bf16")       # saved conv output of the stage that produced h
    gamma, beta = torch.rand(2 * c, generator=g) + 0.5, torch.randn(2 * c, generator=g) * 0.2
    hr = torch.zeros(n, 2 * c, hlo, wlo, requires_grad=True)
    sr = torch.zeros(n, c, 2 * hlo, 2 * wlo, requires_grad=True)
    y = F.conv2d(torch.cat([F.conv_transpose2d(hr, wt, bt, stride=2), sr], 1), w3, b3, padding=1)
    dh_ref, dskip_ref = torch.autograd.grad(y, (hr, sr), dy)
    yd = yprev.double()
    mean, var = yd.mean(dim=(0, 2, 3)), yd.var(dim=(0, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma.double() * rstd, beta.double() - mean * gamma.double() * rstd
    W = composed_weights(w3, wt, b3, bt, c)
    plo, phi = n * hlo * wlo, n * 4 * hlo * wlo
    dyv = G.to_nhwc(dy, "bf16")
    dhv = G.empty_nhwc(plo, 2 * c, "bf16", ld=2 * c + 64, off=0)
    ypv = G.to_nhwc(yprev, "bf16")
    dev = lambda t_: t_.float().cuda()      # noqa: E731
    sc, sh, mu, rs = dev(scale), dev(shift), dev(mean), dev(rstd)
    rows = _lib.load().unetdc_conv3x3_stats_rows(plo, 2 * c)
    parts = torch.full(((rows + 64) * 3 * 2 * c,), float("nan"), device="cuda")
    npart = ctypes.c_int(0)
    call("unetdc_upcomp_dgrad_bnstats", dyv.data_ptr(), dyv.stride(0), W["wc_d"].data_ptr(), dhv.data_ptr(), dhv.stride(0),
         ypv.data_ptr(), ypv.stride(0), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), parts.data_ptr(), parts.numel(),
         ctypes.byref(npart), n, hlo, wlo, c, G.DT["bf16"], G.stream())
    dh = G.from_nhwc(dhv, n, hlo, wlo)
    e = rel(dh, dh_ref)
    print(f"[upcomp dgrad {case}] rel-L2 {e:.2e}")
    assert e < 8e-3, e
    # skip half of the concat gradient: the ordinary dgrad on the sliced image, written into the second half of a [pixels, 2C] buffer
    dcat = G.empty_nhwc(phi, c, "bf16", ld=2 * c, off=c)
    G.conv3x3_dgrad(dyv, W["ws_d"], dcat, n, 2 * hlo, 2 * wlo, c, c, 1, "bf16")
    assert rel(G.from_nhwc(dcat, n, 2 * hlo, 2 * wlo), dskip_ref) < 6e-3
    # fused BatchNorm-backward sums, from the gradient as stored
    nrm = yd * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    gh = torch.where(nrm > 0, dh.double(), torch.zeros_like(nrm))
    xh = (yd - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    assert 1 <= npart.value <= rows
    pc = parts.cpu()[: npart.value * 3 * 2 * c].reshape(npart.value, 3, 2 * c).double().sum(0)
    s1, s2 = gh.sum(dim=(0, 2, 3)), (gh * xh).sum(dim=(0, 2, 3))
    assert float((pc[0] - s1).abs().max()) <= 2e-2 * float(gh.abs().sum(dim=(0, 2, 3)).max())
    assert float((pc[1] - s2).abs().max()) <= 2e-2 * float((gh * xh).abs().sum(dim=(0, 2, 3)).max())
    assert float(pc[2].abs().max()) == 0.0
    # run-to-run: bitwise
    dhv2 = G.empty_nhwc(plo, 2 * c, "bf16", ld=2 * c + 64, off=0)
    parts2 = torch.full_like(parts, float("nan"))
    call("unetdc_upcomp_dgrad_bnstats", dyv.data_ptr(), dyv.stride(0), W["wc_d"].data_ptr(), dhv2.data_ptr(), dhv2.stride(0),
         ypv.data_ptr(), ypv.stride(0), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), parts2.data_ptr(), parts2.numel(),
         ctypes.byref(npart), n, hlo, wlo, c, G.DT["bf16"], G.stream())
    assert torch.equal(dhv, dhv2) and torch.equal(parts[: npart.value * 6 * c], parts2[: npart.value * 6 * c])


@pytest.mark.parametrize("case", [(2, 16, 32, 64), (1, 8, 64, 128), (3, 24, 32, 64), (8, 32, 32, 128), (2, 64, 64, 256)])
def test_composed_weight_gradients(case):
    """unetdc_upcomp_wgrad: dW3 (both halves), dWT and dbT vs autograd of the explicit chain (fp32 CPU, bf16-rounded operands)."""
    n, hlo, wlo, c = case
    lib = _lib.load()
    g = gen(14)
    h = G.quant(torch.randn(n, 2 * c, hlo, wlo, generator=g), "bf16")
    skip = G.quant(torch.randn(n, c, 2 * hlo, 2 * wlo, generator=g), "bf16")
    w3 = G.quant(torch.randn(c, 2 * c, 3, 3, generator=g) / (3 * (2 * c) ** 0.5), "bf16").requires_grad_(True)
    wt = G.quant(torch.randn(2 * c, c, 2, 2, generator=g) / (2 * (2 * c) ** 0.5), "bf16").requires_grad_(True)
    b3 = torch.randn(c, generator=g)
    bt = torch.randn(c, generator=g).requires_grad_(True)
    dy = G.quant(torch.randn(n, c, 2 * hlo, 2 * wlo, generator=g), "bf16")
    y = F.conv2d(torch.cat([F.conv_transpose2d(h, wt, bt, stride=2), skip], 1), w3, b3, padding=1)
    dw3_ref, dwt_ref, dbt_ref = torch.autograd.grad(y, (w3, wt, bt), dy)
    W = composed_weights(w3.detach(), wt.detach(), b3, bt.detach(), c)
    hv = G.to_nhwc(h, "bf16", ld=2 * c + 64, off=0)
    sv = G.to_nhwc(skip, "bf16", ld=2 * c, off=c)
    dyv = G.to_nhwc(dy, "bf16")
    total = dy.double().sum(dim=(0, 2, 3)).float().cuda()
    nbytes = lib.unetdc_upcomp_wgrad_workspace(n, hlo, wlo, c, G.DT["bf16"])
    ws = G.workspace(nbytes)
    dw3 = torch.full((c, 2 * c, 3, 3), float("nan"), device="cuda")
    dwt = torch.full((2 * c, c, 2, 2), float("nan"), device="cuda")
    dbt = torch.full((c,), float("nan"), device="cuda")
    w3m, _, btd = W["keep"]
    args = (hv.data_ptr(), hv.stride(0), sv.data_ptr(), sv.stride(0), dyv.data_ptr(), dyv.stride(0), W["wtf"].data_ptr(),
            W["w3d"].data_ptr(), w3m.data_ptr(), btd.data_ptr(), total.data_ptr())
    call("unetdc_upcomp_wgrad", *args, dw3.data_ptr(), dwt.data_ptr(), dbt.data_ptr(), ws.data_ptr(), nbytes, n, hlo, wlo, c,
         G.DT["bf16"], G.stream())
    e_up, e_sk = rel(dw3[:, :c].cpu(), dw3_ref[:, :c]), rel(dw3[:, c:].cpu(), dw3_ref[:, c:])
    e_t, e_b = rel(dwt.cpu(), dwt_ref), rel(dbt.cpu(), dbt_ref)
    print(f"[upcomp wgrad {case}] rel-L2: dW3 up half {e_up:.2e}, skip half {e_sk:.2e}, dWT {e_t:.2e}, dbT {e_b:.2e}")
    # (dW' is rounded to bf16 before the decomposition GEMMs: one more 2^-9 than the explicit path's weight gradients)
    assert e_up < 1.2e-2 and e_sk < 8e-3 and e_t < 1.2e-2 and e_b < 2e-3, (e_up, e_sk, e_t, e_b)
    # run-to-run: bitwise
    dw3b, dwtb, dbtb = torch.full_like(dw3, float("nan")), torch.full_like(dwt, float("nan")), torch.full_like(dbt, float("nan"))
    call("unetdc_upcomp_wgrad", *args, dw3b.data_ptr(), dwtb.data_ptr(), dbtb.data_ptr(), ws.data_ptr(), nbytes, n, hlo, wlo, c,
         G.DT["bf16"], G.stream())
    assert torch.equal(dw3, dw3b) and torch.equal(dwt, dwtb) and torch.equal(dbt, dbtb)
