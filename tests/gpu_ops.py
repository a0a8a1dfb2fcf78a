"""Thin test-side wrappers over the C ABI (raw pointers through ctypes) + layout helpers."""
import ctypes

import torch

from unet_dc_segmentation_amd import _lib
from unet_dc_segmentation_amd._lib import call

TD = {"f32": torch.float32, "bf16": torch.bfloat16}
DT = {"f32": _lib.F32, "bf16": _lib.BF16}


LIVE_ROWS = ctypes.c_int(-1)      # stats_rows out-parameter of the last conv3x3_fwd() below (rows that carry data)


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_nhwc(x_nchw, dtype, ld=None, off=0, fill=0.0):
    """NCHW cpu/gpu float tensor -> [N*H*W, C] view (stride ld) of a fresh device buffer."""
    n, c, h, w = x_nchw.shape
    buf = torch.full((n * h * w, ld or c), fill, dtype=TD[dtype], device="cuda")
    view = buf[:, off:off + c]
    view.copy_(x_nchw.permute(0, 2, 3, 1).reshape(-1, c).to("cuda"))
    return view


def from_nhwc(view, n, h, w):
    return view.float().reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous().cpu()


def quant(x, dtype):
    """Round a float tensor through the storage type (what the kernels read)."""
    return x.to(TD[dtype]).float()


def empty_nhwc(npix, c, dtype, ld=None, off=0, fill=float("nan")):
    buf = torch.full((npix, ld or c), fill, dtype=TD[dtype], device="cuda")
    return buf[:, off:off + c]


def pack_conv(w, dtype, dgrad=True):
    co, ci = w.shape[:2]
    wd = w.to("cuda").contiguous()
    wf = torch.empty(9 * co * ci, dtype=TD[dtype], device="cuda")
    wg = torch.empty(9 * co * ci, dtype=TD[dtype], device="cuda") if dgrad else None
    call("unetdc_pack_conv3x3", wd.data_ptr(), wf.data_ptr(), wg.data_ptr() if dgrad else None, co, ci, DT[dtype],
         stream())
    return wf, wg


def pack_convT(w, dtype):
    ci, co = w.shape[:2]
    wd = w.to("cuda").contiguous()
    wf = torch.empty(4 * co * ci, dtype=TD[dtype], device="cuda")
    wg = torch.empty(4 * co * ci, dtype=TD[dtype], device="cuda")
    call("unetdc_pack_convT2x2", wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), ci, co, DT[dtype], stream())
    return wf, wg


def conv3x3_fwd(xv, wf, bias, n, h, w, cin, cout, d, dtype, yv, stats=False, scale=None, shift=None):
    rows = _lib.load().unetdc_conv3x3_stats_rows(n * h * w, cout)
    st = torch.full(((rows + 64) * 2 * cout,), float("nan"), device="cuda") if stats else None
    LIVE_ROWS.value = -1
    call("unetdc_conv3x3_fwd", xv.data_ptr(), xv.stride(0), wf.data_ptr(), None if bias is None else bias.data_ptr(),
         None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), yv.data_ptr(),
         yv.stride(0), None if st is None else st.data_ptr(), ctypes.byref(LIVE_ROWS), n, h, w, cin, cout, d, DT[dtype],
         stream())
    return st, rows


def conv3x3_dgrad(dyv, wd, dxv, n, h, w, cin, cout, d, dtype):
    call("unetdc_conv3x3_dgrad", dyv.data_ptr(), dyv.stride(0), wd.data_ptr(), dxv.data_ptr(), dxv.stride(0), n, h, w,
         cin, cout, d, DT[dtype], stream())


def conv3x3_wgrad(xv, dyv, n, h, w, cin, cout, d, dtype):
    nbytes = _lib.load().unetdc_conv3x3_wgrad_workspace(n, h, w, cin, cout, DT[dtype])
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dw = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    call("unetdc_conv3x3_wgrad", xv.data_ptr(), xv.stride(0), dyv.data_ptr(), dyv.stride(0), dw.data_ptr(),
         ws.data_ptr(), nbytes, n, h, w, cin, cout, d, DT[dtype], stream())
    return dw


def workspace(nbytes):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device="cuda")
