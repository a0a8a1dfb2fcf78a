"""Input preprocessing on the HIP device (csrc/preprocess.hip): rolling-ball background correction, bilinear resize to the
network size, /255, HWC -> CHW -- the steps of ``preprocess()`` (/root/reference/quantify_droplets_batch.py:40-46) and of
``SegmentationDataset.__getitem__`` (/root/reference/utils/data_loader.py:46-56) after the PNG has been decoded.

The yardstick is the numpy restatement of the OpenCV operators in ``utils/data_loader.py`` (bit-exact, tests/
test_gpu_preprocess.py); cv2 itself is not available to this build, so parity against cv2 is unpinned.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_tables = {}


def _resize_tables(src, dst, device, clamp):
    """Device copies of utils.data_loader.linear_tables (cached per (src, dst, device))."""
    key = (src, dst, str(device), clamp)
    if key not in _tables:
        from utils.data_loader import linear_tables
        ofs, coef = linear_tables(src, dst)
        if clamp:                                  # x direction: outside the source -> one tap with weight 1
            coef = coef.copy()
            coef[(ofs < 0) | (ofs >= src - 1)] = (2048, 0)
            ofs = np.clip(ofs, 0, src - 1)
        _tables[key] = (torch.from_numpy(ofs.astype(np.int32)).to(device),
                        torch.from_numpy(np.ascontiguousarray(coef.astype(np.int16))).to(device))
    return _tables[key]


def rolling_ball_device(img_u8, radius=50):
    """img_u8: [H, W, C] uint8 tensor on the HIP device -> corrected [H, W, C] uint8 tensor (same device)."""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or img_u8.dim() != 3:
        raise _lib.UnetdcError("rolling_ball_device needs an [H, W, C] uint8 tensor on the HIP device")
    img_u8 = img_u8.contiguous()
    h, w, c = img_u8.shape
    out = torch.empty_like(img_u8)
    nbytes = _lib.load().unetdc_rolling_ball_workspace(h, w, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=img_u8.device)
    _lib.call("unetdc_rolling_ball_u8", img_u8.data_ptr(), out.data_ptr(), h, w, c, int(radius), ws.data_ptr(), nbytes,
              torch.cuda.current_stream().cuda_stream)
    return out


def resize_to_input_device(img_u8, size):
    """[H, W, C] uint8 (device) -> [C, size, size] float32 in [0, 1] (device): bilinear (OpenCV 8-bit rule), /255, CHW."""
    img_u8 = img_u8.contiguous()
    h, w, c = img_u8.shape
    dev = img_u8.device
    xo, xa = _resize_tables(w, size, dev, True)
    yo, ya = _resize_tables(h, size, dev, False)
    out = torch.empty(c, size, size, dtype=torch.float32, device=dev)
    _lib.call("unetdc_resize_linear_u8_to_chw_f32", img_u8.data_ptr(), h, w, c, out.data_ptr(), size, size, xo.data_ptr(),
              xa.data_ptr(), yo.data_ptr(), ya.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out


MAX_ELEMENT = 128          # csrc/preprocess.hip PP_MAXK: the structuring element and its halo must fit one LDS tile

def _upload(img_u8_host, device):
    """Decoded image (numpy uint8) -> device tensor.  A plain pageable-memory upload: staging through a ring of pinned buffers
    was tried in round 3 and made the quantification flow 9x SLOWER (90 vs 800 images/s) -- the CPU copy of 4.3 MB into HIP
    pinned (uncached on the host side) memory takes ~10 ms, far more than the runtime's own staged copy."""
    return torch.from_numpy(np.ascontiguousarray(img_u8_host)).to(device)


def preprocess_device(img_u8_host, radius, size, device="cuda"):
    """numpy [H, W, 3] uint8 (a decoded image) -> network input [3, size, size] float32 on the device.
    A structuring element larger than the kernel's LDS tile allows (--background_radius > 128) takes the host operator for
    the rolling ball only (utils.data_loader.rolling_ball_correction_rgb, the same arithmetic); resize, /255 and CHW stay
    on the device."""
    if int(radius) > MAX_ELEMENT:
        from utils.data_loader import rolling_ball_correction_rgb
        return resize_to_input_device(_upload(rolling_ball_correction_rgb(img_u8_host, int(radius)), device), size)
    return resize_to_input_device(rolling_ball_device(_upload(img_u8_host, device), radius), size)
