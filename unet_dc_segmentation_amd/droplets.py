"""Droplet quantification on the HIP device (csrc/ccl.hip): probabilities -> mask at the original image size -> 4-connected
components -> the per-droplet table of the reference's ``quantify()`` (/root/reference/quantify_droplets_batch.py:81-95).

Only the uint8 mask (needed for the PNG the script writes) and three integers per droplet cross PCIe; the reference
copies every fp32 probability map to the host, labels it twice with scikit-image and walks ``np.unique`` in Python.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def nearest_index(dst, src):
    """Source index of each destination index under cv2.resize(..., INTER_NEAREST): min(floor(d * src / dst), src - 1)."""
    return np.minimum(np.floor(np.arange(dst) * (src / dst)).astype(np.int64), src - 1)


def resize_nearest_cv2(mask, ow, oh):
    """numpy restatement of cv2.resize(mask, (ow, oh), interpolation=cv2.INTER_NEAREST) (used where cv2 is absent)."""
    return mask[nearest_index(oh, mask.shape[0])[:, None], nearest_index(ow, mask.shape[1])[None, :]]


# How the 512 x 512 mask gets to the original image size.  The reference writes
#     cv2.resize(mask512, (ow, oh), cv2.INTER_NEAREST)                    (/root/reference/quantify_droplets_batch.py:57)
# with the flag in the positional slot of `dst`: OpenCV ignores it and runs its default 8-bit INTER_LINEAR on the {0,1} mask.
# "reference" reproduces what that call computes, "nearest" what it names.  (cv2 is not installed here: both rules are
# restatements -- utils/data_loader.py:resize_linear_cv2_u8 and resize_nearest_cv2 below -- parity unpinned against cv2.)
MASK_RESIZE = "reference"


def resize_mask_like_reference(mask, ow, oh):
    """CPU path: uint8 {0,1} mask [h, w] -> [oh, ow] under MASK_RESIZE."""
    if MASK_RESIZE == "nearest" or (mask.shape[0] == oh and mask.shape[1] == ow):
        return resize_nearest_cv2(mask, ow, oh)
    from utils.data_loader import resize_linear_cv2_u8
    return resize_linear_cv2_u8(np.ascontiguousarray(mask, dtype=np.uint8), ow, oh)


def mask_and_droplets_batch(probs, thresh, out_hws, min_area, max_droplets=1 << 14):
    """probs: [B, H, W] fp32 probabilities on the HIP device; out_hws: B (oh, ow) pairs.  Every launch of the batch (mask,
    union-find, per-label sums, compaction) is enqueued back to back on the current stream into ONE set of output
    planes; the host then waits ONCE: one device->host copy brings the B droplet counts, a second the filled part of the
    per-droplet integers.  Returns a list of (mask uint8 [oh, ow] DEVICE tensor, area int64 [n], centroid_row float64
    [n], centroid_col float64 [n]) -- droplets in the reference's label order."""
    if not probs.is_cuda or probs.dtype != torch.float32 or probs.dim() != 3:
        raise _lib.UnetdcError("mask_and_droplets_batch needs a [B, H, W] fp32 tensor on the HIP device")
    probs = probs.contiguous()
    B, ph, pw = probs.shape
    dev = probs.device
    s = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()
    out_hws = [(int(h), int(w)) for h, w in out_hws]
    cap = int(min(max_droplets, max(h * w for h, w in out_hws)))
    wsb = max(lib.unetdc_ccl_workspace(h, w) for h, w in out_hws)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)           # one workspace: the launches are stream-ordered
    count = torch.zeros(B, dtype=torch.int32, device=dev)
    area = torch.empty(B, cap, dtype=torch.int32, device=dev)
    sums = torch.empty(2, B, cap, dtype=torch.int64, device=dev)
    masks = []
    for i, (oh, ow) in enumerate(out_hws):
        mask = torch.empty(oh, ow, dtype=torch.uint8, device=dev)
        p2 = probs[i]
        if MASK_RESIZE == "nearest" or (ph == oh and pw == ow):      # same size: both rules are the identity
            _lib.call("unetdc_mask_from_probs", p2.data_ptr(), ph, pw, float(thresh), mask.data_ptr(), oh, ow, s)
        else:
            from .preprocess import _resize_tables
            xo, xa = _resize_tables(pw, ow, dev, True)
            yo, ya = _resize_tables(ph, oh, dev, False)
            _lib.call("unetdc_mask_from_probs_linear", p2.data_ptr(), ph, pw, float(thresh), mask.data_ptr(), oh, ow,
                      xo.data_ptr(), xa.data_ptr(), yo.data_ptr(), ya.data_ptr(), s)
        _lib.call("unetdc_ccl_stats", mask.data_ptr(), oh, ow, int(max(min_area, 1)), ws.data_ptr(), wsb,
                  count[i:].data_ptr(), area[i].data_ptr(), sums[0, i].data_ptr(), sums[1, i].data_ptr(), None, cap, s)
        masks.append(mask)
    n = count.cpu().numpy().astype(np.int64)                        # the batch's only host wait
    nmax = int(min(n.max(initial=0), cap))
    a_h = area[:, :nmax].cpu().numpy().astype(np.int64)
    s_h = sums[:, :, :nmax].cpu().numpy()
    out = []
    for i, (oh, ow) in enumerate(out_hws):
        if n[i] > cap:                            # more droplets than the output capacity: this image again with room for all
            out.append(mask_and_droplets(probs[i], thresh, (oh, ow), min_area, max_droplets=int(n[i])))
            continue
        a = a_h[i, :n[i]]
        d = np.maximum(a, 1)
        out.append((masks[i], a, s_h[0, i, :n[i]].astype(np.float64) / d, s_h[1, i, :n[i]].astype(np.float64) / d))
    return out


def mask_and_droplets(probs2d, thresh, out_hw, min_area, max_droplets=1 << 16):
    """probs2d: [H, W] fp32 probabilities on the HIP device.  Returns (mask uint8 [oh, ow] DEVICE tensor,
    area int64 [n], centroid_row float64 [n], centroid_col float64 [n]) -- droplets in the reference's label order."""
    if not probs2d.is_cuda or probs2d.dtype != torch.float32:
        raise _lib.UnetdcError("mask_and_droplets needs an fp32 tensor on the HIP device")
    probs2d = probs2d.contiguous()
    ph, pw = probs2d.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = probs2d.device
    s = torch.cuda.current_stream().cuda_stream
    mask = torch.empty(oh, ow, dtype=torch.uint8, device=dev)
    if MASK_RESIZE == "nearest" or (ph == oh and pw == ow):          # same size: both rules are the identity
        _lib.call("unetdc_mask_from_probs", probs2d.data_ptr(), ph, pw, float(thresh), mask.data_ptr(), oh, ow, s)
    else:
        from .preprocess import _resize_tables
        xo, xa = _resize_tables(pw, ow, dev, True)
        yo, ya = _resize_tables(ph, oh, dev, False)
        _lib.call("unetdc_mask_from_probs_linear", probs2d.data_ptr(), ph, pw, float(thresh), mask.data_ptr(), oh, ow,
                  xo.data_ptr(), xa.data_ptr(), yo.data_ptr(), ya.data_ptr(), s)
    nbytes = _lib.load().unetdc_ccl_workspace(oh, ow)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cap = int(min(max_droplets, oh * ow))
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    area = torch.empty(cap, dtype=torch.int32, device=dev)
    sy = torch.empty(cap, dtype=torch.int64, device=dev)
    sx = torch.empty(cap, dtype=torch.int64, device=dev)
    _lib.call("unetdc_ccl_stats", mask.data_ptr(), oh, ow, int(max(min_area, 1)), ws.data_ptr(), nbytes, count.data_ptr(),
              area.data_ptr(), sy.data_ptr(), sx.data_ptr(), None, cap, s)
    n = int(count.item())
    if n > cap:                                   # more droplets than the output capacity: run again with room for all
        return mask_and_droplets(probs2d, thresh, out_hw, min_area, max_droplets=n)
    a = area[:n].cpu().numpy().astype(np.int64)
    cy = sy[:n].cpu().numpy().astype(np.float64) / np.maximum(a, 1)
    cx = sx[:n].cpu().numpy().astype(np.float64) / np.maximum(a, 1)
    return mask, a, cy, cx
