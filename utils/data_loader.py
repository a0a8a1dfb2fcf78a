"""Host-side data preparation -- same names as the reference's ``utils/data_loader.py``.

Off the hot path (CPU image prep): provided so the kept entry points run
end to end.  cv2 / albumentations are optional; without them the same operations run on
PIL + SciPy (grey opening with an elliptical footprint == cv2.MORPH_OPEN with MORPH_ELLIPSE).
Reference lines: rolling_ball_correction_rgb :11-24, SegmentationDataset :26-76.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

try:                                    # optional accelerators for the CPU prep
    import cv2
except Exception:                       # pragma: no cover - cv2 absent in the build image
    cv2 = None


def _ellipse(radius):
    """Elliptical structuring element of size (radius, radius), like cv2.getStructuringElement."""
    r = max(int(radius), 1)
    yy, xx = np.mgrid[0:r, 0:r]
    c = (r - 1) / 2.0
    return (((yy - c) / max(c, 0.5)) ** 2 + ((xx - c) / max(c, 0.5)) ** 2) <= 1.0 + 1e-9


def rolling_ball_correction_rgb(image, radius=50):
    """Per channel: subtract the morphological opening (background), stretch to 0..255."""
    if cv2 is not None:
        kernel = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (radius, radius))
        chans = []
        for ch in cv2.split(image):
            bg = cv2.morphologyEx(ch, cv2.MORPH_OPEN, kernel)
            chans.append(cv2.normalize(cv2.subtract(ch, bg), None, 0, 255, cv2.NORM_MINMAX))
        return cv2.merge(chans)
    from scipy import ndimage
    fp = _ellipse(radius)
    out = np.empty_like(image)
    for c in range(image.shape[2]):
        ch = image[..., c]
        bg = ndimage.grey_opening(ch, footprint=fp, mode="nearest")
        corr = np.clip(ch.astype(np.int32) - bg.astype(np.int32), 0, 255).astype(np.float32)
        lo, hi = float(corr.min()), float(corr.max())
        out[..., c] = np.round((corr - lo) * (255.0 / (hi - lo))).astype(image.dtype) if hi > lo else 0
    return out


def resize_image(arr, size, nearest=False):
    """Resize HxW(xC) uint8/float array to (size, size)."""
    if cv2 is not None:
        return cv2.resize(arr, (size, size), interpolation=cv2.INTER_NEAREST if nearest else cv2.INTER_AREA)
    mode = Image.NEAREST if nearest else Image.BILINEAR
    if arr.ndim == 2:
        return np.array(Image.fromarray(arr).resize((size, size), mode))
    return np.stack([np.array(Image.fromarray(arr[..., c]).resize((size, size), mode))
                     for c in range(arr.shape[2])], axis=-1)


class SegmentationDataset(Dataset):
    """(image, mask[, (orig_h, orig_w)][, filename]) with the reference's preprocessing:
    RGB -> rolling ball (r=50) -> resize 512 -> /255; mask binarised and resized (nearest)."""

    def __init__(self, image_dir, mask_dir, image_list, mask_list, transform=None, return_filename=True,
                 return_orig_size=True, size=512, radius=50):
        self.image_dir, self.mask_dir = image_dir, mask_dir
        self.image_list, self.mask_list = list(image_list), list(mask_list)
        self.transform = transform
        self.return_filename, self.return_orig_size = return_filename, return_orig_size
        self.size, self.radius = size, radius

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, idx):
        img = np.array(Image.open(os.path.join(self.image_dir, self.image_list[idx])).convert("RGB"))
        oh, ow = img.shape[:2]
        img = rolling_ball_correction_rgb(img, radius=self.radius)
        mask = np.array(Image.open(os.path.join(self.mask_dir, self.mask_list[idx])).convert("L"))
        mask = (mask > 0).astype(np.uint8)
        img = resize_image(img, self.size).astype(np.float32) / 255.0
        mask = resize_image(mask, self.size, nearest=True)
        if self.transform is not None:
            img, mask = self.transform(img, mask)
        img_t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float()
        mask_t = torch.from_numpy(np.ascontiguousarray(mask)).float().unsqueeze(0)
        out = [img_t, mask_t]
        if self.return_orig_size:
            out.append((oh, ow))
        if self.return_filename:
            out.append(self.image_list[idx])
        return tuple(out)


def flip_rotate_augment(seed=0):
    """Horizontal/vertical flip + 90-degree rotation (the geometric part of the reference's
    augmentation list, train_DC_focal.py:183-190), as a plain (img, mask) -> (img, mask) callable."""
    rng = np.random.default_rng(seed)

    def f(img, mask):
        if rng.random() < 0.5:
            img, mask = img[:, ::-1], mask[:, ::-1]
        if rng.random() < 0.2:
            img, mask = img[::-1], mask[::-1]
        if rng.random() < 0.5:
            k = int(rng.integers(1, 4))
            img, mask = np.rot90(img, k, (0, 1)), np.rot90(mask, k, (0, 1))
        return img, mask
    return f


class SyntheticDropletDataset(Dataset):
    """Seeded stand-in for microscopy tiles: noise background + bright discs, mask = discs.
    Same tuple layout as SegmentationDataset (there is no network access for real data)."""

    def __init__(self, length=64, size=512, channels=3, discs=200, seed=0):
        self.length, self.size, self.channels, self.discs, self.seed = length, size, channels, discs, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        rng = np.random.default_rng(self.seed * 100003 + idx)
        s = self.size
        img = rng.random((self.channels, s, s), dtype=np.float32) * 0.6
        mask = np.zeros((1, s, s), dtype=np.float32)
        yy, xx = np.mgrid[0:s, 0:s]
        n = max(1, self.discs * s * s // (512 * 512))
        for _ in range(n):
            cy, cx, r = rng.integers(0, s), rng.integers(0, s), rng.uniform(1, 12)
            y0, y1, x0, x1 = max(cy - 13, 0), min(cy + 14, s), max(cx - 13, 0), min(cx + 14, s)
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
            mask[0, y0:y1, x0:x1][m] = 1.0
        img += 0.4 * mask
        return torch.from_numpy(img), torch.from_numpy(mask), (s, s), f"synthetic_{idx:05d}.png"
