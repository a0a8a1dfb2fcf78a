"""Host-side data preparation -- same names as the reference's ``utils/data_loader.py``.

Off the hot path (CPU image prep): provided so the kept entry points run
end to end.  cv2 / albumentations are optional; without them the same operations run on
PIL + SciPy (grey opening with an elliptical footprint == cv2.MORPH_OPEN with MORPH_ELLIPSE).
Reference lines: rolling_ball_correction_rgb :11-24, SegmentationDataset :26-76.
``transform`` may follow the albumentations protocol (keyword call, dict result -- what the reference passes)
or be a plain ``(img, mask) -> (img, mask)`` callable; :class:`TrainAugment` restates the reference's
augmentation list (train_DC_focal.py:183-190) on numpy / SciPy.
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
            img, mask = _apply_transform(self.transform, img, mask)
        img_t = _to_chw_tensor(img)
        mask_t = torch.as_tensor(np.ascontiguousarray(mask)).float()
        if mask_t.ndim == 2:
            mask_t = mask_t.unsqueeze(0)
        out = [img_t, mask_t]
        if self.return_orig_size:
            out.append((oh, ow))
        if self.return_filename:
            out.append(self.image_list[idx])
        return tuple(out)


def _apply_transform(transform, img, mask):
    """Both call conventions: the albumentations protocol the reference uses
    (``transform(image=img, mask=mask) -> {"image": ..., "mask": ...}``, utils/data_loader.py:59-62) and a plain
    ``(img, mask) -> (img, mask)`` callable."""
    try:
        out = transform(image=img, mask=mask)
    except TypeError:
        out = transform(img, mask)
    if isinstance(out, dict):
        return out["image"], out["mask"]
    return out


def _to_chw_tensor(img):
    """HWC numpy (or an already-CHW tensor, as albumentations' ToTensorV2 returns) -> CHW float tensor."""
    if torch.is_tensor(img):
        return img.float()
    return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float()


class TrainAugment:
    """The reference's training augmentation list (train_DC_focal.py:183-190) without albumentations:
    HorizontalFlip(0.5), VerticalFlip(0.2), RandomRotate90(0.5), RandomBrightnessContrast(0.2) (limits 0.2,
    float images: ``clip(alpha*img + beta*max, 0, 1)``), ElasticTransform(alpha=1, sigma=50, p=0.3).
    Callable both ways (``t(img, mask)`` and ``t(image=img, mask=mask)``).

    The random stream is seeded per DataLoader worker AND per epoch: each worker process derives its generator
    from ``torch.utils.data.get_worker_info().seed`` (base_seed + worker id, re-drawn by the DataLoader every
    epoch), so workers and epochs do not replay one sequence; in the main process (``num_workers=0``) one
    generator seeded with ``seed`` simply runs on across epochs."""

    def __init__(self, seed=0, brightness_contrast=True, elastic=True):
        self.seed, self.bc, self.elastic = int(seed), brightness_contrast, elastic
        self._rng, self._key = None, None

    def _generator(self):
        info = torch.utils.data.get_worker_info()
        key = ("worker", info.id, info.seed) if info is not None else ("main", os.getpid())
        if self._rng is None or self._key != key:
            entropy = [self.seed, info.seed & 0xFFFFFFFF, info.id] if info is not None else [self.seed]
            self._rng, self._key = np.random.default_rng(entropy), key
        return self._rng

    def __call__(self, img=None, mask=None, *, image=None):
        as_dict = image is not None
        if as_dict:
            img = image
        rng = self._generator()
        if rng.random() < 0.5:
            img, mask = img[:, ::-1], mask[:, ::-1]
        if rng.random() < 0.2:
            img, mask = img[::-1], mask[::-1]
        if rng.random() < 0.5:
            k = int(rng.integers(1, 4))
            img, mask = np.rot90(img, k, (0, 1)), np.rot90(mask, k, (0, 1))
        if self.bc and rng.random() < 0.2:
            alpha, beta = 1.0 + rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)
            img = np.clip(alpha * img + beta * float(img.max() if img.size else 1.0), 0.0, 1.0).astype(np.float32)
        if self.elastic and rng.random() < 0.3:
            img, mask = _elastic(img, mask, rng, alpha=1.0, sigma=50.0)
        if as_dict:
            return {"image": img, "mask": mask}
        return img, mask


def _elastic(img, mask, rng, alpha, sigma):
    """Elastic deformation: smooth random displacement field (Gaussian sigma) scaled by alpha; image bilinear,
    mask nearest (Simard et al. 2003 as albumentations implements it)."""
    from scipy import ndimage
    h, w = img.shape[:2]
    dx = ndimage.gaussian_filter(rng.random((h, w)) * 2 - 1, sigma, mode="constant") * alpha
    dy = ndimage.gaussian_filter(rng.random((h, w)) * 2 - 1, sigma, mode="constant") * alpha
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    coords = [yy + dy, xx + dx]
    img = np.ascontiguousarray(img)
    out = np.stack([ndimage.map_coordinates(img[..., c], coords, order=1, mode="reflect")
                    for c in range(img.shape[2])], axis=-1).astype(img.dtype)
    m = ndimage.map_coordinates(np.ascontiguousarray(mask), coords, order=0, mode="reflect").astype(mask.dtype)
    return out, m


def flip_rotate_augment(seed=0):
    """The geometric part only (flips + 90-degree rotations) of the reference's augmentation list."""
    return TrainAugment(seed, brightness_contrast=False, elastic=False)


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
