"""Host-side data preparation -- same names as the reference's ``utils/data_loader.py``.

Off the hot path (CPU image prep): provided so the kept entry points run
end to end.  cv2 / albumentations are optional; without them the same operations run on
numpy restatements of the OpenCV operators (ellipse element, opening, saturating subtract, min-max
normalise, 8-bit bilinear resize) written from OpenCV's definitions.
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


# ---------------------------------------------------------------------------------------------------------------------
# OpenCV's operators restated on numpy (cv2 is not installed in the build image).  These functions are the CPU path of
# the entry points AND the yardstick of the GPU preprocessing kernels (unet_dc_segmentation_amd/csrc/preprocess.hip,
# bit-exact against them).  Against cv2 itself they are "parity unpinned": written from OpenCV's documented definitions
# and source structure, not checked against a cv2 run.
def ellipse_spans(k):
    """Row spans [j1, j2) of cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)): r = c = k // 2,
    dx = round(c * sqrt(1 - dy^2 / r^2)), anchor (k // 2, k // 2)."""
    r = c = k // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    spans = []
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            spans.append((max(c - dx, 0), min(c + dx + 1, k)))
        else:
            spans.append((0, 0))
    return spans


def _window_reduce(a, width, op, ident):
    """out[..., x] = op over a[..., x : x + width] (a already padded on the right by >= width - 1 with `ident`)."""
    if width == 1:
        return a
    p = 1
    m = a
    while 2 * p <= width:                       # doubling: m[x] = op(a[x : x + p])
        m2 = np.full_like(m, ident)
        m2[..., : m.shape[-1] - p] = op(m[..., : m.shape[-1] - p], m[..., p:])
        m, p = m2, 2 * p
    if p == width:
        return m
    out = np.full_like(m, ident)
    sh = width - p
    out[..., : m.shape[-1] - sh] = op(m[..., : m.shape[-1] - sh], m[..., sh:])
    return out


def morph_cv2(plane, k, is_max):
    """cv2.erode / cv2.dilate of a 2-D uint8 array with the k x k ellipse: dst(y, x) = min|max over element pixels (i, j) of
    src(y + i - r, x + j - r); pixels outside the image do not take part (OpenCV's default border value)."""
    h, w = plane.shape
    r = k // 2
    ident = 0 if is_max else 255
    op = np.maximum if is_max else np.minimum
    pad = np.full((h + k, w + 2 * k), ident, dtype=np.uint8)
    pad[r:r + h, r:r + w] = plane               # pad[y + r, x + r] = src(y, x)
    out = np.full((h, w), ident, dtype=np.uint8)
    cache = {}
    for i, (j1, j2) in enumerate(ellipse_spans(k)):
        if j2 <= j1:
            continue
        if (j1, j2) not in cache:               # horizontal reduction over columns x + j - r, j in [j1, j2), for every row
            red = _window_reduce(pad, j2 - j1, op, ident)
            cache[(j1, j2)] = red[:, j1:j1 + w]     # column x + j1 - r of src = pad column x + j1
        out = op(out, cache[(j1, j2)][i:i + h])     # row y + i - r of src = pad row y + i
    return out


def rolling_ball_correction_rgb(image, radius=50):
    """Per channel: subtract the morphological opening (background), stretch to 0..255
    (/root/reference/utils/data_loader.py:11-24).  `radius` is the SIZE of the elliptical element, as in the reference."""
    if cv2 is not None:
        kernel = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (radius, radius))
        chans = []
        for ch in cv2.split(image):
            bg = cv2.morphologyEx(ch, cv2.MORPH_OPEN, kernel)
            chans.append(cv2.normalize(cv2.subtract(ch, bg), None, 0, 255, cv2.NORM_MINMAX))
        return cv2.merge(chans)
    out = np.empty_like(image)
    k = int(radius)
    for c in range(image.shape[2]):
        ch = np.ascontiguousarray(image[..., c])
        bg = morph_cv2(morph_cv2(ch, k, False), k, True)           # MORPH_OPEN = dilate(erode(.)) with the same element
        corr = np.clip(ch.astype(np.int16) - bg.astype(np.int16), 0, 255).astype(np.uint8)      # cv2.subtract saturates
        out[..., c] = normalize_minmax_u8(corr)
    return out


def normalize_minmax_u8(a):
    """cv2.normalize(a, None, 0, 255, cv2.NORM_MINMAX) on uint8: scale / shift formed in double, applied in float32
    (separate multiply and add), rounded half to even, saturated."""
    smin, smax = float(a.min()), float(a.max())
    scale = 255.0 * (1.0 / (smax - smin) if (smax - smin) > np.finfo(np.float64).eps else 0.0)
    shift = 0.0 - smin * scale
    v = a.astype(np.float32) * np.float32(scale) + np.float32(shift)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def linear_tables(src, dst):
    """Source index of the first tap and the two 11-bit coefficients of OpenCV's 8-bit INTER_LINEAR per destination index."""
    scale = src / dst
    ofs = np.empty(dst, dtype=np.int32)
    coef = np.empty((dst, 2), dtype=np.int16)
    for d in range(dst):
        f = (d + 0.5) * scale - 0.5
        s = int(np.floor(f))
        f -= s
        ofs[d] = s
        coef[d] = (int(np.rint(np.float32(1.0 - np.float32(f)) * 2048)), int(np.rint(np.float32(f) * 2048)))
    return ofs, coef


def resize_linear_cv2_u8(img, dw, dh):
    """cv2.resize(img, (dw, dh)) with the default INTER_LINEAR on uint8 HxWxC (or HxW): fixed-point coefficients (11 bits),
    horizontal pass in int32, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2."""
    a = img if img.ndim == 3 else img[..., None]
    h, w = a.shape[:2]
    xo, xa = linear_tables(w, dw)
    yo, ya = linear_tables(h, dh)
    neg, top = xo < 0, xo >= w - 1                 # left of the first / right of the last source pixel: one tap, weight 1
    xa = xa.copy()
    xa[neg | top] = (2048, 0)
    x0 = np.clip(xo, 0, w - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    y0, y1 = np.clip(yo, 0, h - 1), np.clip(yo + 1, 0, h - 1)
    s = a.astype(np.int32)
    rows = s[:, x0] * xa[:, 0].astype(np.int32)[None, :, None] + s[:, x1] * xa[:, 1].astype(np.int32)[None, :, None]
    r0, r1 = rows[y0], rows[y1]
    b0, b1 = ya[:, 0].astype(np.int32)[:, None, None], ya[:, 1].astype(np.int32)[:, None, None]
    v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def resize_image(arr, size, nearest=False):
    """Resize HxW(xC) uint8 array to (size, size): bilinear like the reference's calls end up doing (their third positional
    argument of cv2.resize is `dst`, so the interpolation stays at its INTER_LINEAR default), or nearest for masks."""
    if cv2 is not None:
        return cv2.resize(arr, (size, size), interpolation=cv2.INTER_NEAREST if nearest else cv2.INTER_LINEAR)
    if nearest:
        from unet_dc_segmentation_amd.droplets import resize_nearest_cv2
        return resize_nearest_cv2(arr, size, size)
    return resize_linear_cv2_u8(arr, size, size)


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
