#!/usr/bin/env python3
"""Batch inference + droplet quantification -- drop-in for the reference's
``quantify_droplets_batch.py`` (same CLI flags: the Tk/Qt front-ends build this argv,
gui.py:26-39, gui_qt.py:372-400; same output files and CSV columns, outputs/all_droplets.csv:1).

Flow (reference lines): load_model :34-37 -> preprocess :40-46 (RGB, rolling ball, resize 512, /255)
-> run_batch :48-79 (model(batch) under no_grad, ``> thresh`` on the probabilities, nearest resize
to the original size, mask PNG, per-image CSV) -> quantify :81-95 (4-connected components, area
filter, area / equivalent diameter / centroid) -> summary CSV / XLSX / histogram :163-199.
The network runs on the HIP kernels when a GPU is present (``DEVICE = "cuda"``), otherwise on
PyTorch-CPU exactly like the reference.  On the GPU the threshold, the nearest-neighbour resize to the original
size, the connected-component labelling and the per-droplet sums run on the device as well
(unet_dc_segmentation_amd/droplets.py, csrc/ccl.hip): only the uint8 mask and three integers per droplet are
copied back.  cv2 / scikit-image are optional on the CPU path: SciPy's ``ndimage.label`` with its default
cross-shaped structure is skimage's ``label(connectivity=1)``; the resize of the mask to the original size reproduces what the
reference's call computes (its interpolation flag sits in the positional slot of ``dst``, so OpenCV's default 8-bit bilinear
runs on the {0,1} mask: unet_dc_segmentation_amd/droplets.py:MASK_RESIZE).
"""
import argparse
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from PIL import Image

from models.model_2 import UNetDC
from unet_dc_segmentation_amd.droplets import resize_mask_like_reference
from utils.data_loader import resize_image, rolling_ball_correction_rgb

DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
IMG_SIZE = 512          # matches the training resize


def load_model(ckpt, dtype="f32"):
    m = UNetDC(in_channels=3, out_channels=1)
    m.load_state_dict(torch.load(ckpt, map_location=DEVICE, weights_only=True))
    m.set_compute_dtype(dtype)
    return m.to(DEVICE).eval()


def decode_rgb(path):
    """File -> uint8 RGB array (runs in the decode pool: PIL releases the GIL while it inflates)."""
    return np.array(Image.open(path).convert("RGB"))


def preprocess(path, background_radius, im=None):
    im = decode_rgb(path) if im is None else im
    oh, ow = im.shape[:2]
    if DEVICE == "cuda":                                 # rolling ball + resize + /255 + CHW on the GPU (csrc/preprocess.hip)
        from unet_dc_segmentation_amd.preprocess import preprocess_device
        return preprocess_device(im, background_radius, IMG_SIZE, DEVICE), (oh, ow)
    im = rolling_ball_correction_rgb(im, background_radius)
    im = resize_image(im, IMG_SIZE).astype(np.float32) / 255.0
    return torch.from_numpy(im).permute(2, 0, 1), (oh, ow)


def _droplet_table(area, cen_row, cen_col, px_per_um):
    """DataFrame with the reference's columns (outputs/all_droplets.csv:1) from per-droplet areas and centroids given in
    label order; equivalent_diameter = sqrt(4 area / pi) (known answer: area 18224 -> 152.327, outputs/all_droplets.csv:2)."""
    n = len(area)
    if n == 0:
        return pd.DataFrame()
    area = np.asarray(area, dtype=np.int64)
    df = pd.DataFrame({"label": np.arange(1, n + 1), "area": area, "equivalent_diameter": np.sqrt(4.0 * area / np.pi),
                       "centroid-0": np.asarray(cen_row, dtype=np.float64), "centroid-1": np.asarray(cen_col, dtype=np.float64)})
    if px_per_um is not None:
        df["area_sqmicron"] = df["area"] / (px_per_um ** 2)
        df["eq_diam_micron"] = df["equivalent_diameter"] / px_per_um
    return df


def quantify(bin_mask, min_area, px_per_um):
    """Per-droplet table: label, area, equivalent_diameter, centroid-0/1 (+ micron columns) -- CPU path (SciPy)."""
    from scipy import ndimage
    lbl, n = ndimage.label(bin_mask)                     # 4-connectivity, labels in raster order of the first pixel
    if n:
        areas = np.bincount(lbl.ravel(), minlength=n + 1)
        keep = areas >= min_area
        keep[0] = False
        lbl, n = ndimage.label(keep[lbl])
    if n == 0:
        return pd.DataFrame()
    idx = np.arange(1, n + 1)
    area = np.bincount(lbl.ravel(), minlength=n + 1)[1:]
    cen = np.array(ndimage.center_of_mass(np.ones_like(lbl), lbl, idx)).reshape(n, 2)
    return _droplet_table(area, cen[:, 0], cen[:, 1], px_per_um)


def quantify_device(probs2d, thresh, out_hw, min_area, px_per_um):
    """The same table from the probability map still on the HIP device; also returns the uint8 mask (host) for the PNG."""
    from unet_dc_segmentation_amd.droplets import mask_and_droplets
    mask, area, cy, cx = mask_and_droplets(probs2d, thresh, out_hw, min_area)
    return mask.cpu().numpy(), _droplet_table(area, cy, cx, px_per_um)


def _outline(mask):
    from scipy import ndimage
    return mask.astype(bool) & ~ndimage.binary_erosion(mask.astype(bool), iterations=2)


def _write_outputs(mask, df, fpath, name, mask_dir, overlay_dir):
    """The per-image files of run_batch (reference :66-79): mask PNG, droplet CSV, optional overlay."""
    Image.fromarray(mask * 255).save(str(mask_dir / f"{name}_pred.png"))
    df.to_csv(mask_dir.parent / f"{name}_droplets.csv", index=False)
    if overlay_dir is not None:
        img = np.array(Image.open(fpath).convert("RGB"))
        img[_outline(mask)] = (0, 255, 0)
        Image.fromarray(img).save(str(overlay_dir / f"{name}_overlay.png"))


@torch.no_grad()
def run_batch(tensors, meta, model, mask_dir, overlay_dir, thresh, min_area, px_per_um, per_image_rows, all_props, writers=None):
    batch = torch.stack(tensors).to(DEVICE)
    probs = model(batch)                                 # sigmoid probabilities (model_2.py:80)
    on_device = probs.is_cuda
    masks512 = None if on_device else (probs[:, 0] > thresh).to(torch.uint8).numpy()
    if on_device:                                        # the whole batch enqueued back to back, ONE host wait (droplets.py)
        from unet_dc_segmentation_amd.droplets import mask_and_droplets_batch
        dev_out = mask_and_droplets_batch(probs[:, 0], thresh, [m[1] for m in meta], min_area)
    for i in range(len(tensors)):
        fpath, (oh, ow) = meta[i]
        name = Path(fpath).stem
        if on_device:
            mask_d, area, cy, cx = dev_out[i]
            mask, df = mask_d.cpu().numpy(), _droplet_table(area, cy, cx, px_per_um)
        else:
            mask = resize_mask_like_reference(masks512[i], ow, oh)
            df = quantify(mask, min_area, px_per_um)
        df.insert(0, "filename", Path(fpath).name) if not df.empty else None
        all_props.append(df)
        per_image_rows.append({"filename": Path(fpath).name, "droplet_count": len(df),
                               "total_area_px": df["area"].sum() if not df.empty else 0})
        # PNG deflate and CSV formatting are the slowest part of an image once the network runs on the device: they go to
        # the writer pool (same files, same contents; main() waits for them before the summary is written)
        if writers is None:
            _write_outputs(mask, df, fpath, name, mask_dir, overlay_dir)
        else:
            writers[1].append(writers[0].submit(_write_outputs, mask, df, fpath, name, mask_dir, overlay_dir))
            # back-pressure: at most ~4 batches of masks / tables wait for the writers; waiting on the OLDEST write also surfaces a
            # failed write while the run is still going
            while len(writers[1]) > writers[2]:
                writers[1].popleft().result()


def build_parser():
    p = argparse.ArgumentParser("Segment lipid droplets and build a report")
    p.add_argument("--img_dir", required=True)
    p.add_argument("--ckpt_path", default="best_UNetDC_focal_model.pth")
    p.add_argument("--out_dir", default="quant_results")
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--prob_thresh", type=float, default=0.3)
    p.add_argument("--min_area", type=int, default=1, help="ignore objects smaller than this (pixels^2)")
    p.add_argument("--px_per_micron", type=float, help="pixels per micron for physical-unit columns")
    p.add_argument("--save_overlays", action="store_true")
    p.add_argument("--background_radius", type=int, default=50, help="radius for rolling ball background correction")
    p.add_argument("--skip_excel", action="store_true", help="skip generation of the Excel workbook")
    p.add_argument("--skip_histogram", action="store_true", help="skip histogram plot generation")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="compute type of the HIP path")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    in_dir, out_dir = Path(args.img_dir), Path(args.out_dir)
    mask_dir = out_dir / "predicted_masks"
    overlay_dir = out_dir / "overlays" if args.save_overlays else None
    out_dir.mkdir(parents=True, exist_ok=True)
    mask_dir.mkdir(exist_ok=True)
    if overlay_dir:
        overlay_dir.mkdir(exist_ok=True)
    model = load_model(args.ckpt_path, args.dtype)
    tensors, meta, per_image_rows, all_props = [], [], [], []
    images = sorted(p for p in in_dir.iterdir() if p.suffix.lower() in {".png", ".jpg", ".jpeg", ".tif", ".tiff"})
    # File decode runs ahead of the device and the per-image writes behind it, in two small thread pools (both are
    # zlib-bound and release the GIL); the order of the rows in the summary files is the order of `images` as before.
    import os
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(8, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(nthreads) as dec_pool, ThreadPoolExecutor(nthreads) as wr_pool:
        writers = (wr_pool, deque(), 4 * max(1, args.batch))
        ahead = deque()
        it = iter(images)

        def refill():
            while len(ahead) < 2 * args.batch:
                nxt = next(it, None)
                if nxt is None:
                    return
                ahead.append((nxt, dec_pool.submit(decode_rgb, nxt)))

        refill()
        while ahead:
            img, fut = ahead.popleft()
            refill()
            t, osize = preprocess(img, args.background_radius, fut.result())
            tensors.append(t)
            meta.append((str(img), osize))
            if len(tensors) == args.batch:
                run_batch(tensors, meta, model, mask_dir, overlay_dir, args.prob_thresh, args.min_area,
                          args.px_per_micron, per_image_rows, all_props, writers)
                tensors, meta = [], []
        if tensors:
            run_batch(tensors, meta, model, mask_dir, overlay_dir, args.prob_thresh, args.min_area,
                      args.px_per_micron, per_image_rows, all_props, writers)
        for f in writers[1]:
            f.result()                                   # re-raises a failed write
    summary_df = pd.DataFrame(per_image_rows)
    summary_df.to_csv(out_dir / "summary_per_image.csv", index=False)
    props = [d for d in all_props if not d.empty]
    if props:
        combined = pd.concat(props, ignore_index=True)
        combined.to_csv(out_dir / "all_droplets.csv", index=False)
        if not args.skip_excel:
            try:
                import xlsxwriter  # noqa: F401
                with pd.ExcelWriter(out_dir / "all_droplets.xlsx", engine="xlsxwriter") as xw:
                    combined.to_excel(xw, index=False, sheet_name="droplets")
                    summary_df.to_excel(xw, index=False, sheet_name="per_image")
            except (ImportError, AttributeError):
                combined.to_csv(out_dir / "all_droplets_noexcel.csv", index=False)
                print("Skipped Excel file (xlsxwriter not available); wrote all_droplets_noexcel.csv")
        size_col = "eq_diam_micron" if "eq_diam_micron" in combined.columns else "equivalent_diameter"
        stats = combined[size_col].describe()[["mean", "50%", "std"]].rename({"50%": "median"})
        stats.to_csv(out_dir / "droplet_size_stats.csv")
        if not args.skip_histogram:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.figure(figsize=(6, 4))
            plt.hist(combined[size_col], bins=40)
            plt.xlabel("Diameter (um)" if "micron" in size_col else "Diameter (pixels)")
            plt.ylabel("Count")
            plt.title("Droplet size distribution")
            plt.tight_layout()
            plt.savefig(out_dir / "size_histogram.png", dpi=300)
            plt.close()
    print("\n All done. Outputs are in ", out_dir)
    return out_dir


if __name__ == "__main__":
    main()
