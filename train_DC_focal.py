#!/usr/bin/env python3
"""Training entry point -- drop-in for the reference's ``train_DC_focal.py`` on the MI355X path.

The reference is a module-level script with hard-coded constants; this keeps its loop
(train_DC_focal.py:241-358: Adam lr 1e-3, 15 epochs, batch 8, focal+dice (1.0, 2.0, 0.3),
0.3 threshold metrics, early stopping on validation Dice with patience 5, best ``state_dict`` saved
to ``best_UNetDC_focal_model.pth``) behind argparse flags whose defaults are those constants.
New: ``--synthetic`` (seeded droplet tiles, no dataset needed), ``--dtype`` (f32 | bf16 compute on
the HIP path), ``--steps`` (cap the steps per epoch), and data-parallel training when launched with
``python -m torch.distributed.run --nproc-per-node N train_DC_focal.py ...`` (RCCL all-reduce
overlapped with backward, unet_dc_segmentation_amd/dp.py).  The numbers of the reference's test section (:365-402, :452-467:
best checkpoint reloaded, test loss / Dice / pixel accuracy, precision / recall / F1 / specificity / confusion matrix) are
computed and printed; its PNG dumps and plots (:404-450, :468-611) are visualisation and out of scope.
"""
import argparse
import gc
import os
import time

import torch
from torch.utils.data import DataLoader, Subset

from unet_dc_segmentation_amd import dp as dpmod
from utils.data_loader import SegmentationDataset, SyntheticDropletDataset, TrainAugment
from utils.metrics_DC import combined_loss, dice_coef, focal_dice_loss, metrics_from_counts


def build_parser(arch="unetdc", epochs=15, ckpt="best_UNetDC_focal_model.pth", loss="focal_dice"):
    p = argparse.ArgumentParser("Train the U-Net(-DC) segmentation model")
    p.add_argument("--image_dir")
    p.add_argument("--mask_dir")
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic droplet tiles instead of a dataset")
    p.add_argument("--synthetic_len", type=int, default=64)
    p.add_argument("--arch", default=arch, choices=["unetdc", "unet"])
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--img_size", type=int, default=512)
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--epochs", type=int, default=epochs)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--patience", type=int, default=5)
    p.add_argument("--loss", default=loss, choices=["focal_dice", "bce_dice"])
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    p.add_argument("--steps", type=int, default=0, help="max training steps per epoch (0 = all)")
    p.add_argument("--optimizer", default="hip", choices=["hip", "torch"],
                   help="Adam implementation on a GPU: hip = fused step + weight re-pack kernel, torch = torch.optim.Adam")
    p.add_argument("--workers", type=int, default=4)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--ckpt_path", default=ckpt)
    p.add_argument("--no_test_eval", dest="test_eval", action="store_false",
                   help="skip the evaluation of the best checkpoint on the held-out test split after training")
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    return p


def make_datasets(args):
    if args.synthetic:
        full = SyntheticDropletDataset(args.synthetic_len, args.img_size, args.in_channels, seed=args.seed)
        n = len(full)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(args.seed)).tolist()
        n_test, n_val = max(1, n // 5), max(1, n // 5)                    # 60/20/20 like the reference
        return (Subset(full, idx[n_test + n_val:]), Subset(full, idx[n_test:n_test + n_val]),
                Subset(full, idx[:n_test]))
    if not args.image_dir or not args.mask_dir:
        raise SystemExit("--image_dir and --mask_dir are required unless --synthetic is given")
    exts = (".png", ".jpg", ".jpeg", ".tif")
    imgs = sorted(f for f in os.listdir(args.image_dir) if f.lower().endswith(exts))
    masks = sorted(f for f in os.listdir(args.mask_dir) if f.lower().endswith(exts))
    assert len(imgs) == len(masks), "Mismatch between the number of images and masks!"
    idx = torch.randperm(len(imgs), generator=torch.Generator().manual_seed(args.seed)).tolist()
    n_test = max(1, len(imgs) // 5)
    n_val = max(1, (len(imgs) - n_test) // 4)
    pick = lambda ids: ([imgs[i] for i in ids], [masks[i] for i in ids])   # noqa: E731
    tr, va, te = pick(idx[n_test + n_val:]), pick(idx[n_test:n_test + n_val]), pick(idx[:n_test])
    mk = lambda pair, tf: SegmentationDataset(args.image_dir, args.mask_dir, pair[0], pair[1], transform=tf,  # noqa: E731
                                              size=args.img_size)
    return mk(tr, TrainAugment(args.seed)), mk(va, None), mk(te, None)


class History(list):
    """Per-epoch records of main(); ``.test`` holds the held-out-split results of the final evaluation (None if skipped)."""
    test = None


def evaluate_test(model, loader, criterion, device):
    """The reference's final test pass (train_DC_focal.py:365-402, :452-467) without its image dumps: mean loss and Dice over
    the batches, pixel accuracy over all pixels, and calculate_metrics() on the thresholded predictions.  Sums stay on the
    device (one read-back), like the training loop's."""
    model.eval()
    loss_t = torch.zeros((), dtype=torch.float64, device=device)
    dice_t = torch.zeros((), dtype=torch.float64, device=device)
    cm_t = torch.zeros(4, dtype=torch.int64, device=device)           # tn, fp, fn, tp
    with torch.no_grad():
        for batch in loader:
            images, masks = batch[0].float().to(device), batch[1].float().to(device)
            outputs = model(images)
            loss_t += criterion(outputs, masks).double()
            pred = (outputs > 0.3).float()
            dice_t += dice_coef(masks, pred).double()
            yt, yp = masks > 0.5, pred > 0.5
            cm_t += torch.stack([(~yp & ~yt).sum(), (yp & ~yt).sum(), (~yp & yt).sum(), (yp & yt).sum()])
    nb = max(1, len(loader))
    tn, fp, fn, tp = (int(v) for v in cm_t.tolist())
    precision, recall, f1, specificity, cm = metrics_from_counts(tn, fp, fn, tp)     # = calculate_metrics() on the label arrays
    total = max(1, tn + fp + fn + tp)
    return dict(test_loss=float(loss_t.item()) / nb, test_dice=float(dice_t.item()) / nb, test_acc=(tn + tp) / total,
                precision=precision, recall=recall, f1=f1, specificity=specificity, confusion=cm.tolist())


def main(argv=None, parser=None):
    args = (parser or build_parser()).parse_args(argv)
    rank, local, world = dpmod.init_from_env()
    device = torch.device(args.device if args.device != "cuda" else f"cuda:{local}")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    if args.arch == "unetdc":
        from models.model_2 import UNetDC as Net
    else:
        from models.model import UNet as Net
    model = Net(in_channels=args.in_channels, out_channels=1).to(device)
    model.set_compute_dtype(args.dtype)
    wrapper = dpmod.DataParallel(model) if world > 1 else None
    if args.loss == "focal_dice":
        criterion = lambda pred, tgt: focal_dice_loss(pred, tgt, alpha=1.0, gamma=2.0, ratio=0.3)  # noqa: E731
    else:
        criterion = combined_loss
    # same update as the reference's torch.optim.Adam(lr) (train_DC_focal.py:224).  On the HIP path: one kernel that also
    # rewrites the packed weight images (unet_dc_segmentation_amd/optim.py); --optimizer torch keeps torch.optim.Adam
    if device.type == "cuda" and args.optimizer == "hip":
        from unet_dc_segmentation_amd.optim import FusedAdam
        optimizer = FusedAdam(model, lr=args.lr)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, fused=next(model.parameters()).is_cuda)

    train_ds, val_ds, test_ds = make_datasets(args)
    if world > 1:
        # each rank draws its own shard; shards are truncated to EQUAL length (as DistributedSampler with
        # drop_last does) so that every rank runs the same number of steps -- an extra step on one rank would
        # leave its gradient all-reduce without partners
        per_rank = len(train_ds) // world
        train_ds = Subset(train_ds, list(range(rank, per_rank * world, world)))
    pin = device.type == "cuda"
    # worker processes live across epochs (re-spawning them costs seconds per epoch: an interpreter + torch import each)
    keep = args.workers > 0
    # no drop_last, like the reference's loader (train_DC_focal.py:200): the ragged last batch trains too (the engines of both
    # batch shapes stay alive, unet.py::_engine_for; under data parallelism the shards are equal, so is every rank's last batch)
    train_loader = DataLoader(train_ds, batch_size=args.batch, shuffle=True, num_workers=args.workers,
                              pin_memory=pin, persistent_workers=keep)
    val_loader = DataLoader(val_ds, batch_size=args.batch, shuffle=False, num_workers=args.workers, pin_memory=pin,
                            persistent_workers=keep)
    if rank == 0:
        print(f"Training set: {len(train_ds)} images/rank, validation set: {len(val_ds)} images, "
              f"{world} rank(s), device {device}, compute {args.dtype}")

    best_dice, patience_counter = 0.0, 0
    saved_this_run = False          # the final test evaluation only reloads a checkpoint THIS run wrote
    history = History()
    for epoch in range(args.epochs):
        model.train()
        # the per-step metrics of train_DC_focal.py:256-262 are ACCUMULATED ON THE DEVICE (fp64 / int64: the same values added in
        # the same order as the reference's Python floats) and read back once per epoch: three .item() calls per step would
        # stall the launch queue three times per step
        tr_loss_t = torch.zeros((), dtype=torch.float64, device=device)
        tr_dice_t = torch.zeros((), dtype=torch.float64, device=device)
        correct_t = torch.zeros((), dtype=torch.int64, device=device)
        total = 0
        t0, seen = time.time(), 0
        for step, batch in enumerate(train_loader):
            if args.steps and step >= args.steps:
                break
            images, masks = batch[0].float().to(device, non_blocking=True), batch[1].float().to(device, non_blocking=True)
            optimizer.zero_grad()
            outputs = model(images)
            loss = criterion(outputs, masks)
            loss.backward()
            if wrapper is not None and not images.is_cuda:
                wrapper.sync_gradients()                    # ATen-CPU path: explicit all-reduce
            optimizer.step()
            with torch.no_grad():
                tr_loss_t += loss.detach().double()
                pred = (outputs > 0.3).float()
                tr_dice_t += dice_coef(masks, pred).double()
                correct_t += (pred == masks).sum()
            total += masks.numel()
            seen += images.shape[0]
            if epoch == 0 and step == 0:
                # Everything alive after the first step (torch, the model, the engine's buffers and descriptor tables, the
                # loader's workers) stays alive for the whole run: move it out of the cyclic collector's generations, so that
                # the full collections Python triggers during training scan the few objects of the loop instead of torch's
                # whole heap (a generation-2 pass there was measured at 40-60 ms: five training steps of host time).
                gc.collect()
                gc.freeze()
        nb = max(1, min(len(train_loader), args.steps or len(train_loader)))
        tr_loss, tr_dice, correct = float(tr_loss_t.item()), float(tr_dice_t.item()), int(correct_t.item())   # (waits for the epoch's work)
        dt = time.time() - t0
        # -------- validation --------
        if wrapper is not None:
            wrapper.broadcast_buffers()                     # rank 0's BatchNorm running statistics everywhere
        model.eval()
        va_loss_t = torch.zeros((), dtype=torch.float64, device=device)
        va_dice_t = torch.zeros((), dtype=torch.float64, device=device)
        vc_t = torch.zeros((), dtype=torch.int64, device=device)
        vt = 0
        with torch.no_grad():
            for batch in val_loader:
                images, masks = batch[0].float().to(device), batch[1].float().to(device)
                outputs = model(images)
                va_loss_t += criterion(outputs, masks).double()
                pred = (outputs > 0.3).float()
                va_dice_t += dice_coef(masks, pred).double()
                vc_t += (pred == masks).sum()
                vt += masks.numel()
        va_loss, va_dice, vc = float(va_loss_t.item()), float(va_dice_t.item()), int(vc_t.item())
        nv = max(1, len(val_loader))
        rec = dict(epoch=epoch + 1, train_loss=tr_loss / nb, val_loss=va_loss / nv, train_dice=tr_dice / nb,
                   val_dice=va_dice / nv, train_acc=correct / max(total, 1), val_acc=vc / max(vt, 1),
                   images_per_sec=seen * world / max(dt, 1e-9))
        if world > 1:
            # replicas must stay identical: one checksum per rank and epoch (compared by tests/test_dp_gloo.py)
            with torch.no_grad():
                rec["param_checksum"] = float(sum(p.double().sum() for p in model.parameters()))
            # early stopping / checkpointing are decided from rank 0's validation Dice on every rank, so all ranks
            # leave the loop in the same epoch (a rank that stopped alone would strand the others in all_reduce)
            rec["val_dice"] = dpmod.broadcast_scalar(rec["val_dice"], device)
        history.append(rec)
        if rank == 0:
            print(f"Epoch {epoch + 1}/{args.epochs} | Train Loss: {rec['train_loss']:.4f}, Val Loss: {rec['val_loss']:.4f}, "
                  f"Train Dice: {rec['train_dice']:.4f}, Val Dice: {rec['val_dice']:.4f}")
            print(f"Train Acc: {rec['train_acc']:.4f}, Val Acc: {rec['val_acc']:.4f} | {rec['images_per_sec']:.1f} img/s")
            print("-------------------------------------------------------")
        if rec["val_dice"] > best_dice:
            best_dice, patience_counter = rec["val_dice"], 0
            saved_this_run = True       # (decided from rank 0's broadcast Dice: the same on every rank)
            if rank == 0:
                torch.save(model.state_dict(), args.ckpt_path)
                print("Model saved!")
        else:
            patience_counter += 1
        if patience_counter >= args.patience:
            if rank == 0:
                print("Early stopping!")
            break
    # -------- final test evaluation (train_DC_focal.py:365-402, :452-467): best checkpoint, held-out split --------
    if world > 1:
        torch.distributed.barrier()                         # rank 0 has finished writing the checkpoint
    if args.test_eval and len(test_ds) > 0:
        if saved_this_run and os.path.exists(args.ckpt_path):
            model.load_state_dict(torch.load(args.ckpt_path, map_location=device, weights_only=True))
        elif rank == 0:
            # validation Dice never rose above 0: nothing was written by THIS run, and a file of that name left by an earlier
            # run (possibly another architecture) is not this run's result
            print(f"No checkpoint was written this run ({args.ckpt_path} not reloaded): evaluating the current weights")
        test_loader = DataLoader(test_ds, batch_size=args.batch, shuffle=False, num_workers=args.workers, pin_memory=pin)
        history.test = evaluate_test(model, test_loader, criterion, device)       # every rank: same weights, same split
        if rank == 0:
            t = history.test
            print("========== Test Results ==========")
            print(f"Test Loss: {t['test_loss']:.4f}")
            print(f"Test Dice: {t['test_dice']:.4f}")
            print(f"Test Accuracy (pixel-wise): {t['test_acc']:.4f}")
            print(f"Precision: {t['precision']:.4f}, Recall: {t['recall']:.4f}, F1: {t['f1']:.4f}, "
                  f"Specificity: {t['specificity']:.4f}")
            print(f"Confusion matrix [[tn, fp], [fn, tp]]: {t['confusion']}")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    gc.unfreeze()                   # (a caller that runs main() in-process gets its objects back under the collector)
    return history


if __name__ == "__main__":
    main()
