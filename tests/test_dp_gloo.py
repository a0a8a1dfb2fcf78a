"""Data-parallel path on CPU: world_size-2 gloo processes exercise the bucketed gradient exchange
(unet_dc_segmentation_amd/dp.py) exactly as the HIP backward drives it (ready-range hooks in reverse
parameter order + finish), replica broadcast, and an end-to-end averaged training step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from models.model_2 import UNetDC
        from oracle import recipe
        from unet_dc_segmentation_amd.dp import DataParallel
        from utils.metrics_DC import focal_dice_loss
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        model = UNetDC(1, 1)
        dp = DataParallel(model, bucket_bytes=8 << 20)
        # 1. replicas identical after the rank-0 broadcast
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()])
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # 2. the hook protocol of the HIP backward: one flat buffer, blocks ready in reverse order
        params = list(model.parameters())
        offs, o = {}, 0
        for p in params:
            offs[id(p)] = o
            o += p.numel()
        flat = torch.full((o,), float(rank + 1))
        flat[:1000] += torch.arange(1000.0) * (rank + 1)
        order = ["out_conv", "dec1", "upconv1", "dec2", "upconv2", "dec3", "upconv3", "dec4", "upconv4",
                 "bottleneck", "enc4", "enc3", "enc2", "enc1"]
        for name in order:
            ps = list(getattr(model, name).parameters())
            dp._on_ready(flat, offs[id(ps[0])], offs[id(ps[-1])] + ps[-1].numel())
        dp.finish()
        mean_scale = sum(range(1, world + 1)) / world
        expect = torch.full((o,), mean_scale)
        expect[:1000] += torch.arange(1000.0) * mean_scale
        assert torch.allclose(flat, expect, rtol=0, atol=1e-5)
        assert 2 <= dp.stats["buckets"] <= 14 and dp.stats["elems"] == o       # every element exactly once
        # 3. end-to-end step on the ATen-CPU path: averaged gradients == mean of the per-rank gradients
        xs = [recipe.seeded_input(50 + r, (1, 1, 32, 32)) for r in range(world)]
        ts = [recipe.seeded_target(60 + r, (1, 1, 32, 32)) for r in range(world)]
        model.train()
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        per_rank = []
        for r in range(world):
            model.load_state_dict(sd0)
            model.zero_grad()
            focal_dice_loss(model(xs[r]), ts[r]).backward()
            per_rank.append([p.grad.clone() for p in params])
        model.load_state_dict(sd0)
        model.zero_grad()
        focal_dice_loss(model(xs[rank]), ts[rank]).backward()
        dp.sync_gradients()
        for i, p in enumerate(params):
            mean = sum(g[i] for g in per_rank) / world
            assert torch.allclose(p.grad, mean, rtol=1e-5, atol=1e-7), i
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(e) + "\n" + traceback.format_exc()))


@pytest.mark.timeout(600)
def test_data_parallel_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in results), results


def _train_worker(rank, world, port, q, args):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        torch.set_num_threads(2)
        import train_DC_focal as t
        hist = t.main(args)
        q.put((rank, "ok", hist))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(e) + "\n" + traceback.format_exc(), None))


@pytest.mark.timeout(900)
def test_train_entry_world2_uneven_shards_and_early_stop(tmp_path):
    """train_DC_focal.py under world size 2 (gloo, ATen-CPU path) with a training set whose length is NOT a multiple
    of the world size and a patience of one epoch: every rank must run the same number of steps (equal shards), stop
    in the same epoch (rank 0's validation Dice is broadcast) and hold bit-identical parameters after every epoch
    (replicated init + averaged gradients + identical Adam state) -- none of which may hang."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    # 23 tiles -> 4 test + 4 val + 15 train: shards of 7 and 8 before truncation, 3 vs 4 steps at batch 2
    args = ["--synthetic", "--synthetic_len", "23", "--img_size", "32", "--batch", "2", "--epochs", "4", "--patience", "1",
            "--workers", "0", "--in_channels", "1", "--device", "cpu", "--lr", "1e-3", "--ckpt_path", str(tmp_path / "b.pth")]
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=800) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in results), results
    h0, h1 = results[0][2], results[1][2]
    assert len(h0) == len(h1) >= 1                          # both ranks left the loop in the same epoch
    for a, b in zip(h0, h1):
        assert a["val_dice"] == b["val_dice"]              # rank 0's value everywhere
        assert a["param_checksum"] == b["param_checksum"]  # replicas bit-identical after each epoch
