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


def _stage_order(model):
    """The ready ranges exactly as engine.UNetEngine.backward announces them: per conv3x3 stage (conv + BatchNorm), per
    up-convolution, the head -- in reverse parameters() order."""
    order = [[model.out_conv]]
    for lvl in (1, 2, 3, 4):
        d = getattr(model, f"dec{lvl}")
        order += [[d[3], d[4]], [d[0], d[1]], [getattr(model, f"upconv{lvl}")]]
    b = model.bottleneck
    order += [[b[3], b[4]], [b[0], b[1]]]
    for n in ("enc4", "enc3", "enc2", "enc1"):
        e = getattr(model, n)
        order += [[e[3], e[4]], [e[0], e[1]]]
    return order


def _worker(rank, world, port, q, full=True):
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
        dp = DataParallel(model)                            # default bucket policy: merge to >= 16 MiB, cut at 32 MiB
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
        dp.start_trace()                                    # the timeline bench.py puts into its N > 1 line (collective.*)
        dp.mark_step_start()
        for mods in _stage_order(model):
            ps = [p for m in mods for p in m.parameters()]
            dp._on_ready(flat, offs[id(ps[0])], offs[id(ps[-1])] + ps[-1].numel())
        dp.finish()
        coll = dp.stop_trace()
        assert coll["traced_steps"] == 1 and coll["exposed_ms_per_step"] >= 0.0 and "perf_counter" in coll["clock"]
        assert [round(b["MB"], 1) for b in coll["buckets"]] == [21.5, 18.9, 23.1, 23.1, 18.9, 17.7, 1.0]
        assert all(0.0 <= b["ready_at_ms"] <= b["done_at_ms"] for b in coll["buckets"])
        assert [b["ready_at_ms"] for b in coll["buckets"]] == sorted(b["ready_at_ms"] for b in coll["buckets"])
        assert round(dp.stats["elems"] * 4 / max(dp.stats["steps"], 1) / 1e6, 1) == 124.2        # bench.py's collective.payload_MB_per_step
        assert dp._trace is None                            # tracing is off again: the training loop records nothing
        # the bucket schedule DESIGN.md section 5 states (fp32 MB): head...dec4.3 | dec4.0 | upconv4 + bottleneck.3 cut in
        # two | bottleneck.0 | enc4 + enc3 | enc2 + enc1 at finish()
        assert [round(n * 4 / 1e6, 1) for n in dp.schedule] == [21.5, 18.9, 23.1, 23.1, 18.9, 17.7, 1.0], dp.schedule
        assert max(dp.schedule) * 4 <= 32 << 20
        mean_scale = sum(range(1, world + 1)) / world
        expect = torch.full((o,), mean_scale)
        expect[:1000] += torch.arange(1000.0) * mean_scale
        assert torch.allclose(flat, expect, rtol=0, atol=1e-5)
        assert dp.stats["buckets"] == 7 and dp.stats["elems"] == o             # every element exactly once
        if not full:
            dist.barrier()
            dist.destroy_process_group()
            q.put((rank, "ok"))
            return
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


@pytest.mark.timeout(600)
def test_data_parallel_world4_gloo_bucket_schedule():
    """Four ranks: coalesced replica broadcast, the per-stage ready ranges and the 16 / 32 MiB bucket policy; every
    element averaged exactly once."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, False)) for r in range(world)]
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
