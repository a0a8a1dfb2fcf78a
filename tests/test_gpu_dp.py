"""GPU: data-parallel rehearsal on ONE card -- two ranks share cuda:0 and exchange gradients over
gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  This exercises the
real thing end to end: HIP backward -> grad_ready_hook per block -> bucketed async all-reduce ->
finish() before autograd hands out the gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from models.model_2 import UNetDC
        from oracle import recipe
        from unet_dc_segmentation_amd.dp import DataParallel
        from utils.metrics_DC import focal_dice_loss
        torch.cuda.set_device(0)
        torch.manual_seed(100 + rank)                       # broadcast must make the replicas equal
        model = UNetDC(1, 1).cuda().train()
        dp = DataParallel(model, bucket_bytes=16 << 20)
        xs = [recipe.seeded_input(50 + r, (2, 1, 64, 64)).cuda() for r in range(world)]
        ts = [recipe.seeded_target(60 + r, (2, 1, 64, 64)).cuda() for r in range(world)]
        # reference: every rank's gradient computed locally WITHOUT the exchange
        hook, fin = model.grad_ready_hook, model.grad_sync_finish
        model.grad_ready_hook = model.grad_sync_finish = None
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        per_rank = []
        for r in range(world):
            model.load_state_dict(sd0)
            model.zero_grad(set_to_none=True)
            focal_dice_loss(model(xs[r]), ts[r]).backward()
            per_rank.append([p.grad.clone() for p in model.parameters()])
        # the data-parallel step
        model.load_state_dict(sd0)
        model.grad_ready_hook, model.grad_sync_finish = hook, fin
        model.zero_grad(set_to_none=True)
        focal_dice_loss(model(xs[rank]), ts[rank]).backward()
        torch.cuda.synchronize()
        worst = 0.0
        for i, p in enumerate(model.parameters()):
            mean = sum(g[i] for g in per_rank) / world
            err = float((p.grad - mean).abs().max())
            scale = float(mean.abs().max()) + 1e-12
            worst = max(worst, err / scale)
            assert err <= 1e-5 * scale + 1e-9, (i, err, scale)
        assert dp.stats["buckets"] >= 2 and dp.stats["elems"] == sum(p.numel() for p in model.parameters())
        # K optimizer steps on per-rank batches: the replicas must stay BITWISE identical (verification ii)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
        for k in range(3):
            opt.zero_grad(set_to_none=True)
            focal_dice_loss(model(xs[rank] * (1.0 - 0.1 * k)), ts[rank]).backward()
            opt.step()
        torch.cuda.synchronize()
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).cpu()
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(gathered[0], g) for g in gathered), "replicas diverged after 3 data-parallel steps"
        assert not torch.equal(chk, torch.stack([v.double().sum() for k, v in sd0.items()
                                                 if not ("running" in k or "num_batches" in k)]).cpu())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", worst, dp.stats["buckets"]))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(e) + "\n" + traceback.format_exc(), 0, 0))


@pytest.mark.timeout(900)
def test_hip_backward_with_bucketed_allreduce_two_ranks_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in results), results
    print("DP rehearsal:", results)


def _rccl_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
        from models.model_2 import UNetDC
        from oracle import recipe
        from unet_dc_segmentation_amd.dp import DataParallel
        from utils.metrics_DC import focal_dice_loss
        torch.manual_seed(7)
        model = UNetDC(1, 1).cuda().train()
        model.set_compute_dtype("bf16")
        x = recipe.seeded_input(50, (2, 1, 128, 128)).cuda()
        t = recipe.seeded_target(60, (2, 1, 128, 128)).cuda()
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        focal_dice_loss(model(x), t).backward()                          # no exchange
        torch.cuda.synchronize()
        ref = [p.grad.clone() for p in model.parameters()]
        model.load_state_dict(sd0)
        dp = DataParallel(model, bucket_bytes=16 << 20, single_rank_collectives=True)
        model.zero_grad(set_to_none=True)
        focal_dice_loss(model(x), t).backward()                          # every bucket goes through ncclAllReduce(AVG)
        torch.cuda.synchronize()
        for p, g in zip(model.parameters(), ref):
            assert torch.equal(p.grad, g)                                # AVG over one rank is the identity, bit for bit
        assert dp.stats["buckets"] >= 2 and dp.stats["elems"] == sum(p.numel() for p in model.parameters())
        dp.broadcast_buffers()
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", dp.stats["buckets"]))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(("FAIL: " + repr(e) + "\n" + traceback.format_exc(), 0))


@pytest.mark.timeout(600)
def test_rccl_branch_executes_on_one_rank():
    """The RCCL ('nccl') branch of dp.py -- async all_reduce(AVG) on slices of the flat gradient buffer behind the HIP
    backward, finish() on the compute stream -- run for real on the one card a test box has (one-rank group with
    single_rank_collectives=True).  The N > 1 arithmetic is covered by the gloo tests; this covers the backend calls."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=500)
    p.join(60)
    assert res[0] == "ok", res
