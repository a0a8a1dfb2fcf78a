"""Data-parallel training: one process per GPU, gradients all-reduced with RCCL over xGMI,
overlapped with the backward pass.

The reference is single-device (train_DC_focal.py:208); this is the new exchange step that
BASELINE config 3 asks for.  Semantics: replicated parameters (rank 0 broadcast at
start), gradients averaged over ranks, per-rank BatchNorm BATCH statistics (the reference uses plain
BatchNorm2d, models/model_2.py:45,52, no SyncBN).  BatchNorm RUNNING statistics drift apart between ranks
during an epoch; ``broadcast_buffers()`` makes rank 0's the common ones (DistributedDataParallel does that
before every forward with its default broadcast_buffers=True; here the caller does it once before
validation / checkpointing, which is where running statistics are read -- train_DC_focal.py), and
``broadcast_scalar`` lets every rank take control-flow decisions (early stopping) from rank 0's value.

Mechanics: the HIP backward (engine.py) writes all gradients into ONE flat fp32 buffer in
``parameters()`` order and calls ``grad_ready_hook(flat, lo, hi)`` after the kernels of each block
have been enqueued -- blocks finish in exactly reverse parameter order, so ready ranges are
contiguous slices.  Slices are merged into buckets of ``bucket_bytes`` and all-reduced with
``async_op=True``: ProcessGroupNCCL (= RCCL on ROCm) runs the collective on its own side stream
after waiting for the work already enqueued on the compute stream, so communication of
decoder/bottleneck gradients overlaps the encoder backward kernels; ``finish()`` makes the compute
stream wait for the outstanding collectives before autograd hands the gradients to the optimizer.
Ready ranges arrive per STAGE (conv + BatchNorm parameters of one conv3x3 stage, one up-convolution, the head); they
are merged until a bucket holds >= ``bucket_bytes`` (16 MiB) and cut at ``max_bucket_bytes`` (32 MiB), which keeps each
xGMI link busy with few, large messages, lets the 37.7 MB of bottleneck.3 leave while bottleneck.0 still computes, and
leaves only the last ~1 MB (enc2 + enc1 gradients) to be reduced after the backward pass has finished.  The resulting
schedule for the U-Net-DC (fp32 bytes): 21.5 MB (head ... dec4.3) | 18.9 (dec4.0) | 23.0 + 23.0 (upconv4 + bottleneck.3)
| 18.9 (bottleneck.0) | 17.7 (enc4 + enc3) | 1.0 (enc2 + enc1, flushed by finish()).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, model, process_group=None, bucket_bytes=16 << 20, broadcast=True, single_rank_collectives=False,
                 max_bucket_bytes=32 << 20):
        """``single_rank_collectives``: issue the collectives even in a one-rank group (RCCL then runs every call of the
        exchange on one card: the rehearsal of the multi-GPU path that a one-GPU box allows, tests/test_gpu_dp.py)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend 'nccl' = RCCL on ROCm, or 'gloo')")
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.active = self.world > 1 or single_rank_collectives
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.max_elems = max(self.bucket_elems, max_bucket_bytes // 4)
        self.schedule = []              # (elements) of every bucket launched in the current step (tests / DESIGN table)
        self.backend = dist.get_backend(process_group)
        self._works = []
        self._pending = None            # (flat, lo, hi) not yet launched
        self._flat = None
        self.stats = {"buckets": 0, "elems": 0, "steps": 0}
        self._trace = None              # per-step timeline records while start_trace() ... stop_trace() (bench.py, after its timed region)
        self._probe = None              # side stream that only waits for collectives and records their completion events
        if broadcast:
            self.broadcast_state()
        model.grad_ready_hook = self._on_ready
        model.grad_sync_finish = self.finish

    # -------------------------------------------------------------- replication
    def _broadcast_coalesced(self, tensors):
        """ONE broadcast per dtype over a flat copy (82 parameters / 54 buffers would otherwise be 136 / 54 collectives
        of a few KB each)."""
        groups = {}
        for t in tensors:
            groups.setdefault(t.dtype, []).append(t)
        with torch.no_grad():
            for ts in groups.values():
                flat = torch.cat([t.detach().reshape(-1) for t in ts])
                dist.broadcast(flat, src=0, group=self.pg)
                o = 0
                for t in ts:
                    t.detach().copy_(flat[o:o + t.numel()].view_as(t))
                    o += t.numel()
        w = getattr(self.model, "_weights", None)
        if w is not None:
            w.invalidate()              # parameters were rewritten: the packed compute-type images follow at the next forward

    def broadcast_state(self):
        """Rank 0's parameters and buffers become everyone's (identical replicas at step 0)."""
        self._broadcast_coalesced(list(self.model.parameters()) + list(self.model.buffers()))

    def broadcast_buffers(self):
        """Rank 0's buffers (BatchNorm running_mean / running_var / num_batches_tracked) become everyone's."""
        if self.world == 1:
            return
        self._broadcast_coalesced(list(self.model.buffers()))

    # -------------------------------------------------------------- gradient exchange
    def _launch(self, flat, lo, hi):
        while hi - lo > self.max_elems:                  # cut from the top: that part has been ready longest
            cut = max(lo + (hi - lo) // 2, hi - self.max_elems) if hi - lo <= 2 * self.max_elems else hi - self.max_elems
            self._launch_one(flat, cut, hi)
            hi = cut
        self._launch_one(flat, lo, hi)

    def _launch_one(self, flat, lo, hi):
        view = flat[lo:hi]
        self.schedule.append(hi - lo)
        rec = self._trace_bucket(flat, hi - lo) if self._trace is not None else None
        if self.backend == "nccl":
            work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self._works.append((work, None, rec))
        else:  # gloo has no AVG
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((work, view, rec))
        if rec is not None and rec["device"]:
            # completion time of THIS collective: a side stream that does nothing but wait for it (Work.wait() makes the
            # current stream wait for RCCL's stream) and record an event -- the compute stream is not touched
            with torch.cuda.stream(self._probe):
                work.wait()
                rec["done"].record()
        self.stats["buckets"] += 1
        self.stats["elems"] += hi - lo

    # -------------------------------------------------------------- timeline of the exchange (instrumented passes only)
    def start_trace(self):
        """Record, for every step until stop_trace(): when each bucket became ready (all its producers enqueued: an event
        on the compute stream), when its all-reduce completed (an event on a probe stream behind RCCL's), and how long the
        compute stream stood waiting in finish() (= the EXPOSED part of the exchange).  Device events with RCCL on HIP
        tensors; host perf_counter stamps otherwise (gloo: Work.wait() blocks the host)."""
        self._trace = []

    def mark_step_start(self):
        """Optional: the zero of the per-bucket times (default: the first ready range of the step)."""
        if self._trace is not None:
            self._trace.append(dict(self._new_step_record(None), marked=True))

    def _new_step_record(self, flat):
        import time
        device = self.backend == "nccl" and torch.cuda.is_available()
        rec = {"device": device, "buckets": [], "fin": None, "closed": False}
        if device:
            if self._probe is None:
                self._probe = torch.cuda.Stream()
            rec["t0"] = torch.cuda.Event(enable_timing=True)
            rec["t0"].record()
        else:
            rec["t0"] = time.perf_counter()
        return rec

    def _trace_bucket(self, flat, elems):
        import time
        if not self._trace or self._trace[-1]["closed"]:
            self._trace.append(self._new_step_record(flat))
        step = self._trace[-1]
        b = {"elems": elems, "device": step["device"]}
        if step["device"]:
            b["ready"], b["done"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b["ready"].record()
        else:
            b["ready"], b["done"] = time.perf_counter(), None
        step["buckets"].append(b)
        return b

    def stop_trace(self):
        """-> {"exposed_ms_per_step", "buckets": [{"MB", "ready_at_ms", "done_at_ms"}], "traced_steps", "clock"} averaged over
        the traced steps (bucket k of every step is the same slice: the schedule is static)."""
        trace, self._trace = self._trace, None
        if not trace:
            return None
        device = trace[0]["device"]
        if device:
            torch.cuda.synchronize()
        steps = [t for t in trace if t["closed"] and t["buckets"]]
        if not steps:
            return None
        ms = (lambda a, b: a.elapsed_time(b)) if device else (lambda a, b: (b - a) * 1e3)
        nb = min(len(t["buckets"]) for t in steps)
        out = {"traced_steps": len(steps),
               "clock": "HIP events (compute stream / probe stream behind RCCL's)" if device else "host perf_counter (blocking backend)",
               "exposed_ms_per_step": sum(ms(*t["fin"]) for t in steps) / len(steps),
               "zero": "mark_step_start() (= start of the step's forward)" if "marked" in steps[0] else "first ready range of the step",
               "buckets": []}
        for k in range(nb):
            out["buckets"].append({
                "MB": steps[0]["buckets"][k]["elems"] * 4 / 1e6,
                "ready_at_ms": sum(ms(t["t0"], t["buckets"][k]["ready"]) for t in steps) / len(steps),
                "done_at_ms": sum(ms(t["t0"], t["buckets"][k]["done"]) for t in steps) / len(steps)})
        out["last_done_after_backward_ms"] = sum(ms(t["fin"][0], t["buckets"][nb - 1]["done"]) for t in steps) / len(steps)
        return out

    def _on_ready(self, flat, lo, hi):
        """Called by the backward schedule: gradients flat[lo:hi] are enqueued on the compute stream."""
        if not self.active:
            return
        if not self._works and self._pending is None:
            self.schedule = []                              # first range of a new step
        if self._pending is not None and self._pending[0] is flat and self._pending[1] == hi:
            lo, hi = lo, self._pending[2]                   # extend the pending range downwards
        elif self._pending is not None:
            self._launch(*self._pending)
        self._pending = (flat, lo, hi)
        if hi - lo >= self.bucket_elems:
            self._launch(flat, lo, hi)
            self._pending = None

    def finish(self):
        """Flush the last bucket and make the compute stream wait for every collective."""
        import time
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        step = self._trace[-1] if self._trace else None
        if step is not None and step["closed"]:
            step = None
        if step is not None:
            f0 = torch.cuda.Event(enable_timing=True) if step["device"] else time.perf_counter()
            if step["device"]:
                f0.record()
        for work, view, rec in self._works:
            work.wait()
            if rec is not None and not rec["device"]:
                rec["done"] = time.perf_counter()
            if view is not None:
                view.div_(self.world)
        if step is not None:
            f1 = torch.cuda.Event(enable_timing=True) if step["device"] else time.perf_counter()
            if step["device"]:
                f1.record()
            step["fin"], step["closed"] = (f0, f1), True
        self._works.clear()
        self.stats["steps"] += 1

    # -------------------------------------------------------------- fallback for the ATen-CPU path
    def sync_gradients(self):
        """All-reduce ``param.grad`` of every parameter (used when backward did not go through the
        HIP engine, i.e. CPU tensors with the gloo backend)."""
        params = [p for p in self.model.parameters() if p.grad is not None]
        if not self.active or not params:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        n = flat.numel()
        hi = n
        while hi > 0:
            lo = max(0, hi - self.bucket_elems)
            self._on_ready(flat, lo, hi)
            hi = lo
        self.finish()
        o = 0
        for p in params:
            p.grad.copy_(flat[o:o + p.numel()].view_as(p.grad))
            o += p.numel()


def broadcast_scalar(value, device="cpu", src=0, group=None):
    """Rank ``src``'s Python float on every rank (no-op without an initialised process group)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.broadcast(t, src=src, group=group)
    return float(t.item())


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("UNETDC_DP_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
