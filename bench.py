#!/usr/bin/env python3
"""Headline benchmark: images/sec of one U-Net-DC training step (forward + Focal/Dice loss +
backward, plus the Adam step and -- for N > 1 -- the RCCL gradient all-reduce), 512x512x1,
batch 8 per GPU, bf16 storage / fp32 accumulate (BASELINE.json metric, configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With ``--gpus N`` (N > 1) and no torchrun environment, bench.py launches its own N ranks: the parent -- before any
HIP call -- checks that N devices exist (else it exits non-zero: it never reports a one-rank number as an N-GPU
one), starts N child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and
fails if any rank fails.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline     -- the dominant kernel (bf16 implicit-GEMM conv): algorithmic dense FLOPs per launch /
                  average launch time measured live with HIP events on the launch stream over the
                  timed steps, against the 2.5 PFLOP/s dense bf16 MFMA peak;
  hbm_leg      -- the HBM-bound kernels (BatchNorm passes, first layer, head): algorithmic bytes / launch time from a
                  short HIP-event-instrumented pass after the timed region, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU port of the reference path (oracle/unetdc_torch_cpu.py, the same ATen ops
                  the reference calls) timed on this host's cores on a bounded sample (N = 1 only): median of 3 steps.
Everything inside the timed region is real work: no step is skipped, the optimizer step and the
weight re-packing it triggers are included; the reference loop's per-step .item()/.cpu() syncs
(train_DC_focal.py:257-269) are not part of the metric and are excluded.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
# empirical ceilings measured on MI355X (MI355X_MICROARCH.md): float4 copy 6.29 TB/s; the best plain 256 x 256 bf16 GEMM on random
# data 1.32-1.47 PFLOP/s (the chip lowers its clock under MFMA load: "DVFS give-back").  Reported NEXT to the datasheet fractions.
EMPIRICAL_HBM_GBS = 6290.0
EMPIRICAL_BF16_TFLOPS = 1470.0


def synthetic_batch(seed, n, h, w, cin=1, discs=200):
    """U[0,1) noise + `discs` bright discs per image ("droplets"); target = disc mask."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, cin, h, w), dtype=np.float32) * 0.6
    t = np.zeros((n, 1, h, w), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        cy, cx = rng.integers(0, h, discs), rng.integers(0, w, discs)
        rad = rng.uniform(1, 12, discs)
        for k in range(discs):
            r = int(rad[k]) + 1
            y0, y1, x0, x1 = max(cy[k] - r, 0), min(cy[k] + r + 1, h), max(cx[k] - r, 0), min(cx[k] + r + 1, w)
            m = (yy[y0:y1, x0:x1] - cy[k]) ** 2 + (xx[y0:y1, x0:x1] - cx[k]) ** 2 <= rad[k] ** 2
            t[i, 0, y0:y1, x0:x1][m] = 1.0
    x += 0.4 * t
    return torch.from_numpy(x), torch.from_numpy(t)


def igemm_flops(name, a, es=2):
    """(nominal dense FLOPs incl. padded taps, algorithmic bytes = input + weights + output read/written
    once) of one implicit-GEMM launch, from its C-ABI arguments."""
    if name in ("unetdc_conv3x3_fwd", "unetdc_conv3x3_fwd_bnin"):      # (bnin: input = the raw output of the stage in front)
        n, h, w, cin, cout = a[10:15] if name == "unetdc_conv3x3_fwd" else a[12:17]
        return 2.0 * n * h * w * cout * cin * 9, (n * h * w * (cin + cout) + 9 * cin * cout) * es
    if name == "unetdc_conv3x3_dgrad":
        n, h, w, cin, cout = a[5:10]
        return 2.0 * n * h * w * cout * cin * 9, (n * h * w * (cin + cout) + 9 * cin * cout) * es
    if name == "unetdc_convT2x2_fwd":
        n, h, w, cin, cout = a[6:11]
        return 2.0 * n * h * w * cin * 4 * cout, (n * h * w * (cin + 4 * cout) + 4 * cin * cout) * es
    if name == "unetdc_convT2x2_dgrad":
        n, h, w, cin, cout = a[5:10]
        return 2.0 * n * h * w * cin * 4 * cout, (n * h * w * (cin + 4 * cout) + 4 * cin * cout) * es
    # dgrad fused with the BatchNorm-backward reduction: also reads the consumer's saved conv output once
    if name == "unetdc_conv3x3_dgrad_bnstats":
        n, h, w, cin, cout = a[14:19]
        return 2.0 * n * h * w * cout * cin * 9, (n * h * w * (2 * cin + cout) + 9 * cin * cout) * es
    if name == "unetdc_convT2x2_dgrad_bnstats":
        n, h, w, cin, cout = a[14:19]
        return 2.0 * n * h * w * cin * 4 * cout, (n * h * w * (2 * cin + 4 * cout) + 4 * cin * cout) * es
    if name == "unetdc_conv3x3_dgrad_colsum":          # dgrad + per-channel sums of its output (no extra traffic)
        n, h, w, cin, cout = a[10:15]
        return 2.0 * n * h * w * cout * cin * 9, (n * h * w * (cin + cout) + 9 * cin * cout) * es
    raise KeyError(name)


def pmc_traffic(kernel):
    """Per-launch bytes past the L2 for `kernel` from the committed rocprofv3 PMC summary (separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 x2 correction on the read side; tools/pmc_traffic.sh), or None -- also None when the
    summary was measured on different kernel sources than the ones built here (source-hash stamp)."""
    # the C ABI reports "igemm_lattice_wide_kernel<1>" (+ " bnin" for the input-normalising instantiation); the symbol rocprofv3
    # prints carries every template argument: igemm_lattice_wide_kernel<1, false> / <1, true>
    base = kernel.split(" ")[0]
    names = [kernel, base]
    if base.startswith("igemm_lattice_wide_kernel<") and base.endswith(">"):
        names.insert(0, base[:-1] + (", true>" if kernel.endswith(" bnin") else ", false>"))
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        try:
            doc = json.load(open(path))
            table = doc["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        if doc.get("kernel_source_sha16") != kernel_source_hash():
            continue
        # rocprofv3 prints Itanium-mangled names for the __bf16 instantiations (c++filt cannot demangle
        # DF16b), e.g. igemm_dma_kernel<__bf16, 2, 4, 4> -> igemm_dma_kernelIDF16bLi2ELi4ELi4EE
        base, _, targs = kernel.partition("<")
        frag = base + "I" + "".join("DF16b" if a.strip() == "__bf16" else ("f" if a.strip() == "float" else f"Li{a.strip()}E")
                                    for a in targs.rstrip(">").split(",")) + "E" if targs else base
        for cand in names:
            for sym, v in table.items():
                if cand in sym:
                    return v["bytes_corrected"]
        for sym, v in table.items():
            if frag in sym:
                return v["bytes_corrected"]
    return None


def host_cores():
    """Cores this process may actually use: cgroup quota if there is one, else the affinity mask,
    capped at 16 (the CPU share of a one-GPU box; os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("UNETDC_BENCH_CPU_CORES", "16"))))


def per_layer_table(step, args):
    """Diagnostic: time every C-ABI call of one step with HIP events and print name/shape/ms/TFLOP/s."""
    from unet_dc_segmentation_amd import _lib
    names = [n for n in _lib.SIGNATURES if n not in ("unetdc_version", "unetdc_last_error") and "workspace" not in n
             and "rows" not in n]
    _lib.start_timing(names)
    for _ in range(3):
        step()
    rec = _lib.stop_timing()
    per = len(rec) // 3
    rows = rec[2 * per:]
    tot = {}
    for tagged, a, ms in rows:
        name = tagged.split("|")[0]
        fl, shape = 0.0, ""
        ints = [v for v in a if isinstance(v, int) and 0 < v < 100000]
        if name in ("unetdc_conv3x3_fwd", "unetdc_conv3x3_fwd_bnin", "unetdc_conv3x3_dgrad", "unetdc_convT2x2_fwd",
                    "unetdc_convT2x2_dgrad", "unetdc_conv3x3_dgrad_bnstats", "unetdc_convT2x2_dgrad_bnstats",
                    "unetdc_conv3x3_dgrad_colsum"):
            fl, _ = igemm_flops(name, a)
        elif name == "unetdc_conv3x3_wgrad":
            n, h, w, cin, cout = a[7:12]
            fl = 2.0 * n * h * w * cin * cout * 9
        elif name == "unetdc_conv3x3_wgrad_bnin":
            n, h, w, cin, cout = a[9:14]
            fl = 2.0 * n * h * w * cin * cout * 9
        elif name == "unetdc_convT2x2_wgrad":
            n, h, w, cin, cout = a[7:12]
            fl = 2.0 * n * h * w * cin * cout * 4
        shape = "x".join(str(v) for v in ints[-8:])
        print(f"{name:32s} {shape:40s} {ms * 1e3:9.1f} us {fl / (ms * 1e-3) / 1e12 if fl else 0:8.1f} TF", file=sys.stderr)
        t = tot.setdefault(name, [0.0, 0.0])
        t[0] += ms
        t[1] += fl
    print("---- per entry point (one step) ----", file=sys.stderr)
    for name, (ms, fl) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"{name:32s} {ms:8.3f} ms {fl / (ms * 1e-3) / 1e12 if fl else 0:8.1f} TF", file=sys.stderr)
    print(f"sum {sum(v[0] for v in tot.values()):.3f} ms", file=sys.stderr)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch, h, w, cin, iters=3):
    """CPU port of the reference path: forward + focal/dice loss + backward on the host cores (one warm-up,
    >= 3 timed iterations, median, core count and CPU model stated)."""
    from oracle import unetdc_torch_cpu as otc
    from models.model_2 import UNetDC
    torch.manual_seed(0)
    model = UNetDC(in_channels=cin, out_channels=1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cores = host_cores()
    torch.set_num_threads(cores)
    xw, tw = synthetic_batch(99, 1, h, w, cin, discs=20)
    otc.train_step_grads(xw, tw, sd, dict(model.DILATIONS))          # warm-up (oneDNN primitive caches)
    x, t = synthetic_batch(100, batch, h, w, cin, discs=50)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        otc.train_step_grads(x, t, {k: v.clone() for k, v in sd.items()}, dict(model.DILATIONS))
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_name(),
            "sample": f"median of {iters} fwd+loss+bwd steps of bs={batch} {h}x{w}x{cin} fp32 on PyTorch-CPU "
                      f"(oracle/unetdc_torch_cpu.py) after a bs=1 warm-up; step times "
                      + "/".join(f"{v:.1f}" for v in times) + " s"}


def kernel_source_hash():
    """sha256 over the HIP sources the shared library is built from: stamps profiles/*_pmc_traffic.json so that a
    traffic figure measured on OTHER kernels is never reported for these."""
    import hashlib
    from unet_dc_segmentation_amd import build as b
    h = hashlib.sha256()
    for f in sorted(b.SOURCES + b.HEADERS):
        with open(os.path.join(b.CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def executed_fraction(a, name, blocks16=False):
    """Fraction of the nominal taps the LDS-DMA conv kernel executes: a 256-pixel block skips a tap when no pixel of the
    block can reach the image through it (block-level tap skipping, igemm_dma16.hip).  Row-major blocks of the flattened
    map (blocks16 False): d = 16 on a 32 x 32 map executes 6 of 9 taps (an 8-row block reaches the image through the
    centre row of taps and one of the two outer rows).  16 x 16 pixel blocks (blocks16 True, the kernel's block order for
    d % 16 == 0): exactly the in-bounds tap-pixel pairs, 4 of 9 there.  1.0 for transposed convs and the halo-patch kernel."""
    if "convT" in name:
        return 1.0
    if name == "unetdc_conv3x3_fwd":
        n, h, w, d = a[10], a[11], a[12], a[15]
    elif name == "unetdc_conv3x3_fwd_bnin":
        n, h, w, d = a[12], a[13], a[14], a[17]
    elif name == "unetdc_conv3x3_dgrad":
        n, h, w, d = a[5], a[6], a[7], a[10]
    elif name == "unetdc_conv3x3_dgrad_bnstats":
        n, h, w, d = a[14], a[15], a[16], a[19]
    elif name == "unetdc_conv3x3_dgrad_colsum":
        n, h, w, d = a[10], a[11], a[12], a[15]
    else:
        return 1.0
    bm, live, total = 256, 0, 0
    if blocks16:
        for by in range(0, h, 16):
            for bx in range(0, w, 16):
                for t in range(9):
                    dy, dx = (t // 3 - 1) * d, (t % 3 - 1) * d
                    total += 1
                    if by + 15 + dy >= 0 and by + dy < h and bx + 15 + dx >= 0 and bx + dx < w:
                        live += 1
        return live / total
    for m0 in range(0, h * w, bm):                        # blocks never straddle images when h*w % 256 == 0
        m1 = min(m0 + bm, h * w) - 1
        y0, x0, y1, x1 = m0 // w, m0 % w, m1 // w, m1 % w
        bx0, bx1 = (x0, x1) if y0 == y1 else (0, w - 1)
        for t in range(9):
            dy, dx = (t // 3 - 1) * d, (t % 3 - 1) * d
            total += 1
            if y1 + dy >= 0 and y0 + dy < h and bx1 + dx >= 0 and bx0 + dx < w:
                live += 1
    return live / total if (h * w) % bm == 0 else 1.0


def hbm_bytes(name, a, es):
    """Algorithmic bytes of one launch of an HBM-bound entry point (each operand read or written once)."""
    if name == "unetdc_bn_relu_apply":                     # y -> a (+ pooled)
        n, h, w, c = a[8:12]
        return n * h * w * c * es * (2 + (0.25 if a[6] else 0))
    if name == "unetdc_bn_relu_bwd":                       # dskip (+ dpool) + y -> dy
        n, h, w, c = a[20:24]
        return n * h * w * c * es * ((1 if a[0] else 0) + (0.25 if a[2] else 0) + 2)
    if name == "unetdc_head_fwd":                          # a -> probs (fp32)
        n, h, w, c, oc = a[5:10]
        return n * h * w * (c * es + oc * 4)
    if name == "unetdc_head_fwd_bn":                       # y (normalised on load) -> probs (fp32)
        n, h, w, c, oc = a[7:12]
        return n * h * w * (c * es + oc * 4)
    if name == "unetdc_head_bwd":                          # dprobs, probs, a -> da
        n, h, w, c, oc = a[11:16]
        return n * h * w * (2 * c * es + 2 * oc * 4)
    if name == "unetdc_head_bwd_bnstats":                  # dprobs, probs, [a,] y (of the last stage) -> [da]
        n, h, w, c, oc = a[20:25]
        return n * h * w * (((2 if a[2] else 1) + (1 if a[5] else 0)) * c * es + 2 * oc * 4)
    if name == "unetdc_bn_relu_bwd_head":                  # dprobs, probs, y -> dy
        n, h, w, c = a[19:23]
        return n * h * w * (2 * c * es + 2 * 4)
    if name == "unetdc_conv3x3_first_fwd":                 # x (fp32 NCHW) -> y
        n, h, w, cin, cout = a[8:13]
        return n * h * w * (cin * 4 + cout * es)
    if name == "unetdc_conv3x3_first_wgrad_bn":            # x, dz, y -> dW (BatchNorm backward on load)
        n, h, w, cin, cout = a[13:18]
        return n * h * w * (cin * 4 + 2 * cout * es)
    if name == "unetdc_conv3x3_first_wgrad":               # x, dy -> dW
        n, h, w, cin, cout = a[6:11]
        return n * h * w * (cin * 4 + cout * es)
    raise KeyError(name)


HBM_CALLS = ["unetdc_bn_relu_apply", "unetdc_bn_relu_bwd", "unetdc_bn_relu_bwd_head", "unetdc_head_fwd", "unetdc_head_fwd_bn",
             "unetdc_head_bwd", "unetdc_head_bwd_bnstats",
             "unetdc_conv3x3_first_fwd", "unetdc_conv3x3_first_wgrad", "unetdc_conv3x3_first_wgrad_bn"]


def hbm_leg(step, es, nsteps=3):
    """Instrumented pass (separate from the timed region): per HBM-bound entry point, GB/s vs the 8 TB/s peak."""
    from unet_dc_segmentation_amd import _lib
    _lib.start_timing(HBM_CALLS)
    for _ in range(nsteps):
        step()
    rec = _lib.stop_timing()
    agg = {}
    for tagged, a, ms in rec:
        name = tagged.split("|")[0]
        g = agg.setdefault(name, [0.0, 0.0, 0])
        g[0] += hbm_bytes(name, a, es)
        g[1] += ms
        g[2] += 1
    rows = []
    for name, (nbytes, ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({"entry": name, "launches_per_step": cnt / nsteps, "ms_per_step": ms / nsteps,
                     "algorithmic_GB_per_step": nbytes / nsteps / 1e9, "achieved_GBps": gbs, "frac": gbs / PEAK_HBM_GBS})
    tot_b, tot_ms = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
    return {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else None,
            "frac": tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if tot_ms else None,
            "empirical_peak": EMPIRICAL_HBM_GBS, "frac_of_empirical": tot_b / (tot_ms * 1e-3) / 1e9 / EMPIRICAL_HBM_GBS if tot_ms else None,
            "ms_per_step": tot_ms / nsteps,
            "measured": f"HIP events around each call, {nsteps} instrumented steps after the timed region", "entries": rows}


def synthetic_micrograph(seed, h=1040, w=1388, discs=300):
    """A decoded RGB micrograph stand-in (uint8 [h, w, 3]): smooth background gradient + noise + bright discs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 40.0 + 30.0 * (yy / h) + 20.0 * (xx / w) + rng.normal(0, 6, (h, w))
    for _ in range(discs):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.uniform(2, 18)
        y0, y1, x0, x1 = max(int(cy - r), 0), min(int(cy + r) + 1, h), max(int(cx - r), 0), min(int(cx + r) + 1, w)
        m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] += 90.0
    g = np.clip(img, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.stack([g, g, (g * 0.9).astype(np.uint8)], axis=-1))


def quantify_bench(args, real_stdout):
    """--mode quantify: the droplet-quantification flow of quantify_droplets_batch.py (preprocess -> forward -> threshold
    -> resize to the original size -> connected components -> per-droplet table) on decoded 1040 x 1388 RGB arrays that
    sit in host memory (PNG decode and the PNG/CSV writes of the script are excluded on both sides), batch 8; next to it
    the same flow through the script's own DEVICE == "cpu" path on a bounded sample."""
    import quantify_droplets_batch as qdb
    from models.model_2 import UNetDC
    from unet_dc_segmentation_amd.droplets import mask_and_droplets_batch
    from unet_dc_segmentation_amd.preprocess import preprocess_device
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UNetDC(in_channels=3, out_channels=1)
    model = model.to(dev).eval()
    model.set_compute_dtype(args.dtype)
    B, size, radius, thresh, min_area = args.batch, 512, 50, 0.3, 1
    imgs = [synthetic_micrograph(7 + i) for i in range(B)]
    hw = [im.shape[:2] for im in imgs]
    with torch.no_grad():                                  # random init: calibrate the head bias so that ~10 % of the pixels
        p0 = model(torch.stack([preprocess_device(im, radius, size, dev) for im in imgs])).clamp(1e-6, 1 - 1e-6)   # are "droplet"
        z = torch.log(p0 / (1 - p0)).flatten()[::7]
        model.out_conv.bias += float(np.log(0.3 / 0.7)) - float(torch.quantile(z, 0.9))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def batch_once():
        with torch.no_grad():
            x = torch.stack([preprocess_device(im, radius, size, dev) for im in imgs])
            probs = model(x)
            res = mask_and_droplets_batch(probs[:, 0], thresh, hw, min_area)
            tabs = [qdb._droplet_table(a, cy, cx, None) for _, a, cy, cx in res]
            masks = [m.cpu() for m, *_ in res]             # the script writes them as PNGs
        return tabs, masks

    tabs, _ = batch_once()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        batch_once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tabs, _ = batch_once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out = {"metric": "images/sec droplet quantification (preprocess + forward + mask + connected components), "
                     "1040x1388 RGB -> 512x512x3 U-Net-DC, bs=8",
           "value": B / dt, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic",
           "config": {"workload": "quantify_droplets_batch.py flow on decoded host arrays: H2D, rolling ball r=50, bilinear "
                                  "resize, eval forward, threshold 0.3, resize to 1040x1388, 4-connected components, "
                                  "droplet tables + uint8 masks back on the host; PNG decode / PNG+CSV writes excluded",
                      "global_batch": B, "parallelism": "dp1"},
           "droplets_per_image": float(np.mean([len(t) for t in tabs]))}
    if not args.no_cpu_baseline:
        ncpu = 2
        cores = host_cores()
        torch.set_num_threads(cores)
        from utils.data_loader import resize_image, rolling_ball_correction_rgb
        from unet_dc_segmentation_amd.droplets import resize_mask_like_reference
        cpu_model = UNetDC(in_channels=3, out_channels=1)
        cpu_model.load_state_dict(sd)
        cpu_model.eval()
        t0 = time.perf_counter()
        with torch.no_grad():
            xs = []
            for im in imgs[:ncpu]:
                c = rolling_ball_correction_rgb(im, radius)
                xs.append(torch.from_numpy(resize_image(c, size).astype(np.float32) / 255.0).permute(2, 0, 1))
            pr = cpu_model(torch.stack(xs))
            m512 = (pr[:, 0] > thresh).to(torch.uint8).numpy()
            for i in range(ncpu):
                qdb.quantify(resize_mask_like_reference(m512[i], hw[i][1], hw[i][0]), min_area, None)
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": ncpu / ct, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "cpu_model": cpu_model_name(),
                               "sample": f"{ncpu} images through the script's DEVICE == 'cpu' path (numpy restatement of the "
                                         f"OpenCV operators, ATen-CPU fp32 forward, SciPy labelling): {ct:.1f} s"}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def self_launch(args, argv):
    """--gpus N without a torchrun environment: become the launcher.  Nothing here touches HIP (device_count() reads
    sysfs on this image); the children are fresh processes, never an exec of an initialised one."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n:
        print(f"[bench] --gpus {n} requested but this host exposes {have} GPU(s): refusing to report a {have}-GPU "
              f"number as a {n}-GPU one", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}: failing", file=sys.stderr)
        sys.stdout.write(out0 or "")
        return 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


IGEMM_CALLS = ["unetdc_conv3x3_fwd", "unetdc_conv3x3_fwd_bnin", "unetdc_conv3x3_dgrad", "unetdc_convT2x2_fwd", "unetdc_convT2x2_dgrad",
               "unetdc_conv3x3_dgrad_bnstats", "unetdc_convT2x2_dgrad_bnstats", "unetdc_conv3x3_dgrad_colsum"]


def _pct(sorted_vals, q):
    return sorted_vals[min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))]


def timed_region(step, steps, make_event=None, sync=None, barrier=None, alloc_stats=None, warmup=0):
    """`warmup` untimed calls of step(), then the timed region and NOTHING else: `steps` calls of step() between barrier + device
    sync on both sides.  The collector work (full collection + freeze, 40-60 ms of host time during which the device would
    sit idle and come back at a lower clock: the first timed steps then read 12.3 / 11.0 / 10.8 ms instead of 10.4) is done IN
    FRONT of the warm-up steps, so that only the barrier + sync separate the last warm-up step from the first timed one.

    No per-call instrumentation runs in here (the per-kernel roofline legs are separate passes AFTER it); what is recorded costs
    one pre-created event record and one perf_counter() per step:
      * device time per step  -- events at the step boundaries on the launch stream,
      * host time per step    -- perf_counter() stamps when the host has finished ENQUEUEING each step,
      * garbage collections   -- gc.callbacks (generation, duration); the cyclic collector is disabled for the region after a
                                 full collection + gc.freeze() (a generation-2 pass over torch's heap costs 40-60 ms of host time),
      * allocator activity    -- deltas of the caching allocator's device-malloc / retry counters.
    make_event / sync / barrier / alloc_stats are injectable so that the CPU suite can run this very function."""
    import gc
    make_event = make_event or (lambda: torch.cuda.Event(enable_timing=True))
    sync = sync or torch.cuda.synchronize
    alloc_stats = alloc_stats or (lambda: {k: torch.cuda.memory_stats().get(k, 0) for k in ("num_device_alloc", "num_alloc_retries", "num_ooms")})
    marks = [make_event() for _ in range(steps + 1)]            # created BEFORE the region
    host = [0.0] * (steps + 1)
    gc_events, gc_t0 = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_events.append({"generation": info.get("generation"), "ms": (time.perf_counter() - gc_t0[0]) * 1e3,
                              "at_ms": (gc_t0[0] - host[0]) * 1e3, "collected": info.get("collected")})

    gc.collect()
    gc.freeze()                 # everything alive now (torch, the model, the engine) leaves the collector's generations
    was_enabled = gc.isenabled()
    gc.disable()
    gc.callbacks.append(on_gc)
    loss = None
    try:
        for _ in range(warmup):
            step()
        a0 = alloc_stats()
        if barrier:
            barrier()
        sync()
        t0 = time.perf_counter()
        host[0] = t0
        marks[0].record()
        for i in range(steps):
            loss = step()
            marks[i + 1].record()
            host[i + 1] = time.perf_counter()
        sync()
        if barrier:
            barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.callbacks.remove(on_gc)
        if was_enabled:
            gc.enable()
        gc.unfreeze()
    a1 = alloc_stats()
    dev_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    host_ms = [(host[i + 1] - host[i]) * 1e3 for i in range(steps)]
    srt = sorted(dev_ms)
    imax = dev_ms.index(srt[-1])
    stats = {"median": _pct(srt, 0.5), "p10": _pct(srt, 0.1), "p90": _pct(srt, 0.9), "min": srt[0], "max": srt[-1],
             "max_at_step": imax, "n": steps,
             "host_ms_at_max": host_ms[imax], "host_ms_median": _pct(sorted(host_ms), 0.5), "host_ms_max": max(host_ms),
             "host_max_at_step": host_ms.index(max(host_ms)),
             # how far the host ran ahead of the device when the slowest step was enqueued (ms of queued work): > 0 means a
             # host pause of that size would have been hidden
             "host_lead_ms_at_max": sum(dev_ms[:imax + 1]) - (host[imax + 1] - host[0]) * 1e3,
             "device_ms_all": [round(v, 3) for v in dev_ms], "host_ms_all": [round(v, 3) for v in host_ms],
             "gc_events": gc_events, "gc": "collect + freeze + disable around the timed region",
             "device_allocs_in_timed_region": a1.get("num_device_alloc", 0) - a0.get("num_device_alloc", 0),
             "alloc_retries_in_timed_region": a1.get("num_alloc_retries", 0) - a0.get("num_alloc_retries", 0),
             "measured": "HIP events on the launch stream at the step boundaries + host perf_counter stamps after each "
                         "step's enqueue (rank 0); no per-call events inside the region"}
    return elapsed, stats, loss


def igemm_leg(step, dtype, mode, nsteps=3, headline=False):
    """Instrumented pass AFTER the timed region: HIP events around every implicit-GEMM C-ABI call of `nsteps` steps (on the
    stream the kernels are launched on), grouped per dispatched kernel; the dominant one (most time per step) gives `roofline`.
    Per-kernel time = MEDIAN over its launches of the same shape group (one outlier launch does not move it)."""
    from unet_dc_segmentation_amd import _lib
    es = 2 if dtype == "bf16" else 4
    _lib.start_timing(IGEMM_CALLS)
    for _ in range(nsteps):
        step()
    records = _lib.stop_timing()
    groups = {}
    for tagged, a, ms in records:
        name, key = tagged.split("|")              # C-ABI entry point | dispatched kernel symbol
        fl, nbytes = igemm_flops(name, a, es)
        gsum = groups.setdefault(key, [0.0, [], 0, 0.0, 0.0])
        gsum[0] += fl
        gsum[1].append(ms)
        gsum[2] += 1
        gsum[3] += nbytes
        gsum[4] += fl * (executed_fraction(a, name, "blocks16x16" in key) if "dma" in key else 1.0)
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    kernels = []
    for key, (fl, mss, cnt, nbytes, flx) in groups.items():
        ms = sum(mss)
        kernels.append({"kernel": key, "launches_per_step": cnt / nsteps, "avg_launch_ms": ms / cnt,
                        "median_launch_ms": sorted(mss)[len(mss) // 2],
                        "gflop_per_launch": fl / cnt / 1e9, "tflops": fl / (ms * 1e-3) / 1e12,
                        "executed_gflop_per_launch": flx / cnt / 1e9, "tflops_executed": flx / (ms * 1e-3) / 1e12,
                        "ms_per_step": ms / nsteps, "algorithmic_bytes_per_launch": nbytes / cnt})
    kernels.sort(key=lambda k: -k["ms_per_step"])
    dom = kernels[0]
    roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": peak,
                "unit": "TFLOP/s", "frac": dom["tflops"] / peak,
                "empirical_peak": EMPIRICAL_BF16_TFLOPS if dtype == "bf16" else None,
                "frac_of_empirical": dom["tflops"] / EMPIRICAL_BF16_TFLOPS if dtype == "bf16" else None,
                "traffic": pmc_traffic(dom["kernel"]) if headline else None,
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "flop_per_launch": dom["gflop_per_launch"] * 1e9,
                "executed_flop_per_launch": dom["executed_gflop_per_launch"] * 1e9,
                "flop_note": "nominal dense FLOPs, padded taps included (SURVEY 8d); executed = after block-level "
                             "skipping of taps that cannot reach the image",
                "avg_launch_ms": dom["avg_launch_ms"], "launches_per_step": dom["launches_per_step"],
                "measured": f"HIP events around each call on the launch stream, {nsteps} instrumented steps AFTER the timed region",
                "igemm_ms_per_step": sum(k["ms_per_step"] for k in kernels)}
    if headline:
        roofline["all_igemm_kernels"] = kernels
    else:
        roofline["top_kernels"] = [{k: v for k, v in kk.items() if k in ("kernel", "launches_per_step", "tflops", "ms_per_step")}
                                   for kk in kernels[:3]]
    return roofline


NOMINAL_GFLOP_IMG = {("train", 512): 1153.9, ("infer", 512): 384.74, ("train", 1024): 4615.6, ("infer", 1024): 1538.94}   # SURVEY 8d, C_in = 1


class Job:
    """One workload of this rank: module, optimizer, resident synthetic batch, step()."""

    def __init__(self, mode, dtype, batch, size, cin, arch, adam, dev, rank=0):
        from utils.metrics_DC import focal_dice_loss
        if arch == "unetdc":
            from models.model_2 import UNetDC as Net
        else:
            from models.model import UNet as Net
        self.mode, self.dtype, self.batch, self.size, self.cin, self.arch, self.adam = mode, dtype, batch, size, cin, arch, adam
        torch.manual_seed(0)                                   # identical replicas
        self.model = Net(in_channels=cin, out_channels=1).to(dev)
        self.model.train() if mode == "train" else self.model.eval()
        self.model.set_compute_dtype(dtype)
        self.opt = None
        x, t = synthetic_batch(1000 + rank, batch, size, size, cin)
        self.x, self.t = x.to(dev), t.to(dev)
        self._loss = focal_dice_loss

    def make_optimizer(self):
        # train_DC_focal.py:224 (Adam, lr 1e-3): the same update rule in ONE HIP kernel that also rewrites the packed weight
        # images (unet_dc_segmentation_amd/optim.py); --adam fused / foreach select torch.optim.Adam for comparison
        if self.mode != "train":
            return
        if self.adam == "hip":
            from unet_dc_segmentation_amd.optim import FusedAdam
            self.opt = FusedAdam(self.model, lr=1e-3)
        else:
            self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3, fused=(self.adam == "fused"))

    def step(self):
        if self.mode == "train":
            self.opt.zero_grad(set_to_none=True)
            p = self.model(self.x)
            loss = self._loss(p, self.t, alpha=1.0, gamma=2.0, ratio=0.3)      # train_DC_focal.py:222
            loss.backward()
            self.opt.step()
            return loss
        with torch.no_grad():                                   # quantify_droplets_batch.py:51-56 on device
            p = self.model(self.x)
            return (p > 0.3).sum()

    def workload(self, world=1):
        if self.mode == "train":
            return (f"train step (fwd + Focal/Dice loss + bwd + Adam{' + RCCL grad all-reduce' if world > 1 else ''}) "
                    f"{self.arch} bs={self.batch}/GPU {self.size}x{self.size}x{self.cin} {self.dtype}")
        return f"eval forward + 0.3 threshold, {self.arch} bs={self.batch}/GPU {self.size}x{self.size}x{self.cin} {self.dtype}"


def secondary_entry(dev, label, mode, dtype, batch, size, steps, warmup=2):
    """One more single-GPU BASELINE configuration, measured like the headline (own timed region, own instrumented pass)."""
    job = Job(mode, dtype, batch, size, 1, "unetdc", "hip", dev)
    job.make_optimizer()
    job.step()                                             # initialisation (buffers, descriptor tables)
    torch.cuda.synchronize()
    elapsed, stats, _ = timed_region(job.step, steps, warmup=warmup)
    roof = igemm_leg(job.step, dtype, mode, nsteps=2)
    imgs = batch * steps
    gf = NOMINAL_GFLOP_IMG.get((mode, size))
    ent = {"config": {"workload": job.workload() + f" ({label})", "global_batch": batch, "parallelism": "dp1"},
           "metric": "images/sec " + ("fwd+bwd" if mode == "train" else "forward-only inference"),
           "value": imgs / elapsed, "unit": "images/s", "dtype": dtype, "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3,
           "step_ms": {k: stats[k] for k in ("median", "min", "max", "p90", "n", "host_ms_median")},
           "roofline": roof}
    if gf:
        tf = gf * 1e9 * imgs / elapsed / 1e12
        peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
        ent["whole_step_tflops_nominal"] = tf
        ent["whole_step_frac_of_mfma_peak"] = tf / peak
    del job
    torch.cuda.empty_cache()
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--in-channels", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--arch", default="unetdc", choices=["unetdc", "unet"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other single-GPU BASELINE configurations (fp32 inference, 1024x1024 bs 4, fp32 training) "
                         "that the default N = 1 training run measures after the headline")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--adam", choices=["hip", "fused", "foreach"], default="hip",
                    help="hip = this repo's one-kernel Adam + weight re-pack (default); fused / foreach = torch.optim.Adam")
    ap.add_argument("--per-layer", action="store_true", help="print a per-call timing table to stderr (diagnostic)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "quantify"],
                    help="train = the headline metric (default); infer = forward-only eval (BASELINE configs[1]); "
                         "quantify = the droplet-quantification flow of quantify_droplets_batch.py (SURVEY 8 f1/f2)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  RCCL prints a five-line version banner on stdout when its first
    # communicator comes up (seen on the MI355X box with torch's bundled librccl), so file descriptor 1 is pointed at
    # stderr for the whole run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.mode == "quantify":
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        quantify_bench(args, real_stdout)
        return
    from unet_dc_segmentation_amd import dp as dpmod
    # RCCL ("nccl") is the production backend; UNETDC_DIST_BACKEND=gloo + UNETDC_BENCH_DEVICE=0 lets several
    # ranks rehearse the multi-rank code path on ONE card (RCCL refuses two ranks per device).
    backend = os.environ.get("UNETDC_DIST_BACKEND", "nccl")
    rank, local, world = dpmod.init_from_env(backend)
    if "UNETDC_BENCH_DEVICE" in os.environ:
        local = int(os.environ["UNETDC_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to mislabel the run")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    job = Job(args.mode, args.dtype, args.batch, args.size, args.in_channels, args.arch, args.adam, dev, rank)
    model = job.model
    # UNETDC_DP_FORCE=1: a one-rank process group whose collectives are really issued (RCCL rehearsal on a one-GPU box)
    force_dp = os.environ.get("UNETDC_DP_FORCE") == "1"
    # UNETDC_DP_BUCKET_MB / UNETDC_DP_MAX_BUCKET_MB: bucket policy of the gradient exchange (experiments; default 16 / 32 MiB)
    bkt = {}
    if os.environ.get("UNETDC_DP_BUCKET_MB"):
        bkt["bucket_bytes"] = int(float(os.environ["UNETDC_DP_BUCKET_MB"]) * (1 << 20))
        bkt["max_bucket_bytes"] = int(float(os.environ.get("UNETDC_DP_MAX_BUCKET_MB", 2 * float(os.environ["UNETDC_DP_BUCKET_MB"]))) * (1 << 20))
    wrapper = dpmod.DataParallel(model, single_rank_collectives=force_dp, **bkt) if (world > 1 or force_dp) else None
    job.make_optimizer()
    step = job.step

    # initialisation, not warm-up: the engine behind the module allocates its activation / gradient buffers and builds its
    # descriptor tables on the first forward / backward / optimizer step (like cudnn.benchmark or lazy module init would);
    # one untimed step does that even when the caller asks for --warmup 0
    step()
    torch.cuda.synchronize()
    if args.per_layer:
        for _ in range(args.warmup):
            step()
        per_layer_table(step, args)
    elapsed, step_stats, loss = timed_region(step, args.steps, barrier=dist.barrier if world > 1 else None, warmup=args.warmup)
    ranks = None
    if world > 1:
        # MAX over ranks is the job's time; every rank's own time and median step go into the line as well
        mine = torch.tensor([elapsed, step_stats["median"], step_stats["max"]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = torch.stack(allr).cpu()
        elapsed = float(per[:, 0].max())
        med = sorted(float(v) for v in per[:, 1])
        ranks = {"ranks_seen": dist.get_world_size(), "elapsed_s_per_rank": [float(v) for v in per[:, 0]],
                 "step_ms_median_across_ranks": {"min": med[0], "median": med[len(med) // 2], "max": med[-1]},
                 "step_ms_max_across_ranks": float(per[:, 2].max())}
    final_loss = float(loss.item())
    # ---- instrumented passes, all AFTER the timed region (every rank runs them: the collectives need all ranks) ----
    roofline = igemm_leg(step, args.dtype, args.mode, nsteps=3, headline=(args.mode == "train" and args.dtype == "bf16"))
    hbm = hbm_leg(step, 2 if args.dtype == "bf16" else 4) if args.mode == "train" else None
    coll = None
    if wrapper is not None and args.mode == "train":
        wrapper.start_trace()
        for _ in range(3):
            wrapper.mark_step_start()
            step()
        coll = wrapper.stop_trace()

    if rank == 0:
        nominal_gflop_img = NOMINAL_GFLOP_IMG.get((args.mode, args.size)) if args.in_channels == 1 else None
        imgs = args.batch * world * args.steps
        out = {
            "metric": (f"images/sec fwd+bwd, {args.size}x{args.size}x{args.in_channels} "
                       f"{'U-Net-DC' if args.arch == 'unetdc' else 'U-Net'}, bs={args.batch}/GPU") if args.mode == "train"
                      else (f"images/sec forward-only inference, {args.size}x{args.size}x{args.in_channels} "
                            f"{'U-Net-DC' if args.arch == 'unetdc' else 'U-Net'}, bs={args.batch}/GPU"),
            "value": imgs / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": job.workload(world) + (" (BASELINE configs[2]/[3])" if args.mode == "train" else " (BASELINE configs[1])"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "init_steps_before_warmup": 1,
                       "timed_region": "zero_grad, model(x), focal_dice_loss, backward, all-reduce, Adam step (" + args.adam + "), weight re-pack"
                                       if args.mode == "train" else "model(x) under no_grad + threshold, input resident in HBM"},
            "roofline": roofline,
            "hbm_leg": hbm,
            "step_ms": step_stats,
            "final_value": final_loss,
        }
        if nominal_gflop_img:
            out["whole_step_tflops_nominal"] = nominal_gflop_img * 1e9 * imgs / elapsed / 1e12
            out["whole_step_frac_of_mfma_peak"] = out["whole_step_tflops_nominal"] / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS)
        if wrapper is not None:
            out["allreduce_buckets_per_step"] = wrapper.stats["buckets"] / max(wrapper.stats["steps"], 1)
            out["collective"] = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "ranks": world,
                                 "payload_MB_per_step": wrapper.stats["elems"] * 4 / max(wrapper.stats["steps"], 1) / 1e6}
            if coll:
                out["collective"].update(coll)
        if ranks:
            out["ranks"] = ranks
        if world == 1 and wrapper is None and args.mode == "train" and not args.no_secondary:
            # the other single-GPU BASELINE configurations, each with its own timed region (<= 10 s together)
            del job, model, step
            torch.cuda.empty_cache()
            out["secondary"] = [
                secondary_entry(dev, "BASELINE configs[1]: forward-only inference, parity dtype", "infer", "f32", 8, 512, steps=15),
                secondary_entry(dev, "BASELINE configs[4] per GPU: 1024x1024 tiles bs 4", "train", "bf16", 4, 1024, steps=15),
                secondary_entry(dev, "configs[2] in the parity dtype: fp32 training step", "train", "f32", 8, 512, steps=8),
            ]
        if world == 1 and not args.no_cpu_baseline and args.mode == "train":
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.size, args.size, args.in_channels)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
