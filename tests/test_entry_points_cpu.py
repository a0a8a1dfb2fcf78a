"""CPU plumbing tests of the kept entry points (BASELINE config 0: quantify_droplets_batch.py
inference, random-init U-Net-DC, 4 x 512x512 synthetic images on PyTorch CPU) and of the host
logic either side of the hot path."""
import math
import os

import numpy as np
import pandas as pd
import torch
from PIL import Image


def _disc_image(rng, size=96, n=6):
    img = (rng.random((size, size, 3)) * 60).astype(np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    for _ in range(n):
        cy, cx, r = rng.integers(8, size - 8), rng.integers(8, size - 8), rng.integers(2, 7)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 230
    return img


def test_quantify_known_answers():
    """equivalent_diameter = sqrt(4*area/pi) and area_sqmicron = area/px^2 -- the format known answers
    of the reference's sample output (outputs/all_droplets.csv:2: area 18224 -> 152.327, px 3.45)."""
    import quantify_droplets_batch as q
    mask = np.zeros((40, 40), np.uint8)
    mask[2:6, 2:6] = 1            # 16 px square
    mask[10:12, 10] = 1           # 2 px (removed by min_area=3)
    mask[20, 20] = 1              # diagonal neighbours are separate objects (4-connectivity)
    mask[21, 21] = 1
    df = q.quantify(mask, 1, 3.45)
    assert len(df) == 4 and list(df.columns) == ["label", "area", "equivalent_diameter", "centroid-0", "centroid-1",
                                                 "area_sqmicron", "eq_diam_micron"]
    row = df[df["area"] == 16].iloc[0]
    assert abs(row["equivalent_diameter"] - math.sqrt(4 * 16 / math.pi)) < 1e-9
    assert abs(row["area_sqmicron"] - 16 / 3.45 ** 2) < 1e-9 and abs(row["centroid-0"] - 3.5) < 1e-9
    assert len(q.quantify(mask, 3, None)) == 1
    assert abs(math.sqrt(4 * 18224 / math.pi) - 152.327) < 1e-3
    assert q.quantify(np.zeros((8, 8), np.uint8), 1, None).empty


def test_droplet_table_reproduces_all_rows_of_the_reference_output():
    """All 303 rows of the reference's own sample output table (/root/reference/outputs/all_droplets.csv, committed as the data
    fixture tests/golden/ref_all_droplets.csv by tools/make_goldens.py): _droplet_table (quantify_droplets_batch.py:58-72) must
    give its equivalent_diameter, area_sqmicron and eq_diam_micron from (area, centroid) with px = 3.45, to the last digits the
    CSV prints, per image group and in label order."""
    import pandas as pd
    import quantify_droplets_batch as q
    ref = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "ref_all_droplets.csv"))
    assert len(ref) == 303 and list(ref.columns) == ["filename", "label", "area", "equivalent_diameter", "centroid-0",
                                                     "centroid-1", "area_sqmicron", "eq_diam_micron"]
    rows = 0
    for fname, grp in ref.groupby("filename", sort=False):
        df = q._droplet_table(grp["area"].to_numpy(), grp["centroid-0"].to_numpy(), grp["centroid-1"].to_numpy(), 3.45)
        assert list(df["label"]) == list(range(1, len(grp) + 1))
        for col in ("area", "equivalent_diameter", "centroid-0", "centroid-1", "area_sqmicron", "eq_diam_micron"):
            np.testing.assert_allclose(df[col].to_numpy(dtype=np.float64), grp[col].to_numpy(dtype=np.float64), rtol=1e-13, atol=0,
                                       err_msg=f"{fname}: {col}")
        rows += len(df)
    assert rows == 303


def test_quantify_cli_cpu(tmp_path, monkeypatch):
    import quantify_droplets_batch as q
    from models.model_2 import UNetDC
    monkeypatch.setattr(q, "DEVICE", "cpu")
    monkeypatch.setattr(q, "IMG_SIZE", 64)          # keep the CPU run short; 512 is exercised on the GPU box
    rng = np.random.default_rng(0)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    for i in range(4):
        Image.fromarray(_disc_image(rng)).save(img_dir / f"im{i}.png")
    torch.manual_seed(0)
    ckpt = tmp_path / "best_UNetDC_focal_model.pth"
    torch.save(UNetDC(3, 1).state_dict(), ckpt)
    out = q.main(["--img_dir", str(img_dir), "--ckpt_path", str(ckpt), "--out_dir", str(tmp_path / "out"),
                  "--batch", "3", "--prob_thresh", "0.3", "--skip_excel", "--skip_histogram", "--save_overlays",
                  "--background_radius", "15", "--px_per_micron", "3.45"])
    summary = pd.read_csv(out / "summary_per_image.csv")
    assert list(summary.columns) == ["filename", "droplet_count", "total_area_px"] and len(summary) == 4
    for i in range(4):
        m = np.array(Image.open(out / "predicted_masks" / f"im{i}_pred.png"))
        assert m.shape == (96, 96) and set(np.unique(m)).issubset({0, 255})
        assert (out / "overlays" / f"im{i}_overlay.png").exists()


def test_rolling_ball_and_dataset(tmp_path):
    from utils.data_loader import SegmentationDataset, SyntheticDropletDataset, rolling_ball_correction_rgb
    rng = np.random.default_rng(1)
    img = _disc_image(rng)
    out = rolling_ball_correction_rgb(img, 15)
    assert out.shape == img.shape and out.dtype == np.uint8 and out.max() == 255 and out.min() == 0
    (tmp_path / "i").mkdir()
    (tmp_path / "m").mkdir()
    Image.fromarray(img).save(tmp_path / "i" / "a.png")
    Image.fromarray(((img[..., 0] > 200) * 255).astype(np.uint8)).save(tmp_path / "m" / "a.png")
    ds = SegmentationDataset(str(tmp_path / "i"), str(tmp_path / "m"), ["a.png"], ["a.png"], size=64, radius=15)
    x, m, (oh, ow), name = ds[0]
    assert x.shape == (3, 64, 64) and m.shape == (1, 64, 64) and (oh, ow) == (96, 96) and name == "a.png"
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0 and set(m.unique().tolist()) <= {0.0, 1.0}
    s = SyntheticDropletDataset(3, 64, 1, seed=3)[1]
    assert s[0].shape == (1, 64, 64) and s[1].shape == (1, 64, 64) and 0 < float(s[1].mean()) < 0.5


def test_train_entry_point_cpu(tmp_path):
    import train_DC_focal as t
    ckpt = tmp_path / "best.pth"
    hist = t.main(["--synthetic", "--synthetic_len", "10", "--img_size", "32", "--batch", "2", "--epochs", "1",
                   "--steps", "2", "--workers", "0", "--in_channels", "1", "--device", "cpu",
                   "--ckpt_path", str(ckpt)])
    assert len(hist) == 1 and np.isfinite(hist[0]["train_loss"]) and np.isfinite(hist[0]["val_loss"])
    if ckpt.exists():
        sd = torch.load(ckpt, weights_only=True)
        assert len(sd) == 136
        # final test evaluation of the best checkpoint on the held-out split (reference train_DC_focal.py:365-402, :452-467)
        tr = hist.test
        assert tr is not None and np.isfinite(tr["test_loss"]) and 0.0 <= tr["test_dice"] <= 1.0 and 0.0 <= tr["test_acc"] <= 1.0
        cm = np.asarray(tr["confusion"])
        assert cm.shape == (2, 2) and cm.sum() == 2 * 32 * 32            # 10 tiles: 6 train / 2 val / 2 test
        assert abs(tr["test_acc"] - (cm[0, 0] + cm[1, 1]) / cm.sum()) < 1e-12


def test_train_entry_point_keeps_the_ragged_last_training_batch(tmp_path, monkeypatch):
    """The reference's train loader has no drop_last (train_DC_focal.py:200): with 7 training tiles at batch 2 the epoch has
    FOUR steps and the last one sees a single tile."""
    import train_DC_focal as t
    seen = []
    orig = t.focal_dice_loss
    monkeypatch.setattr(t, "focal_dice_loss", lambda pred, tgt, **kw: (seen.append(pred.shape[0]), orig(pred, tgt, **kw))[1])
    hist = t.main(["--synthetic", "--synthetic_len", "11", "--img_size", "32", "--batch", "2", "--epochs", "1",
                   "--workers", "0", "--in_channels", "1", "--device", "cpu", "--ckpt_path", str(tmp_path / "b.pth"),
                   "--no_test_eval"])
    assert hist.test is None
    assert seen[:4] == [2, 2, 2, 1], seen                               # 11 tiles: 2 test, 2 validation, 7 training


def test_hip_path_fails_loudly_without_library(monkeypatch, tmp_path):
    """No silent fallback: a missing libunetdc_hip.so is an error, not a PyTorch code path."""
    import pytest
    from unet_dc_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.UnetdcError, match="not found"):
        _lib.load()


def test_calculate_metrics_matches_sklearn():
    """Five return values like the reference (utils/metrics_DC.py:75-85, which calls sklearn): checked against
    sklearn itself, including the all-negative corner (zero_division=1)."""
    from sklearn.metrics import confusion_matrix, f1_score, precision_score, recall_score
    from utils.metrics_DC import calculate_metrics
    g = torch.Generator().manual_seed(3)
    for yt, yp in [((torch.rand(2, 1, 16, 16, generator=g) < 0.3).float(), torch.rand(2, 1, 16, 16, generator=g)),
                   (torch.zeros(1, 1, 8, 8), torch.rand(1, 1, 8, 8, generator=g) * 0.2)]:
        p, r, f1, sp, cm = calculate_metrics(yt, yp)
        a, b = yt.view(-1).numpy(), (yp > 0.3).float().view(-1).numpy()
        assert abs(p - precision_score(a, b, average="binary", zero_division=1)) < 1e-12
        assert abs(r - recall_score(a, b, average="binary", zero_division=1)) < 1e-12
        assert abs(f1 - f1_score(a, b, average="binary", zero_division=1)) < 1e-12
        ref = confusion_matrix(a, b, labels=[0, 1])
        assert np.array_equal(cm, ref) and cm.shape == (2, 2)
        tn, fp = ref[0]
        assert abs(sp - (tn / (tn + fp) if tn + fp else 0)) < 1e-12


def test_dataset_transform_conventions_and_worker_seeding(tmp_path):
    """SegmentationDataset accepts the albumentations protocol the reference passes (keyword call, dict result,
    utils/data_loader.py:59-62) and a plain callable; TrainAugment draws a different stream per DataLoader worker and
    per epoch (the generator is derived from the worker's seed, which the DataLoader re-draws every epoch)."""
    from torch.utils.data import DataLoader, Dataset
    from utils.data_loader import SegmentationDataset, TrainAugment
    rng = np.random.default_rng(5)
    img = _disc_image(rng)
    (tmp_path / "i").mkdir()
    (tmp_path / "m").mkdir()
    Image.fromarray(img).save(tmp_path / "i" / "a.png")
    Image.fromarray(((img[..., 0] > 200) * 255).astype(np.uint8)).save(tmp_path / "m" / "a.png")
    seen = {}

    def kw_only(*, image, mask):                          # albumentations style
        seen["kw"] = True
        return {"image": torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1), "mask": mask}

    def positional(img, mask):
        seen["pos"] = True
        return img[:, ::-1], mask[:, ::-1]
    mk = lambda tf: SegmentationDataset(str(tmp_path / "i"), str(tmp_path / "m"), ["a.png"], ["a.png"],   # noqa: E731
                                        transform=tf, size=32, radius=9)
    base = mk(None)[0]
    a = mk(kw_only)[0]
    b = mk(positional)[0]
    assert seen == {"kw": True, "pos": True}
    assert a[0].shape == (3, 32, 32) and torch.equal(a[0], base[0]) and torch.equal(a[1], base[1])
    assert torch.equal(b[0], base[0].flip(-1)) and torch.equal(b[1], base[1].flip(-1))
    full = mk(TrainAugment(1))[0]
    assert full[0].shape == (3, 32, 32) and full[1].shape == (1, 32, 32) and set(full[1].unique().tolist()) <= {0.0, 1.0}

    class Probe(Dataset):                                 # returns the augmenter's first random draws in this process
        def __init__(self):
            self.aug = TrainAugment(7)

        def __len__(self):
            return 4

        def __getitem__(self, i):
            return torch.tensor(self.aug._generator().random(3))
    loader = DataLoader(Probe(), batch_size=1, num_workers=2)
    e1 = torch.cat(list(loader))
    e2 = torch.cat(list(loader))
    assert not torch.equal(e1[0], e1[1])                  # worker 0 and worker 1 differ
    assert not torch.equal(e1, e2)                        # the next epoch does not replay the sequence


def test_bench_gpus_flag_never_mislabels_a_run():
    """`python bench.py --gpus N` without a torchrun environment launches its own ranks -- and refuses (non-zero exit,
    nothing on stdout) when the host has fewer than N GPUs, instead of reporting a one-rank number as an N-GPU one;
    under a torchrun environment a WORLD_SIZE that contradicts --gpus is refused as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing to mislabel" in r.stderr


def test_bench_roofline_bookkeeping():
    """Executed-vs-nominal FLOPs (block-level tap skipping: d = 16 on a 32 x 32 map keeps 6 of 9 taps per 8-row block), algorithmic bytes
    of the HBM-bound entry points, and the kernel-source stamp of the PMC traffic file."""
    import bench
    fwd = lambda n, h, w, cin, cout, d: (0, cin, 0, 0, None, None, 0, cout, 0, None, n, h, w, cin, cout, d, 1, 0)   # noqa: E731
    assert abs(bench.executed_fraction(fwd(8, 32, 32, 1024, 1024, 16), "unetdc_conv3x3_fwd") - 6 / 9) < 1e-12
    assert bench.executed_fraction(fwd(8, 64, 64, 512, 512, 1), "unetdc_conv3x3_fwd") == 1.0
    f8 = bench.executed_fraction(fwd(8, 64, 64, 512, 512, 8), "unetdc_conv3x3_fwd")
    assert 0.7 < f8 < 1.0                                    # d = 8 on 64 x 64: edge blocks lose their outer taps
    fl, nbytes = bench.igemm_flops("unetdc_conv3x3_fwd", fwd(8, 64, 64, 512, 512, 1))
    assert fl == 2.0 * 8 * 64 * 64 * 512 * 512 * 9 and nbytes == (8 * 64 * 64 * 1024 + 9 * 512 * 512) * 2
    apply_args = (1, 64, 1, 1, 1, 64, 1, 64, 8, 512, 512, 64, 1, 0)
    assert bench.hbm_bytes("unetdc_bn_relu_apply", apply_args, 2) == 8 * 512 * 512 * 64 * 2 * 2.25
    h1, h2 = bench.kernel_source_hash(), bench.kernel_source_hash()
    assert h1 == h2 and len(h1) == 16


def test_bench_pmc_traffic_lookup_matches_full_template_arguments(tmp_path, monkeypatch):
    """bench.pmc_traffic: the C ABI reports "igemm_lattice_wide_kernel<1>" (+ " bnin"), rocprofv3 prints the full template argument
    list; a traffic file measured on OTHER kernel sources (stamp mismatch) is never used."""
    import json

    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    table = {"void unetdc::igemm_lattice_wide_kernel<1, false>(unetdc::IgemmParams, unetdc::LatticeParams)": {"bytes_corrected": 111.0},
             "void unetdc::igemm_lattice_wide_kernel<1, true>(unetdc::IgemmParams, unetdc::LatticeParams)": {"bytes_corrected": 222.0},
             "void unetdc::igemm_dma16_kernel<2, 4, 8, 2>(unetdc::IgemmParams)": {"bytes_corrected": 333.0}}
    (prof / "r05_pmc_traffic.json").write_text(json.dumps({"kernel_source_sha16": bench.kernel_source_hash(), "kernels": table}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("igemm_lattice_wide_kernel<1>") == 111.0
    assert bench.pmc_traffic("igemm_lattice_wide_kernel<1> bnin") == 222.0
    assert bench.pmc_traffic("igemm_dma16_kernel<2, 4, 8>") is None or bench.pmc_traffic("igemm_dma16_kernel<2, 4, 8, 2>") == 333.0
    (prof / "r05_pmc_traffic.json").write_text(json.dumps({"kernel_source_sha16": "0" * 16, "kernels": table}))
    assert bench.pmc_traffic("igemm_lattice_wide_kernel<1>") is None


def test_bench_timed_region_has_no_collector_pass_and_no_per_call_events():
    """bench.timed_region (the function the headline number comes from, run here with host-clock stand-ins for the HIP events):
    the cyclic collector is frozen + disabled for the region (no generation-2 pass can land in a step; the step below builds
    reference cycles on purpose), the per-call timing of _lib.call is off (no event is created inside the loop), the
    collector's state is restored afterwards, and the JSON fields the round-4 verdict asked for are there."""
    import gc
    import time

    import bench
    from unet_dc_segmentation_amd import _lib

    class HostEvent:
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    seen = {"timing_on": 0, "gc_enabled": 0, "calls": 0}
    ev0 = _lib.events_created

    def step():
        seen["calls"] += 1
        seen["timing_on"] += _lib._timing is not None
        seen["gc_enabled"] += gc.isenabled()
        junk = []
        for _ in range(2000):                      # reference cycles: would trigger collections if the collector were on
            a, b = [], []
            a.append(b)
            b.append(a)
            junk.append(a)
        return len(junk)

    assert gc.isenabled()
    frozen0 = gc.get_freeze_count()
    elapsed, stats, last = bench.timed_region(step, 12, make_event=HostEvent, sync=lambda: None,
                                              alloc_stats=lambda: {"num_device_alloc": 7, "num_alloc_retries": 0})
    assert seen == {"timing_on": 0, "gc_enabled": 0, "calls": 12} and last == 2000
    assert _lib.events_created == ev0
    assert gc.isenabled() and gc.get_freeze_count() == frozen0          # restored
    assert stats["gc_events"] == [] and stats["device_allocs_in_timed_region"] == 0 and stats["n"] == 12
    for k in ("median", "max", "max_at_step", "host_ms_at_max", "host_ms_median", "host_lead_ms_at_max", "alloc_retries_in_timed_region"):
        assert k in stats, k
    assert 0 < stats["min"] <= stats["median"] <= stats["max"] and elapsed > 0
    # an explicit full collection inside a step IS reported (generation + duration), so a stall can be attributed
    def step_gc():
        gc.collect()
    _, stats2, _ = bench.timed_region(step_gc, 3, make_event=HostEvent, sync=lambda: None, alloc_stats=lambda: {})
    assert len(stats2["gc_events"]) == 3 and all(e["generation"] == 2 and e["ms"] >= 0 for e in stats2["gc_events"])


def test_engine_flat_gradient_buffer_is_reused_only_when_unreferenced():
    """engine.UNetEngine._flat_grads: ONE flat gradient buffer across steps while nothing else refers to its storage; a fresh
    one while gradient views of the previous backward are still alive (accumulation, kept gradients).  Pure host logic: run on
    a stand-in object with CPU tensors."""
    from unet_dc_segmentation_amd import engine

    class Stub:
        _flat, nparams, device = None, 1000, torch.device("cpu")
    eng = Stub()
    get = engine.UNetEngine._flat_grads
    f1 = get(eng)
    p1 = f1.data_ptr()
    del f1
    assert get(eng).data_ptr() == p1                       # nothing held it: same buffer again
    view = get(eng)[10:20].view(2, 5)                      # a .grad-like view stays alive ...
    f2 = get(eng)
    assert f2.data_ptr() != p1                             # ... so the next backward must not write over it
    del view, f2
    assert get(eng).data_ptr() == eng._flat.data_ptr()


def test_fused_adam_cpu_formulas_match_torch():
    """On CPU tensors FusedAdam runs the per-tensor formulas in PyTorch (the reference's own path on a host without a
    GPU): same numbers as torch.optim.Adam, same state_dict layout."""
    from models.model_2 import UNetDC
    from unet_dc_segmentation_amd.optim import FusedAdam
    torch.manual_seed(0)
    a, b = UNetDC(1, 1), UNetDC(1, 1)
    b.load_state_dict(a.state_dict())
    oa, ob = FusedAdam(a, lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        for pa, pb in zip(a.parameters(), b.parameters()):
            pa.grad = torch.randn(pa.shape, generator=g)
            pb.grad = pa.grad.clone()
        oa.step()
        ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert sda["state"].keys() == sdb["state"].keys() and set(sda["state"][0]) == set(sdb["state"][0])
    assert torch.allclose(sda["state"][0]["exp_avg"], sdb["state"][0]["exp_avg"], rtol=1e-6, atol=1e-10)
    # the round trip INTO torch.optim.Adam must also STEP correctly: full default set in the group (torch's step reads
    # weight_decay / amsgrad / maximize / ...), and one `step` tensor per parameter (a shared one would be advanced once
    # per parameter by the non-fused torch implementation and spoil the bias corrections)
    assert sda["param_groups"][0].keys() == sdb["param_groups"][0].keys()
    steps = [st["step"] for st in sda["state"].values()]
    assert len({id(t) for t in steps}) == len(steps) and all(float(t) == 3.0 for t in steps)
    oc = torch.optim.Adam(a.parameters(), lr=1e-2, foreach=False)
    oc.load_state_dict(sda)
    for pa, pb in zip(a.parameters(), b.parameters()):
        pa.grad = torch.randn(pa.shape, generator=g)
        pb.grad = pa.grad.clone()
    oc.step()                                               # torch Adam continues a's optimisation from FusedAdam's state
    ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)
    assert all(float(st["step"]) == 4.0 for st in oc.state.values())
    # ... and back: FusedAdam picks up torch.optim.Adam's state
    od = FusedAdam(a, lr=1e-2)
    od.load_state_dict(ob.state_dict())
    for pa, pb in zip(a.parameters(), b.parameters()):
        pa.grad = torch.randn(pa.shape, generator=g)
        pb.grad = pa.grad.clone()
    od.step()
    ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)


def test_module_copies_and_pickles_without_its_engines():
    """deepcopy / pickle of the drop-in module: parameters and buffers travel, device-side engines and hooks do not."""
    import copy
    import io
    from models.model_2 import UNetDC
    m = UNetDC(1, 1)
    m._engines[("fake", (1, 1, 16, 16))] = object()         # stands in for an engine with device buffers
    m._weights = object()
    m.grad_ready_hook = lambda *a: None
    c = copy.deepcopy(m)
    assert c._engines == {} and c._weights is None and c.grad_ready_hook is None and c._engine is None
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)                        # the reference's checkpoint format (train_DC_focal.py:326)
    buf.seek(0)
    c.load_state_dict(torch.load(buf, weights_only=True))
    x = torch.rand(1, 1, 16, 16)
    assert torch.equal(m.eval()(x), c.eval()(x))


def test_nearest_resize_restates_cv2_rule():
    """cv2.resize(..., INTER_NEAREST) picks source index min(floor(d * src / dst), src - 1) (known answers computed by
    hand: 4 -> 6 gives 0,0,1,2,2,3; 5 -> 2 gives 0,2); identity at equal size."""
    from unet_dc_segmentation_amd.droplets import nearest_index, resize_nearest_cv2
    assert nearest_index(6, 4).tolist() == [0, 0, 1, 2, 2, 3]
    assert nearest_index(2, 5).tolist() == [0, 2]
    m = np.arange(20, dtype=np.uint8).reshape(4, 5)
    assert np.array_equal(resize_nearest_cv2(m, 5, 4), m)
    up = resize_nearest_cv2(m, 10, 8)
    assert up.shape == (8, 10) and np.array_equal(up[::2, ::2], m)


def test_opencv_restatements_known_answers():
    """numpy restatements of the OpenCV operators (the CPU preprocessing path and the yardstick of csrc/preprocess.hip):
    ellipse element of getStructuringElement (5x5 and the 50x50 of the reference), erosion / dilation against a brute-force
    evaluation of the definition, min-max normalisation rounding, identity and 2x bilinear resize."""
    from utils.data_loader import (ellipse_spans, morph_cv2, normalize_minmax_u8, resize_linear_cv2_u8,
                                   rolling_ball_correction_rgb)
    # cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (5, 5)) = rows 00100 / 11111 / 11111 / 11111 / 00100
    assert ellipse_spans(5) == [(2, 3), (0, 5), (0, 5), (0, 5), (2, 3)]
    sp = ellipse_spans(50)
    assert sp[0] == (25, 26) and sp[25] == (0, 50) and sp[49] == (18, 33) and len(sp) == 50
    rng = np.random.default_rng(0)
    img = (rng.random((40, 47)) * 255).astype(np.uint8)

    def brute(plane, k, is_max):
        h, w = plane.shape
        r = k // 2
        out = np.zeros_like(plane)
        for y in range(h):
            for x in range(w):
                vals = []
                for i, (j1, j2) in enumerate(ellipse_spans(k)):
                    yy = y + i - r
                    lo, hi = max(x + j1 - r, 0), min(x + j2 - r, w)
                    if 0 <= yy < h and hi > lo:
                        vals.append(plane[yy, lo:hi].max() if is_max else plane[yy, lo:hi].min())
                out[y, x] = (max(vals) if is_max else min(vals)) if vals else (0 if is_max else 255)
        return out
    for k in (1, 2, 5, 8):
        for is_max in (False, True):
            assert np.array_equal(morph_cv2(img, k, is_max), brute(img, k, is_max)), (k, is_max)
    a = np.array([[10, 20, 30, 137]], np.uint8)
    assert normalize_minmax_u8(a).tolist() == [[0, 20, 40, 255]]            # 10 * 255/127 = 20.08, 20 * ... = 40.16
    assert not normalize_minmax_u8(np.full((3, 3), 7, np.uint8)).any()
    big = (rng.random((33, 21, 3)) * 255).astype(np.uint8)
    assert np.array_equal(resize_linear_cv2_u8(big, 21, 33), big)
    const = np.full((10, 10), 200, np.uint8)
    assert (resize_linear_cv2_u8(const, 23, 17) == 200).all()
    rb = rolling_ball_correction_rgb(big, 5)
    assert rb.shape == big.shape and rb.dtype == np.uint8 and rb.max() == 255 and rb.min() == 0
